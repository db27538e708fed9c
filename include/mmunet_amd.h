/*
 * mmunet_amd.h -- C-ABI of libmmunet_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the MM-UNet hot path.  Each entry point replaces one
 * pybind11 function of the reference's two CUDA torch-extensions; the
 * reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 *   mmu_selective_scan_fwd   <- selective_scan_cuda.fwd
 *        requirements/Mamba/mamba/csrc/selective_scan/selective_scan.cpp:226-336,495
 *   mmu_selective_scan_bwd   <- selective_scan_cuda.bwd
 *        requirements/Mamba/mamba/csrc/selective_scan/selective_scan.cpp:338-492,496
 *   mmu_causal_conv1d_fwd    <- causal_conv1d_cuda.causal_conv1d_fwd
 *        requirements/Mamba/causal-conv1d/csrc/causal_conv1d.cpp:130-189,330
 *   mmu_causal_conv1d_bwd    <- causal_conv1d_cuda.causal_conv1d_bwd
 *        requirements/Mamba/causal-conv1d/csrc/causal_conv1d.cpp:191-268,331
 *   mmu_causal_conv1d_update <- causal_conv1d_cuda.causal_conv1d_update
 *        requirements/Mamba/causal-conv1d/csrc/causal_conv1d.cpp:270-327,332
 *
 * Conventions (same as the reference's):
 *   - stateless; every launch goes to the hipStream_t passed in (`stream`,
 *     a `hipStream_t` cast to void*; NULL = the null stream); no host threads,
 *     no allocation, no synchronisation inside -> graph-capturable;
 *   - all pointers are DEVICE pointers; outputs and workspaces are allocated
 *     by the caller (the reference allocates them inside the pybind function;
 *     here that moved to the host shim so the ABI carries no torch types);
 *   - strides are in ELEMENTS; the sequence (L) stride of every [.,.,L]
 *     tensor must be 1 (selective_scan.cpp:252-253);
 *   - return value 0 = ok; non-zero = rejected/failed, message available via
 *     mmu_last_error() (thread-local).  The host shim turns it into the
 *     RuntimeError the reference's TORCH_CHECKs raise.
 *
 * dtype codes: 0 = float32, 1 = bfloat16 (I/O tensors u, delta, z, B, C, out,
 * dout, du, ddelta, dz / x, out, dout, dx).  A, D, delta_bias, conv weights,
 * chunk states and every gradient accumulated across batch (dA, dB, dC, dD,
 * ddelta_bias, dweight, dbias) are always float32.
 */
#ifndef MMUNET_AMD_H
#define MMUNET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMU_DTYPE_F32 0
#define MMU_DTYPE_BF16 1

/* ---- library info ------------------------------------------------------ */
/* Version of the parameter-struct layouts below.  Bumped whenever a struct changes (fields are only ever appended);
 * a binding must refuse a library whose mmu_abi_version() differs from the header it was written against.
 *   1: round 1;  2: round 2 appended conv1d_bwd.workspace, morph.in_dtype, resize.dtype, conv3x3s.{in_dtype,
 *   dinput_addend, weight_native}, tri.dtype, norm.{x_dtype, act_dtype};  3: round 3 (fused small-map entry points). */
#define MMU_ABI_VERSION 12
int mmu_abi_version(void);
const char *mmu_last_error(void);

/* Chunk length (tokens) the scan kernels use for a given dstate and dtype.  The
 * chunk-state tensor `x` has shape [batch, dim, n_chunks, 2*dstate] float32 with
 * n_chunks = ceil(seqlen / chunk_len).  (The reference hard-codes 2048,
 * selective_scan.cpp:307; here it is smaller because chunks are what run in
 * parallel.)  x[b,d,c,2n+1] = state h_n at the END of chunk c, so
 * x[:, :, -1, 1::2] is the last state exactly as in
 * selective_scan_interface.py:40.  x[b,d,c,2n] = product of a_t over chunk c. */
int mmu_scan_chunk_len(int dstate, int dtype);

/* Bytes of float32 workspace mmu_selective_scan_bwd needs. */
size_t mmu_scan_bwd_workspace_bytes(int batch, int dim, int seqlen, int dstate, int dtype, int have_x);

/* ---- selective scan ------------------------------------------------------ */
typedef struct {
    int32_t batch, dim, seqlen, dstate, ngroups;
    int32_t dtype;           /* MMU_DTYPE_* of u, delta, z, B, C, out, out_z */
    int32_t delta_softplus;  /* bool */
    int32_t n_chunks;        /* must be ceil(seqlen / mmu_scan_chunk_len(dstate, dtype)) */
    const void *u;           /* [batch, dim, L]   strides (u_bs, u_ds, 1) */
    const void *delta;       /* [batch, dim, L] */
    const float *A;          /* [dim, dstate]     strides (A_ds, A_ns), real */
    const void *B;           /* [batch, G, dstate, L] strides (B_bs, B_gs, B_ns, 1) */
    const void *C;           /* same */
    const float *D;          /* [dim] or NULL */
    const void *z;           /* [batch, dim, L] or NULL */
    const float *delta_bias; /* [dim] or NULL */
    void *out;               /* [batch, dim, L] or NULL (y before gating) */
    void *out_z;             /* [batch, dim, L]; required iff z != NULL */
    float *x;                /* [batch, dim, n_chunks, 2*dstate] contiguous; required */
    int64_t u_bs, u_ds, delta_bs, delta_ds, z_bs, z_ds;
    int64_t out_bs, out_ds, out_z_bs, out_z_ds;
    int64_t A_ds, A_ns;
    int64_t B_bs, B_gs, B_ns, C_bs, C_gs, C_ns;
} mmu_scan_fwd_params;

int mmu_selective_scan_fwd(const mmu_scan_fwd_params *p, void *stream);

typedef struct {
    int32_t batch, dim, seqlen, dstate, ngroups;
    int32_t dtype;
    int32_t delta_softplus;
    int32_t n_chunks;
    const void *u, *delta;
    const float *A;
    const void *B, *C;
    const float *D;          /* or NULL */
    const void *z;           /* or NULL */
    const float *delta_bias; /* or NULL */
    const void *dout;        /* [batch, dim, L] */
    const float *x;          /* chunk states from the forward, or NULL (recomputed) */
    void *du, *ddelta;       /* [batch, dim, L], I/O dtype */
    float *dA;               /* [dim, dstate] contiguous, OVERWRITTEN (not accumulated) */
    float *dB, *dC;          /* [batch, G, dstate, L] float32, strides (dB_bs, dB_gs, dB_ns, 1) / dC_*, overwritten */
    float *dD;               /* [dim] or NULL */
    float *ddelta_bias;      /* [dim] or NULL */
    void *dz;                /* [batch, dim, L] (may be a strided view); required iff z */
    void *out_z;             /* optional: recomputed gated output (recompute_out_z) */
    float *workspace;        /* mmu_scan_bwd_workspace_bytes(...) bytes */
    int64_t u_bs, u_ds, delta_bs, delta_ds, z_bs, z_ds, dout_bs, dout_ds;
    int64_t du_bs, du_ds, ddelta_bs, ddelta_ds, dz_bs, dz_ds, out_z_bs, out_z_ds;
    int64_t A_ds, A_ns;
    int64_t B_bs, B_gs, B_ns, C_bs, C_gs, C_ns;
    int64_t dB_bs, dB_gs, dB_ns, dC_bs, dC_gs, dC_ns;
    int32_t dA_times_A;      /* non-zero: dA receives dA * A, the gradient of a parameter a with A = -exp(a) (Mamba's A_log,
                              * mamba_simple.py:209); needs a contiguous A.  0: dA as the reference returns it */
    const void *out;         /* optional (ABI 6): the forward's `out` = y before gating, [batch, dim, L] strides (out_bs, out_ds, 1),
                              * what the reference's backward reads (selective_scan.cpp:338 `out_`, selective_scan_bwd_kernel.cuh:
                              * dz and the recomputed out_z come from it).  NULL: y is recomputed from the states. */
    int64_t out_bs, out_ds;
} mmu_scan_bwd_params;

int mmu_selective_scan_bwd(const mmu_scan_bwd_params *p, void *stream);

/* ---- depthwise causal conv1d (channel-first, unit L stride) ------------- */
typedef struct {
    int32_t batch, dim, seqlen, width; /* width in 2..4 */
    int32_t dtype;                     /* of x / out */
    int32_t silu;                      /* bool */
    const void *x;                     /* [batch, dim, L] strides (x_bs, x_ds, 1) */
    const float *weight;               /* [dim, width] strides (w_ds, w_ws) */
    const float *bias;                 /* [dim] or NULL */
    void *out;                         /* [batch, dim, L] strides (out_bs, out_ds, 1) */
    int64_t x_bs, x_ds, out_bs, out_ds, w_ds, w_ws;
} mmu_conv1d_fwd_params;

int mmu_causal_conv1d_fwd(const mmu_conv1d_fwd_params *p, void *stream);

typedef struct {
    int32_t batch, dim, seqlen, width;
    int32_t dtype;
    int32_t silu;
    const void *x;
    const float *weight;
    const float *bias;  /* or NULL */
    const void *dout;   /* [batch, dim, L] strides (dout_bs, dout_ds, 1) */
    void *dx;           /* [batch, dim, L] strides (dx_bs, dx_ds, 1); may be a view */
    float *dweight;     /* [dim, width] contiguous float32, must be ZEROED by the caller (atomics) */
    float *dbias;       /* [dim] float32 zeroed, or NULL */
    int64_t x_bs, x_ds, dout_bs, dout_ds, dx_bs, dx_ds, w_ds, w_ws;
    float *workspace;   /* mmu_causal_conv1d_bwd_workspace_floats(...) floats: dweight / dbias are then OVERWRITTEN with
                         * sums formed in a fixed order (bit-reproducible, no zero-fill needed); NULL: dweight / dbias
                         * must be zeroed by the caller and are accumulated with float atomics, as the reference does
                         * (causal_conv1d_bwd.cu:256-268) */
} mmu_conv1d_bwd_params;

size_t mmu_causal_conv1d_bwd_workspace_floats(int batch, int dim, int seqlen);
int mmu_causal_conv1d_bwd(const mmu_conv1d_bwd_params *p, void *stream);

typedef struct {
    int32_t batch, dim, width;
    int32_t dtype;
    int32_t silu;
    const void *x;       /* [batch, dim] strides (x_bs, x_ds) */
    void *conv_state;    /* [batch, dim, width] strides (cs_bs, cs_ds, cs_ws), updated in place */
    const float *weight; /* [dim, width] */
    const float *bias;   /* or NULL */
    void *out;           /* [batch, dim] strides (out_bs, out_ds) */
    int64_t x_bs, x_ds, cs_bs, cs_ds, cs_ws, out_bs, out_ds, w_ds, w_ws;
} mmu_conv1d_update_params;

int mmu_causal_conv1d_update(const mmu_conv1d_update_params *p, void *stream);

/* ---- MorphMamba deformable sampling -------- */
/* Replaces, inside MMConv.forward, _coordinate_map_scaling (x and y) + the [B, H*K, W, 2] grid +
 * F.grid_sample(bilinear, zeros, align_corners=True) and their autograd
 * (src/UM_Net/MMUNet.py:196-242,259-263).  The column of tap k at pixel (h, w) is the integer
 * w + k - K/2, only the row coordinate y is learned, so the gather is a 2-tap vertical lerp.
 * All tensors contiguous; input / dinput in in_dtype (float32, or bfloat16 activations under autocast), the
 * coordinates, samples and their gradients float32 (grid_sample is on autocast's float32 list).
 *   fwd: out[b,c,h*K+k,w] from input[b,c,:,:] and y[b,k,h,w] (pixels, unclamped)
 *   bwd: dinput (written in full: gathered, far outliers added with float atomics) and dy (clamp mask applied)
 * out_layout MMU_MORPH_TOKENS_LAST stores the samples as [channels, taps, batch, height, width], i.e. the
 * (channels*taps) x (batch*height*width) matrix the K x 1 / stride K x 1 convolution that follows
 * (MMUNet.py:262, dsc_conv_x) multiplies by its weight viewed as [Cout, Cin*K]: that conv is then one GEMM. */
#define MMU_MORPH_BCHW 0
#define MMU_MORPH_TOKENS_LAST 1
typedef struct {
    int32_t batch, channels, height, width, taps;
    int32_t out_layout;  /* MMU_MORPH_BCHW or MMU_MORPH_TOKENS_LAST (layout of out and dout) */
    const void *input;   /* [batch, channels, height, width], in_dtype */
    const float *y;      /* [batch, taps, height, width] */
    float *out;          /* fwd: [batch, channels, height*taps, width] */
    const float *dout;   /* bwd: same shape as out */
    void *dinput;        /* bwd: [batch, channels, height, width], in_dtype */
    float *dy;           /* bwd: [batch, taps, height, width] */
    int32_t in_dtype;    /* MMU_DTYPE_F32 (0, the default of a zeroed struct) or MMU_DTYPE_BF16 */
    int32_t y_parts;     /* fwd: 0 / 1 = y is the row map; n > 1 = y holds n maps [n][batch, taps, height, width] whose
                          * sum is the row map (mmu_mamba_small_fwd's state-range partials), summed in fixed order */
    float *y_sum;        /* fwd, y_parts > 1: receives the summed map [batch, taps, height, width] (pass it as y to bwd) */
    const void *dinput_addend; /* bwd, optional: [batch, channels, height, width], in_dtype -- added to dinput (another
                                * consumer's gradient of the input: a residual connection; may be dinput itself) */
} mmu_morph_params;

int mmu_morph_sample_fwd(const mmu_morph_params *p, void *stream);
int mmu_morph_sample_bwd(const mmu_morph_params *p, void *stream);

/* ---- the final weight-gradient sums of a backward pass in one launch (csrc/deferred_reduce.hip) -------------------------
 * Between mmu_deferred_begin() and mmu_deferred_end() the launchers of mmu_gemm_nt_splitk, mmu_conv3x3_small_bwd (weight
 * gradient) and mmu_causal_conv1d_bwd (with a workspace) run their main kernels but only RECORD their final ordered sums
 * (one row of eight int64 per job: kind, partials, destinations, shape).  The caller keeps the partial buffers alive,
 * reads the rows with mmu_deferred_jobs (returns the number recorded; copies at most max_jobs rows; rows_out may be NULL),
 * places them and a work list of int32 pairs (job, workgroup) -- kind 0: ceil(row[4] / 64) workgroups, kind 1:
 * ceil(row[4] * ((row[6] * 10 + 3) & ~3) / 64), kind 2: ceil(row[5] / 16), kind 3 (mmu_zigzag_inproj_bwd /
 * mmu_coords_outproj_bwd with a workspace): ceil(row[7] / 64) -- in DEVICE memory and launches them all with
 * mmu_deferred_launch.  Same summation order as the kernels it replaces.  Process-wide state (autograd runs the
 * backward pass on its own thread): one scope at a time. */
void mmu_deferred_begin(void);
void mmu_deferred_pause(int paused);   /* 1: launchers reduce at once again (scope stays open); 0: record again */
void mmu_deferred_end(void);
int mmu_deferred_jobs(int64_t *rows_out, int max_jobs);
int mmu_deferred_job_workgroups(const int64_t *row);   /* workgroups of mmu_deferred_launch one recorded row (8 x int64) needs */
int mmu_deferred_launch(const int64_t *table, const int32_t *work, int n_work, void *stream);

/* ---- AdamW over many tensors in one launch (csrc/adamw_multi.hip) -------------------------------------------------------
 * torch.optim.AdamW's fused step (train.py:197-201 builds timm's adamw = torch.optim.AdamW) for a training step whose
 * parameters, gradients and optimizer state are static (a captured graph).  table: DEVICE array of n_tensors rows of eight
 * int64 {param, grad, exp_avg, exp_avg_sq, step (float scalar), numel, lr (float scalar), weight_decay (float bits in the
 * low 32)}; work: DEVICE array of n_work int32 pairs (tensor index, chunk index), one per 4,096 elements of every tensor.
 * All tensors float32 contiguous.  Increments every step counter, then updates in place.  No amsgrad / maximize. */
typedef struct {
    int32_t n_tensors, n_work;
    const int64_t *table;
    const int32_t *work;
    double beta1, beta2;
    float eps;
} mmu_adamw_params;

int mmu_adamw_multi(const mmu_adamw_params *p, void *stream);

/* ---- MMConv's sampling AFTER the channel mixing (csrc/morph_mix.hip) -------------------------------------------------
 * The row sampling S_k of tap k is linear and the same for every channel, so the K x 1 DSC convolution of the samples
 * (src/UM_Net/MMUNet.py:259-263) equals  out[o] = sum_k S_k(mixed[k * O + o])  with  mixed[k * O + o] = sum_c W[o][c][k]
 * x[c]  (a 1 x 1 convolution of the block's input, computed by the caller as a GEMM).  For the blocks that reduce the
 * channel count (MMUNet.py:344-349,357-359,424-430) the tensor that goes through HBM is Cin / Cout times smaller.
 * mixed / dmixed: [batch][taps * out_channels] planes of height x width contiguous floats, element strides mixed_bs /
 * mixed_cs over batch / channel (dmixed has the same strides); y, dy: [batch, taps, height, width]; out, dout:
 * [batch, out_channels, height, width] contiguous.  taps in {1, 3}.  bwd overwrites dmixed and dy. */
typedef struct {
    int32_t batch, out_channels, height, width, taps;
    const float *mixed;
    int64_t mixed_bs, mixed_cs;
    const float *y;
    float *out;          /* fwd */
    const float *dout;   /* bwd */
    float *dmixed;       /* bwd */
    float *dy;           /* bwd */
} mmu_morph_mix_params;

int mmu_morph_mix_sample_fwd(const mmu_morph_mix_params *p, void *stream);
int mmu_morph_mix_sample_bwd(const mmu_morph_mix_params *p, void *stream);

/* ---- bilinear resize, align_corners=True (a11: DecoderBlock x2, RCG edge map, side outputs) -------- */
/* F.interpolate(mode="bilinear", align_corners=True) on contiguous float32 [planes = batch*channels, h, w]
 * (src/UM_Net/MMUNet.py:362,384,571-575).  Backward is a gather (no atomics, dinput written in full). */
typedef struct {
    int32_t planes, in_h, in_w, out_h, out_w;
    const void *input;   /* fwd: [planes, in_h, in_w] */
    void *out;           /* fwd: [planes, out_h, out_w] */
    const void *dout;    /* bwd: [planes, out_h, out_w] */
    void *dinput;        /* bwd: [planes, in_h, in_w] */
    int32_t dtype;       /* element type of all four: MMU_DTYPE_F32 (0, the default of a zeroed struct) or
                          * MMU_DTYPE_BF16; the interpolation itself is float32 */
    const void *dinput_addend; /* bwd, optional: [planes, in_h, in_w] added to the gathered gradient (another consumer's
                                * gradient of the same input; may be dinput itself) */
} mmu_resize_params;

int mmu_bilinear_resize_fwd(const mmu_resize_params *p, void *stream);
int mmu_bilinear_resize_bwd(const mmu_resize_params *p, void *stream);

/* ---- 3x3 / stride 1 / pad 1 convolution with few output channels (a9: MMConv.offset_conv) ------------ */
/* nn.Conv2d(Cin, CO, 3, padding=1) for CO in {1, 2, 6, 8} (src/UM_Net/MMUNet.py:46,250: Cin -> 2K = 6), contiguous
 * NCHW; input / dinput in in_dtype (float32, or bfloat16 activations under autocast), everything else float32.
 * weight_t is the weight transposed to [Cin][3][3][CO] (a channel's CO*9 weights contiguous) -- or, with weight_native = 1,
 * the weight as the module holds it, [CO][Cin][3][3] (no transposed copy per call).
 *   fwd : out[b,co,h,w] = bias[co] + sum_{ci,ky,kx} W[co][ci][ky][kx] * in[b,ci,h+ky-1,w+kx-1]
 *   bwd : dinput (if non-NULL), dweight [CO][Cin][3][3] and dbias [CO] (if non-NULL; zeroed inside, float atomics) */
typedef struct {
    int32_t batch, in_channels, out_channels, height, width;
    const void *input;      /* [batch, in_channels, height, width], in_dtype */
    const float *weight_t;  /* [in_channels, 3, 3, out_channels] */
    const float *bias;      /* [out_channels] or NULL */
    float *out;             /* fwd: [batch, out_channels, height, width] */
    const float *dout;      /* bwd: same shape as out */
    void *dinput;           /* bwd, optional, in_dtype */
    float *dweight;         /* bwd, optional: [out_channels, in_channels, 3, 3] */
    float *dbias;           /* bwd, optional (only with dweight) */
    float *workspace;       /* fwd: mmu_conv3x3_small_fwd_splits() x (elements of out) floats when splits > 1;
                             * bwd: mmu_conv3x3_small_wgrad_workspace_floats() floats or NULL */
    int32_t in_dtype;       /* MMU_DTYPE_F32 (0, the default of a zeroed struct) or MMU_DTYPE_BF16 */
    const void *dinput_addend;  /* bwd, optional, shape / type of dinput: dinput = conv gradient + addend (the gradient
                                 * another consumer of the same input already produced; may alias dinput) */
    int32_t weight_native;  /* 1: weight_t points at the [out_channels, in_channels, 3, 3] weight itself */
} mmu_conv3x3s_params;

/* small images have too few pixels to fill the chip: the forward then slices the input channels and sums the
 * slices' partial outputs in a fixed order (reproducible); returns the number of slices (1 = no workspace) */
int mmu_conv3x3_small_fwd_splits(int batch, int in_channels, int height, int width);
int mmu_conv3x3_small_fwd(const mmu_conv3x3s_params *p, void *stream);
/* bwd with dweight: floats of workspace that select the deterministic row-walking weight-gradient kernels
 * (0 = shape not covered: width/4 must be a power of two <= 64; without workspace the atomic kernel runs) */
size_t mmu_conv3x3_small_wgrad_workspace_floats(int batch, int in_channels, int out_channels, int height, int width);
int mmu_conv3x3_small_bwd(const mmu_conv3x3s_params *p, void *stream);

/* ---- token re-orderings of the tri-directional Mamba block (first step) ---- */
/* requirements/mamba_simple.py:212-270: the "v3" block scans xz as is, token-reversed and slice-interleaved
 * (token i of slice s -> position i*nslices + s) and adds the three results after undoing the re-orderings.
 * Tensors are [rows][seqlen] with dense rows, float32 -- or bfloat16 (dtype) for 5..64 slices, float32 arithmetic.
 *   split   : a -> flip[L-1-t] = a[t],  slice[i*nslices + s] = a[s*(L/nslices) + i]
 *   combine : out[t] = a[t] + flip[L-1-t] + slice[i*nslices + s]     (t = s*(L/nslices) + i)
 * Each is the adjoint of the other (so each serves as the other's backward). */
typedef struct {
    int32_t rows, seqlen, nslices;
    const void *a;    /* split: input;  combine: first addend (original token order) */
    void *flip;       /* split: output; combine: input (token-reversed order) */
    void *slice;      /* split: output; combine: input (slice-interleaved order) */
    void *out;        /* combine: output */
    int32_t dtype;    /* MMU_DTYPE_F32 (0) or MMU_DTYPE_BF16 */
} mmu_tri_params;

int mmu_tri_split(const mmu_tri_params *p, void *stream);
int mmu_tri_combine(const mmu_tri_params *p, void *stream);

/* ---- f2: the three token orders folded into the block's conv1d and its gate (csrc/tri_fused.hip) ---------- */
/* requirements/mamba_simple.py:212-270 calls mamba_inner_fn_no_out_proj (selective_scan_interface.py:155-289) on xz, on
 * xz.flip(-1) and on the slice-interleaved xz, and adds out + out_b.flip(-1) + unslice(out_s).  Only the conv1d reads x
 * and the gate z is the same tensor in every direction, so:
 *   tri_conv_fwd : x (batch, dim, L; the x half of xz, any batch / channel stride) -> the three silu(causal_conv1d)
 *                  outputs [dim][batch][L], each in ITS scan order (f: natural, b: position L-1-t, s: position i*nslices+s
 *                  for token s*(L/nslices)+i), with the three directions' own weights [dim][4] / biases [dim]
 *                  (causal_conv1d_fwd.cu:39-158, width 4, SiLU);
 *   tri_conv_bwd : the three d conv_out (scan order) -> dx (natural order) and the three dweight / dbias; per-block
 *                  partial sums in `workspace` (mmu_tri_conv_bwd_workspace_floats() floats) added in a fixed order;
 *   tri_gate_fwd : out[t] = silu(z[t]) * (y_f[t] + y_b[L-1-t] + y_s[i*nslices+s]), y_* the un-gated scan outputs
 *                  (selective_scan with z = NULL), [dim][batch][L] in scan order  (= mamba_simple.py:270's sum);
 *   tri_gate_bwd : dout -> dz (natural order) and the three dy in scan order.
 * float32, 4 <= nslices <= 64, L divisible by nslices. */
typedef struct {
    int32_t batch, dim, seqlen, nslices, dtype;
    const void *x;
    int64_t x_bs, x_ds;
    const float *weight_f, *weight_b, *weight_s;
    const float *bias_f, *bias_b, *bias_s;        /* NULL: no bias */
    void *out_f, *out_b, *out_s;                  /* fwd */
    const void *dout_f, *dout_b, *dout_s;         /* bwd */
    void *dx;
    int64_t dx_bs, dx_ds;
    float *dweight_f, *dweight_b, *dweight_s;
    float *dbias_f, *dbias_b, *dbias_s;           /* NULL: not wanted */
    float *workspace;
} mmu_tri_conv_params;

typedef struct {
    int32_t batch, dim, seqlen, nslices, dtype;
    const void *z;
    int64_t z_bs, z_ds;
    const void *y_f, *y_b, *y_s;
    void *out;                                    /* fwd */
    int64_t out_bs, out_ds;
    const void *dout;                             /* bwd */
    int64_t dout_bs, dout_ds;
    void *dz;
    int64_t dz_bs, dz_ds;
    void *dy_f, *dy_b, *dy_s;
} mmu_tri_gate_params;

int mmu_tri_conv_fwd(const mmu_tri_conv_params *p, void *stream);
size_t mmu_tri_conv_bwd_workspace_floats(int batch, int dim, int seqlen, int nslices);
int mmu_tri_conv_bwd(const mmu_tri_conv_params *p, void *stream);
int mmu_tri_gate_fwd(const mmu_tri_gate_params *p, void *stream);
int mmu_tri_gate_bwd(const mmu_tri_gate_params *p, void *stream);

/* ---- MM_Net's stem: nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False) (csrc/stem7_mfma.hip) ---- */
/* src/UM_Net/MMUNet.py:492.  Forward and weight gradient on the bf16 matrix cores with float32-grade products (both
 * operands split into three bf16 parts, six MFMAs per product, float32 accumulation).  input [batch][3][height][width],
 * out / dout [batch][64][height/2][width/2], weight / dweight [64][3][7][7], all float32, contiguous; height even, width
 * a multiple of 16.  workspace: mmu_stem7_workspace_bytes(batch, height, width, backward) bytes, 16-byte aligned (forward:
 * the weight's bf16 images, written by the call; weight gradient: per-workgroup partial sums, added in a fixed order).
 * The image is the network's input: there is no input gradient. */
typedef struct {
    int32_t batch, height, width;
    const float *input;
    const float *weight;      /* fwd */
    float *out;               /* fwd */
    const float *dout;        /* wgrad */
    float *dweight;           /* wgrad */
    void *workspace;
} mmu_stem7_params;

size_t mmu_stem7_workspace_bytes(int batch, int height, int width, int backward);
int mmu_stem7_fwd(const mmu_stem7_params *p, void *stream);
int mmu_stem7_wgrad(const mmu_stem7_params *p, void *stream);

/* ---- GroupNorm [-> BatchNorm2d] [-> ReLU | tanh] as one normalisation (a9/a11 blocks) ------------------ */
/* nn.GroupNorm(groups, C) optionally followed by nn.BatchNorm2d(C) (training or eval statistics) and an
 * activation (src/UM_Net/MMUNet.py:250,265 + :344-349,357-359,424-430,436-452), contiguous NCHW.  Two element
 * types (MMU_DTYPE_*): x_dtype for input / dinput (what the producer -- a GEMM or a convolution -- emits), act_dtype
 * for out, residual, dout, act_out, dresidual (bf16 under autocast); statistics and parameters are float32.
 * Both normalisations are affine in x per (batch, channel) once the statistics are known, and the statistics
 * follow from the per-(batch, channel) moments of x: two passes forward, two backward.
 * Buffers the caller owns: s1, s2, scale, shift [batch*channels]; mu, rstd [batch*groups]; bn_mean, bn_rstd
 * [channels]; all written by fwd and read by bwd.  workspace (bwd): mmu_norm_fused_workspace_floats() floats. */
#define MMU_ACT_NONE 0
#define MMU_ACT_RELU 1
#define MMU_ACT_TANH 2
typedef struct {
    int32_t batch, channels, groups, hw, has_bn, training, act;
    int32_t has_gn;           /* 0: BatchNorm2d [-> act] alone (then groups == channels, has_bn == 1) */
    int32_t dinput_channel_major;  /* bwd: write dinput as [channels][batch][hw] (for a tokens-last consumer) */
    float gn_eps, bn_eps, momentum;
    const void *input;        /* [batch, channels, hw], x_dtype */
    const float *gn_weight;   /* [channels] or NULL */
    const float *gn_bias;     /* [channels] or NULL */
    const float *bn_weight;   /* [channels] or NULL */
    const float *bn_bias;     /* [channels] or NULL */
    const float *pre_bias;    /* [channels] or NULL: normalise input + pre_bias[c] (the bias of the conv before) */
    const void *residual;     /* fwd, or NULL: out = relu(norm(input) + residual)  (ResidualBlock, MMUNet.py:455-467); act_dtype */
    float *running_mean;      /* [channels]: updated in training mode (may be NULL), used in eval mode */
    float *running_var;
    void *out;                /* fwd, act_dtype */
    float *s1, *s2, *mu, *rstd, *bn_mean, *bn_rstd, *scale, *shift;   /* saved statistics */
    const void *dout;         /* bwd, act_dtype */
    const void *act_out;      /* bwd, residual mode: the forward output (its sign is the ReLU mask), act_dtype */
    void *dinput;             /* bwd, x_dtype */
    void *dresidual;          /* bwd, residual mode: gradient of the residual input, act_dtype */
    float *dgn_weight, *dgn_bias, *dbn_weight, *dbn_bias, *dpre_bias; /* bwd, each optional */
    float *workspace;         /* bwd */
    int32_t x_dtype, act_dtype;   /* MMU_DTYPE_F32 (0, the default of a zeroed struct) or MMU_DTYPE_BF16 */
} mmu_norm_params;

size_t mmu_norm_fused_workspace_floats(int batch, int channels, int groups);
int mmu_norm_fused_fwd(const mmu_norm_params *p, void *stream);
int mmu_norm_fused_bwd(const mmu_norm_params *p, void *stream);

/* ---- dense 3x3 / stride 1 / padding 1 convolution on the bf16 matrix cores with float32 accuracy ------------- */
/* nn.Conv2d(Cin, Cout, 3, padding=1) of CBAM (src/UM_Net/MMUNet.py:313-338) and of model.py's Unet (model.py:5-20),
 * float32 NCHW; in_channels % 16 == 0, out_channels % 64 == 0, width % 4 == 0.  Each product is evaluated as
 * xh*wh + xh*wl + xl*wh on bf16 hi/lo splits (float32 accumulation, ~2^-16 relative per product).
 * transposed = 0: out = conv(input, weight[out_channels][in_channels][3][3]) + bias.
 * transposed = 1: the input gradient -- `input` is dout [batch, in_channels, H, W], `weight` the forward weight
 *   [in_channels][out_channels][3][3] (read transposed and flipped), `out` is dinput [batch, out_channels, H, W].
 * workspace: mmu_conv3x3_mfma_workspace_bytes() bytes (prepared bf16 weights), 16-byte aligned. */
typedef struct {
    int32_t batch, in_channels, out_channels, height, width, transposed;
    const void *input;      /* io_dtype */
    const float *weight;
    const float *bias;      /* [out_channels] or NULL */
    void *out;              /* io_dtype */
    void *workspace;
    int32_t io_dtype;       /* ABI 10: MMU_DTYPE_F32 (0, a zeroed struct) or MMU_DTYPE_BF16 -- bf16 activations under
                             * autocast: input and out bfloat16, weight and bias float32, two MFMAs per product (a bf16
                             * value is its own hi part), float32 accumulation.  mmu_conv3x3_wgrad_mfma: x AND dout
                             * bfloat16 (dout 8-byte aligned), one MFMA per product, dweight float32 */
} mmu_conv3x3_mfma_params;

size_t mmu_conv3x3_mfma_workspace_bytes(int in_channels, int out_channels);
int mmu_conv3x3_mfma(const mmu_conv3x3_mfma_params *p, void *stream);

/* The weight gradient of the same convolution (in_channels % 32 == 0, out_channels % 64 == 0, width % 4 == 0), same
 * struct read as: input = x [batch, in_channels, H, W], weight = dout [batch, out_channels, H, W] (16-byte aligned),
 * out = dweight [out_channels][in_channels][3][3]; bias / transposed are ignored.  The contraction runs over pixels: the
 * patch operand is read from LDS with ds_read_b64_tr_b16.  Deterministic (per-workgroup partials added in fixed
 * order).  workspace: mmu_conv3x3_wgrad_mfma_workspace_floats() floats. */
size_t mmu_conv3x3_wgrad_mfma_workspace_floats(int batch, int in_channels, int out_channels, int height, int width);
int mmu_conv3x3_wgrad_mfma(const mmu_conv3x3_mfma_params *p, void *stream);

/* ---- the stride-2 dense convolutions on the bf16 matrix cores with float32 accuracy ----- */
/* src/UM_Net/MMUNet.py:360-375 (RCG: ConvTranspose2d(64, 64, 4, 2, 1) and Conv2d(64, 64, 4, 2, 1)) and :439-452
 * (Conv2d(C, 2C, 3, 2, 1) of the down-sampling ResidualBlocks), their input and weight gradients.  kernel in {3, 4},
 * stride 2, padding 1; float32 NCHW contiguous; in_channels % 16 == 0, out_channels % 64 == 0 (weight gradient: both
 * % 64 == 0, output width % 4 == 0).
 *   mmu_conv_s2_mfma            out [B, out_channels, Ho, Wo] = conv2d(input [B, in_channels, Hi, Wi], weight
 *                               [out_channels][in_channels][K][K], stride 2, padding 1) (+ bias), Ho = (Hi + 2 - K) / 2 + 1.
 *                               Also the input gradient of a transposed convolution: input = its output gradient,
 *                               weight = its [Cin_T][Cout_T][K][K] as it is (in_channels = Cout_T, out_channels = Cin_T).
 *   mmu_conv_s2_transposed_mfma out [B, out_channels, 2 Hi, 2 Wi] = conv_transpose2d(input [B, in_channels, Hi, Wi],
 *                               weight [in_channels][out_channels][K][K], stride 2, padding 1) (+ bias) (K = 3: with
 *                               output_padding 1).  Also the input gradient of a strided convolution: input = its
 *                               output gradient, weight = its [Cout_c][Cin_c][K][K] as it is.
 *   mmu_conv_s2_wgrad_mfma      out = dweight [out_channels][in_channels][K][K] of the strided convolution: input = x,
 *                               weight field = dout [B, out_channels, Ho, Wo] (16-byte aligned).  For a transposed
 *                               convolution's weight [Cin_T][Cout_T]: input = its output gradient (in_channels =
 *                               Cout_T ... the high-resolution tensor), weight field = its input (out_channels = Cin_T).
 * workspace: mmu_conv_s2_workspace_bytes() bytes (prepared bf16 weights), 16-byte aligned /
 *            mmu_conv_s2_wgrad_workspace_floats() floats.  Deterministic (no atomics). */
typedef struct {
    int32_t batch, in_channels, out_channels, in_height, in_width, out_height, out_width, kernel;
    const void *input;      /* io_dtype */
    const void *weight;     /* float32 weight; mmu_conv_s2_wgrad_mfma: dout, io_dtype */
    const float *bias;      /* [out_channels] or NULL */
    void *out;              /* io_dtype; mmu_conv_s2_wgrad_mfma: dweight, float32 */
    void *workspace;
    int32_t io_dtype;       /* ABI 12: MMU_DTYPE_F32 (0, a zeroed struct) or MMU_DTYPE_BF16 -- bfloat16 activations under
                             * autocast: input / out bfloat16 with float32 weights and bias, two MFMAs per product, float32
                             * accumulation (maps within 32-bit byte offsets); weight gradient: x AND dout bfloat16 (dout
                             * 8-byte aligned), one MFMA per product, dweight float32 */
} mmu_conv_s2_params;

size_t mmu_conv_s2_workspace_bytes(int in_channels, int out_channels);
int mmu_conv_s2_mfma(const mmu_conv_s2_params *p, void *stream);
int mmu_conv_s2_transposed_mfma(const mmu_conv_s2_params *p, void *stream);
size_t mmu_conv_s2_wgrad_workspace_floats(int batch, int in_channels, int out_channels, int out_height, int out_width);
int mmu_conv_s2_wgrad_mfma(const mmu_conv_s2_params *p, void *stream);

/* ---- W (rows x inner) times a tokens-last matrix, on the bf16 matrix cores with float32 accuracy ------------- */
/* out[b] = W . X[b] for b < batch;  X[b] = x + b*x_bs, `inner` rows of `tokens` contiguous floats, row stride x_rs;
 * out[b] = out + b*out_bs, `rows` rows, row stride out_rs (strides in elements).  Any rows / inner (ABI 6): the weight
 * image is padded with zeros to 64 rows / 16 columns (x_proj: 36 x 128 and its transpose 128 x 36), rows past `rows` are
 * not stored, rows of X past `inner` are not read.
 * transposed_weight = 0: weight is [rows][inner] with leading dimension w_ld; 1: weight is [inner][rows] (w_ld its
 * leading dimension) and is read transposed.  MMConv's dsc_conv_x on the sampler output (src/UM_Net/MMUNet.py:262):
 * forward = (Cout x 3Cin) . samples, input gradient = transposed weight . dout.
 * workspace: mmu_gemm_tokens_workspace_bytes() bytes, 16-byte aligned.
 * weight = NULL: the workspace ALREADY holds this weight's bf16 image, written by mmu_gemm_tokens_prepare_batch
 * for the same (rows, inner, transposed_weight); the call then launches the product alone.
 * Two kernels behind the entry point (ABI 7): products with >= 520 tiles of 64 rows x 512 tokens stream through the
 * producer / consumer kernel (two-part bf16 split, 2^-16 products); smaller ones -- few tokens, deep inner dimension:
 * the DSC products and projections of the maps <= 64 x 64 -- take a 64-row x 32/64-token kernel whose waves split the
 * inner dimension, with THREE-part (float32-grade) products.  The image holds the hi | mid parts per chunk and the third
 * parts behind them (1.5 x the ABI-6 size). */
typedef struct {
    int32_t rows, inner, tokens, batch, transposed_weight;
    const float *weight;  int64_t w_ld;
    const void *x;        int64_t x_rs, x_bs;
    void *out;            int64_t out_rs, out_bs;
    void *workspace;
    int32_t accumulate;   /* ABI 6: non-zero: out += W . X (x_proj's input gradient lands on the scan's, selective_scan_interface.py:277) */
    int32_t x_dtype, out_dtype;  /* ABI 9: MMU_DTYPE_F32 (0, a zeroed struct) or both MMU_DTYPE_BF16: bf16 activations
                                  * under autocast (selective_scan_interface.py:169-171); x / out are then bf16 pointers,
                                  * strides still in elements, the weight stays float32, accumulation float32 */
} mmu_gemm_tokens_params;

size_t mmu_gemm_tokens_workspace_bytes(int rows, int inner);
int mmu_gemm_tokens_mfma(const mmu_gemm_tokens_params *p, void *stream);
/* The weight images of MANY products in one launch (a model prepares all its DSC weights once per forward pass).
 * table: DEVICE array of n_items rows of six int64 {weight pointer, its leading dimension, image pointer (16-byte
 * aligned, mmu_gemm_tokens_workspace_bytes(rows, inner) bytes), rows, inner, transposed_weight}; max_elements = the
 * largest PADDED rows * inner among them (rows rounded up to 64, inner to 16). */
int mmu_gemm_tokens_prepare_batch(const int64_t *table, int n_items, int64_t max_elements, void *stream);

/* ---- Mamba's dt_proj on tokens-last operands (ABI 6) ------------------------------------------------------------ */
/* forward  (mmu_dt_proj_fwd): delta[d][t] = sum_r weight[d][r] * dt[r][t]      (selective_scan_interface.py:182; the bias
 *                             and softplus are the scan's)
 * backward (mmu_dt_proj_bwd): dt[r][t]    = sum_d weight[d][r] * delta[d][t]   (:274; `delta` holds d delta, `dt`
 *                             receives d dt -- the first `rank` rows of d x_dbl)
 * weight [dim][rank] with leading dimension w_ld; dt `rank` rows, delta `dim` rows of `tokens` contiguous floats with
 * row strides dt_rs / delta_rs (elements).  rank 1..8, tokens % 4 == 0, rows 16-byte aligned.  The weight gradient is
 * mmu_gemm_nt_splitk's. */
typedef struct {
    int32_t rank, dim;
    int64_t tokens;
    const void *dt;       int64_t dt_rs;      /* io_dtype; written by mmu_dt_proj_bwd */
    const float *weight;  int64_t w_ld;
    void *delta;          int64_t delta_rs;   /* io_dtype; read by mmu_dt_proj_bwd */
    int32_t io_dtype;     /* ABI 11: MMU_DTYPE_F32 (0, a zeroed struct) or MMU_DTYPE_BF16 -- bfloat16 token rows (8-byte
                           * aligned), float32 weights and arithmetic, results rounded to nearest even */
} mmu_dt_proj_params;
int mmu_dt_proj_fwd(const mmu_dt_proj_params *p, void *stream);
int mmu_dt_proj_bwd(const mmu_dt_proj_params *p, void *stream);

/* ---- Mamba's x_proj on tokens-last operands (ABI 6) -------------------------------------------------------------- */
/* forward  (mmu_x_proj_fwd): x_dbl[r][t] = sum_d weight[r][d] * x[d][t]          (selective_scan_interface.py:181)
 * backward (mmu_x_proj_bwd): x[d][t]   += sum_r weight[r][d] * x_dbl[r][t]       (:277; `x` holds d conv and is updated
 *                            in place, `x_dbl` holds d x_dbl)
 * weight [rows][dim] with leading dimension w_ld, rows = dt_rank + 2 d_state in {33, 34, 36, 40}, dim % 4 == 0;
 * x `dim` rows, x_dbl `rows` rows of `tokens` contiguous floats (row strides x_rs / x_dbl_rs).  Exact float32 products on
 * the vector pipe (streaming; csrc/dt_proj.hip).  The weight gradient is mmu_gemm_nt_splitk's. */
typedef struct {
    int32_t rows, dim;
    int64_t tokens;
    const void *x;        int64_t x_rs;       /* io_dtype; updated by mmu_x_proj_bwd */
    const float *weight;  int64_t w_ld;
    void *x_dbl;          int64_t x_dbl_rs;   /* io_dtype; read by mmu_x_proj_bwd */
    int32_t io_dtype;     /* ABI 11: as mmu_dt_proj_params.io_dtype */
} mmu_x_proj_params;
int mmu_x_proj_fwd(const mmu_x_proj_params *p, void *stream);
int mmu_x_proj_bwd(const mmu_x_proj_params *p, void *stream);

/* ---- C (m x n) = sum over tokens of A[i][t] * B[j][t]: token-contraction ("NT") product on the fp32 matrix cores - */
/* The weight gradient of every projection applied per token: in_proj / out_proj / x_proj / dt_proj of the Mamba blocks
 * (mamba_ssm/ops/selective_scan_interface.py:272-277,394; mamba_simple.py:201-205,270) and MMConv's K x 1 DSC
 * convolution (src/UM_Net/MMUNet.py:262): dW = G . X^T over batch * seqlen tokens.  Token (b, l) of A's row i is at
 * a + i*a_rs + b*a_bs + l (same for B): channel-major and batch-major operands are both read in place.  float32 in
 * and out, float32 accumulation; products on the bf16 matrix pipe from a hi/lo split of each operand element (three
 * MFMAs per product, ~2^-16 relative per product; HBM-bound), or -- exact_products = 1 -- exact on the fp32 pipe
 * (v_mfma_f32_32x32x2_f32; bound by that pipe).  Split over the token axis with the slab partials added in a fixed
 * order (deterministic).  seqlen % 32 == 0; a, b 16-byte aligned; strides multiples of 4.
 * workspace: mmu_gemm_nt_splitk_workspace_floats() floats. */
typedef struct {
    int32_t m, n, batch, seqlen;
    int32_t exact_products;   /* 1: exact float32 products on the fp32 matrix pipe */
    int32_t narrow_steps;     /* 1: always the 32-token-step kernel (default: 128-token steps when seqlen % 128 == 0) */
    const void *a;   int64_t a_rs, a_bs;   /* ab_dtype; strides in elements */
    const void *b;   int64_t b_rs, b_bs;
    float *c;        /* [m][n] contiguous */
    float *workspace;
    int32_t ab_dtype;         /* ABI 11: MMU_DTYPE_F32 (0, a zeroed struct) or MMU_DTYPE_BF16 -- both operands bfloat16
                               * (autocast): exact products, ONE MFMA each; seqlen % 128 == 0, a / b 8-byte aligned,
                               * exact_products = narrow_steps = 0 */
} mmu_gemm_nt_params;

size_t mmu_gemm_nt_splitk_workspace_floats(int m, int n, int batch, int seqlen);
int mmu_gemm_nt_splitk(const mmu_gemm_nt_params *p, void *stream);

/* ---- CBAM's pooled statistics: mean and max (with arg-max) in one pass, one-pass backward ----------------------------- */
/* src/UM_Net/MMUNet.py:327-333.  float32, contiguous [batch, channels, hw], hw % 4 == 0; ties go to the first maximum.
 *   MMU_STATS_PIXELS  : mean / max / argmax [batch * channels] over the pixels of a (b, c) row (avg_pool, max_pool);
 *                       bwd: dinput = dmean / hw + [p == argmax] dmax
 *   MMU_STATS_CHANNELS: out [batch, 2, hw] = (max, mean) over the channels of a pixel, argmax [batch, hw] (int32);
 *                       bwd: dinput[c] = dout[:, 1] / channels + [c == argmax] dout[:, 0] */
#define MMU_STATS_PIXELS 0
#define MMU_STATS_CHANNELS 1
typedef struct {
    int32_t batch, channels, mode;
    int64_t hw;
    const float *input;    /* fwd */
    float *mean, *max;     /* fwd, MMU_STATS_PIXELS */
    float *out;            /* fwd, MMU_STATS_CHANNELS */
    int32_t *argmax;       /* fwd: written; bwd: read */
    const float *dmean, *dmax;   /* bwd, MMU_STATS_PIXELS */
    const float *dout;     /* bwd, MMU_STATS_CHANNELS */
    float *dinput;         /* bwd */
    const float *dinput_addend;  /* bwd, optional: added to the statistics' gradient (another consumer's gradient of the
                                  * same input; may be dinput itself) */
} mmu_cbam_stats_params;

/* CBAM's channel gate from the pooled vectors: gate = sigmoid(mlp(avg) + mlp(max)), mlp = Conv2d(C, R, 1, bias=False) ->
 * ReLU -> Conv2d(R, C, 1, bias=False) (src/UM_Net/MMUNet.py:319-329), one launch each way.  float32, contiguous;
 * batch * (channels + 4 * hidden) floats must fit 48 KB. */
typedef struct {
    int32_t batch, channels, hidden;
    const float *avg;      /* [batch, channels] */
    const float *max;      /* [batch, channels] */
    const float *w1;       /* [hidden, channels] */
    const float *w2;       /* [channels, hidden] */
    float *gate;           /* [batch, channels]: fwd written, bwd read */
    const float *dgate;    /* bwd */
    float *davg;           /* bwd, optional */
    float *dmax;           /* bwd, optional */
    float *dw1;            /* bwd, optional */
    float *dw2;            /* bwd, optional */
} mmu_cbam_gate_params;

int mmu_cbam_gate_fwd(const mmu_cbam_gate_params *p, void *stream);
int mmu_cbam_gate_bwd(const mmu_cbam_gate_params *p, void *stream);

int mmu_cbam_stats_fwd(const mmu_cbam_stats_params *p, void *stream);
int mmu_cbam_stats_bwd(const mmu_cbam_stats_params *p, void *stream);

/* ---- out = input * gate, gate constant over a channel's pixels or over a pixel's channels; backward in one pass ------- */
/* CBAM's `c_out * x` and `s_out * y1` (src/UM_Net/MMUNet.py:330,336), RCG's gate product (MMUNet.py:415).  float32,
 * contiguous [batch, channels, hw], hw % 4 == 0.  gate / dgate: [batch, channels] (MMU_GATE_CHANNEL) or [batch, hw]
 * (MMU_GATE_SPATIAL).  bwd: dinput = dout * gate (optional), dgate = sum of dout * input over the broadcast axis
 * (optional); no atomics. */
#define MMU_GATE_CHANNEL 0
#define MMU_GATE_SPATIAL 1
typedef struct {
    int32_t batch, channels, mode;
    int64_t hw;
    const float *input;
    const float *gate;
    float *out;            /* fwd */
    const float *dout;     /* bwd */
    float *dinput;         /* bwd, optional */
    float *dgate;          /* bwd, optional */
    /* bwd, MMU_GATE_CHANNEL, optional (both or neither): the gradient that reaches `out` through CBAM's channel
     * statistics of it (mmu_cbam_stats, MMU_STATS_CHANNELS) is added to dout on the fly --
     * dout'[b][c][p] = dout + (stats_dout[b][1][p] / channels + [c == stats_argmax[b][p]] stats_dout[b][0][p]) --
     * instead of being materialised and added by two more passes over the map */
    const float *stats_dout;     /* [batch, 2, hw] */
    const int32_t *stats_argmax; /* [batch, hw] */
    /* MMU_GATE_SPATIAL, optional: a second factor and an addend of the input's shape -- out = input * input2 * gate +
     * addend (RCG's `x0 * gate * x2 + f`, MMUNet.py:415) in one pass; bwd: dinput = dout gate input2,
     * dinput2 = dout gate input, dgate = sum over the channels of dout input input2 (d addend = dout) */
    const float *input2;
    const float *addend;         /* fwd */
    float *dinput2;              /* bwd, optional */
    float *workspace;            /* bwd, MMU_GATE_SPATIAL with dgate: mmu_gated_mul_bwd_workspace_floats() floats (the
                                  * channel range is cut into chunks whose sums meet in a fixed order) */
} mmu_gated_mul_params;

size_t mmu_gated_mul_bwd_workspace_floats(int batch, int channels, int64_t hw, int mode);

int mmu_gated_mul_fwd(const mmu_gated_mul_params *p, void *stream);
int mmu_gated_mul_bwd(const mmu_gated_mul_params *p, void *stream);

/* ---- nn.Conv2d(2, 1, kernel_size=7, padding=3, bias=False): CBAM's spatial-attention convolution -------------------- */
/* src/UM_Net/MMUNet.py:323: float32, contiguous; input [batch, 2, H, W], weight [1, 2, 7, 7], out / dout [batch, 1, H, W].
 * bwd: dinput (optional) and dweight (optional; needs input and workspace: per-workgroup partial rows, ordered sum).
 * workspace: mmu_conv7x7_2to1_workspace_floats() floats. */
typedef struct {
    int32_t batch, height, width;
    const float *input;
    const float *weight;
    float *out;             /* fwd */
    const float *dout;      /* bwd */
    float *dinput;          /* bwd, optional */
    float *dweight;         /* bwd, optional */
    float *workspace;       /* bwd with dweight */
} mmu_conv7x7_params;

size_t mmu_conv7x7_2to1_workspace_floats(int batch, int height, int width);
int mmu_conv7x7_2to1_fwd(const mmu_conv7x7_params *p, void *stream);
int mmu_conv7x7_2to1_bwd(const mmu_conv7x7_params *p, void *stream);

/* ---- backward of nn.MaxPool2d(3, stride=2, padding=1) as a gather (MM_Net's stem pooling) ----------------------- */
/* dinput[b, c, y, x] = sum of dout over the (at most four) windows whose recorded arg-max (indices: int64, flat y*W + x
 * per plane, as F.max_pool2d(..., return_indices=True) returns them) is (y, x).  float32, contiguous NCHW. */
typedef struct {
    int64_t planes;                       /* batch * channels */
    int32_t height, width, out_height, out_width;
    const void *dout;                     /* [planes, out_height, out_width], io_dtype */
    const void *indices;                  /* int64, same shape as dout */
    void *dinput;                         /* [planes, height, width], io_dtype */
    const void *input;                    /* fwd: [planes, height, width], io_dtype */
    void *out;                            /* fwd: [planes, out_height, out_width], io_dtype */
    uint8_t *codes;                       /* fwd (written) / bwd_codes (read): arg-max position 3 * dy + dx inside the window */
    const void *dinput_addend;            /* bwd_codes, optional: added to the gathered gradient (may be dinput itself) */
    int32_t io_dtype;                     /* ABI 12: MMU_DTYPE_F32 (0, a zeroed struct) or, for fwd / bwd_codes on maps with
                                           * width % 8 == 0 and even height, MMU_DTYPE_BF16 (comparisons and sums in float32) */
} mmu_maxpool_params;

int mmu_maxpool3s2_bwd(const mmu_maxpool_params *p, void *stream);
/* The pooling itself with one-byte arg-max codes (first maximum in row-major window order, NaN wins: ATen's choice), and
 * the gather backward from those codes. */
int mmu_maxpool3s2_fwd(const mmu_maxpool_params *p, void *stream);
int mmu_maxpool3s2_bwd_codes(const mmu_maxpool_params *p, void *stream);

/* ---- out = sum of up to four float32 tensors, as float32 or bfloat16 (the state-group sums of a d_state > 16 scan) -- */
/* selective_scan_hip._fwd_groups / _bwd_groups: the per-token outputs of a d_state-64 scan are the sums of its four
 * dstate-16 launches' float32 partial outputs (selective_scan_fwd_kernel.cuh:147-298 computes them in one pass over 64
 * states); contiguous tensors of n elements, 16-byte aligned. */
typedef struct {
    int64_t n;
    int32_t nparts, out_dtype;   /* 1..4; MMU_DTYPE_F32 or MMU_DTYPE_BF16 */
    const float *parts[4];
    void *out;
} mmu_sum_parts_params;

int mmu_sum_parts(const mmu_sum_parts_params *p, void *stream);

/* Bias gradient of a convolution (nn.Conv2d / nn.ConvTranspose2d with bias: RCG's up- / down-sampling, MMUNet.py:360-375):
 * out[c] = sum over (b, h, w) of g[b, c, h, w]; g float32 NCHW contiguous, 16-byte aligned, hw = h * w a multiple of 4;
 * workspace: batch * channels floats.  Ordered sums (reproducible). */
int mmu_channel_sum(const float *g, int batch, int channels, int64_t hw, float *workspace, float *out, void *stream);

/* ---- the training loss: Dice + BCE on sigmoid(logits) (top-level loss.py:5-28) ------------------------------------ */
/* loss = 1 - (2 sum(p t) + smooth) / (sum(p + t) + smooth) + mean BCE(p, t), p = sigmoid(logits), the sums over the whole
 * batch, BCE's logs clamped at -100 as nn.BCELoss does.  fwd: two launches (per-workgroup partial sums, their ordered
 * sum); bwd: one.  float32, contiguous, n elements; no atomics. */
typedef struct {
    int64_t n;
    float smooth;
    const float *logits;   /* [n] */
    const float *targets;  /* [n] */
    float *workspace;      /* fwd: mmu_dice_bce_workspace_floats(n) floats */
    float *out;            /* fwd: written -- out[0] = loss, out[1] = sum(p t), out[2] = sum(p + t); bwd: read */
    const float *dloss;    /* bwd: d loss (one float, on the device) */
    float *dlogits;        /* bwd: [n] */
} mmu_dice_bce_params;

size_t mmu_dice_bce_workspace_floats(int64_t n);
int mmu_dice_bce_fwd(const mmu_dice_bce_params *p, void *stream);
int mmu_dice_bce_bwd(const mmu_dice_bce_params *p, void *stream);

/* Input gradient of `nn.Conv2d(I, O, kernel_size=1, stride=2, bias=False)` (the shortcut of the down-sampling residual
 * blocks, src/UM_Net/MMUNet.py:448) from the gradient `src` [planes, ceil(height/2), ceil(width/2)] of its gathered
 * input: dst [planes, height, width] = src at the even pixels, 0 elsewhere, + addend (optional, [planes, height, width]).
 * addend == dst: in place, only the even pixels are touched. */
int mmu_scatter_stride2(const float *src, float *dst, const float *addend, int64_t planes, int height, int width,
                        void *stream);

/* ---- nn.Conv2d(C, 1, kernel_size=1): one output channel (RCG's gate, the side outputs) --------------------------- */
/* src/UM_Net/MMUNet.py:346,386: out[b, p] = bias + sum_c weight[c] * input[b, c, p] over hw pixels; float32, contiguous
 * NCHW, channels in {16, 64}, hw % 4 == 0.  bwd: dinput[b, c, p] = dout[b, p] weight[c] (optional), dweight [C] and
 * dbias [1] (optional) as ordered sums of per-workgroup partials (deterministic).
 * workspace (bwd): mmu_conv1x1_one_workspace_floats() floats. */
typedef struct {
    int32_t batch, channels;
    int64_t hw;
    const float *input;    /* [batch, channels, hw] */
    const float *weight;   /* [channels] */
    const float *bias;     /* [1] or NULL (fwd) */
    float *out;            /* fwd: [batch, hw] */
    const float *dout;     /* bwd: [batch, hw] */
    float *dinput;         /* bwd, optional */
    float *dweight;        /* bwd, optional */
    float *dbias;          /* bwd, optional */
    float *workspace;      /* bwd */
    const float *scale;    /* optional: [batch, channels] factors on the input (the side outputs' Dropout2d in front of the
                            * convolution, MMUNet.py:345-350: mask / (1 - p)) -- out = sum_c w[c] scale[b][c] x[b][c] */
} mmu_conv1x1_one_params;

size_t mmu_conv1x1_one_workspace_floats(int batch, int channels, long hw);
int mmu_conv1x1_one_fwd(const mmu_conv1x1_one_params *p, void *stream);
int mmu_conv1x1_one_bwd(const mmu_conv1x1_one_params *p, void *stream);

/* ---- conv1d + SiLU + x_proj + dt_proj of a small Mamba block in one kernel (a5/a6 glue, MMConv's blocks) ---- */
/* mamba_ssm/ops/selective_scan_interface.py:173-210 for inner width dim in {2, 6}, conv width 4, dt_rank 1,
 * float32:  conv_out = silu(causal_conv1d(x)),  x_dbl[j] = sum_d x_proj_weight[j][d] conv_out[d]  (rows = dt_rank
 * + 2*dstate; row 0 = dt),  delta[d] = dt_proj_weight[d] * x_dbl[0].  x / conv_out / delta: [batch, dim, L] with
 * unit L stride and the strides given (multiples of 4, 16-byte aligned bases); x_dbl: [rows, batch*L]
 * contiguous, or NULL to skip it (backward recomputation). */
typedef struct {
    int32_t batch, dim, seqlen, rows;
    const float *x;               int64_t x_bs, x_ds;
    const float *conv_weight;     /* [dim, 4] */
    const float *conv_bias;       /* [dim] or NULL */
    const float *x_proj_weight;   /* [rows, dim] */
    const float *dt_proj_weight;  /* [dim] (dt_rank 1) */
    float *conv_out;              int64_t conv_bs, conv_ds;
    float *x_dbl;                 /* [rows, batch*seqlen] or NULL */
    float *delta;                 int64_t delta_bs, delta_ds;
} mmu_mamba_pre_params;

int mmu_mamba_pre_small(const mmu_mamba_pre_params *p, void *stream);

/* The backward mirror (selective_scan_interface.py:268-277), inner width 2 or 6, rows = 33 (dt_rank 1, d_state 16),
 * float32, every [dim]-row tensor laid out [dim][tokens] (tokens = batch*L, a multiple of 4), 16-byte aligned:
 *   d dt = dt_proj_weight^T ddelta (kept in registers);  ddt_proj_weight[d] = sum_t ddelta[d][t] dt[t];
 *   dx_proj_weight[j][d] = sum_t dx_dbl[j][t] conv_out[d][t]  (row 0 of dx_dbl = d dt, rows 1.. read from dx_dbl);
 *   dconv_out[d][t] += sum_j x_proj_weight[j][d] dx_dbl[j][t]   (in place). */
typedef struct {
    int32_t dim, rows;
    int64_t tokens;
    const float *ddelta;          /* [dim][tokens] */
    const float *dt;              /* [tokens] = row 0 of x_dbl */
    const float *dx_dbl;          /* [rows][tokens]; row 0 is ignored */
    const float *conv_out;        /* [dim][tokens] */
    float *dconv_out;             /* [dim][tokens], in/out */
    const float *x_proj_weight;   /* [rows][dim] */
    const float *dt_proj_weight;  /* [dim] */
    float *dx_proj_weight;        /* [rows][dim] out */
    float *ddt_proj_weight;       /* [dim] out */
    float *workspace;             /* mmu_mamba_post_small_workspace_floats() floats */
} mmu_mamba_post_params;

size_t mmu_mamba_post_small_workspace_floats(int dim, int rows, long tokens);
int mmu_mamba_post_small(const mmu_mamba_post_params *p, void *stream);

/* ---- MMConv glue around its K-channel Mamba, fused ----------------- */
/* Replaces ~30 tiny PyTorch kernels per MMConv block and direction (src/UM_Net/MMUNet.py:122-193 +
 * requirements/mamba_simple.py:201-205,365): zig-zag token flatten + in_proj (A), and out_proj + inverse
 * zig-zag + coordinate arithmetic y = max(softplus(altho), .01) * seq + row + scope * cumsum-from-centre (B).
 * taps K in {1, 3}; d_inner = 2K; all tensors contiguous float32; L = height * width.
 *   A fwd : offset [B, 2K, H, W], in_proj_weight [4K, K]            -> xz [4K][B][L]  (tokens-last)
 *   A bwd : dxz [4K][B][L]                                          -> doffset [B, 2K, H, W], din_proj_weight
 *   B fwd : offset, out_z [2K][B][L], out_proj_weight [K, 2K], altho -> y [B, K, H, W]  (row coordinates)
 *   B bwd : dy [B, K, H, W]                        -> dout_z [2K][B][L], dout_proj_weight, daltho, doffset
 * Weight / altho gradients are zeroed inside and accumulated with one float atomic per block and entry. */
typedef struct {
    int32_t batch, height, width, taps;
    float extend_scope;
    const float *offset;
    const float *in_proj_weight;
    const float *out_proj_weight;
    const float *altho;
    float *xz;
    const float *dxz;
    const float *out_z;
    float *y;
    const float *dy;
    float *doffset;
    float *din_proj_weight;
    float *dout_z;
    float *dout_proj_weight;
    float *daltho;
    float *workspace;   /* bwd, optional: mmu_coords_bwd_workspace_floats() floats.  With it the weight gradients are per-block
                         * partials added in a fixed order (reproducible, nothing to zero, deferrable: mmu_deferred_*);
                         * without it they are float atomics into zero-filled targets */
    int32_t accumulate_doffset;   /* mmu_zigzag_inproj_bwd: doffset already holds mmu_coords_outproj_bwd's d offset of the
                                   * same offsets: add to it (autograd then has one gradient to route, not two to add) */
} mmu_coords_params;

size_t mmu_coords_bwd_workspace_floats(int batch, int height, int width, int taps);

int mmu_zigzag_inproj_fwd(const mmu_coords_params *p, void *stream);
int mmu_zigzag_inproj_bwd(const mmu_coords_params *p, void *stream);
int mmu_coords_outproj_fwd(const mmu_coords_params *p, void *stream);
int mmu_coords_outproj_bwd(const mmu_coords_params *p, void *stream);

/* ---- the whole K-channel Mamba chain of an MMConv block on a small map, one kernel each way (rows f1 + f2) ---- */
/* Replaces, for power-of-two maps of 64 .. 1,024 tokens (MM-UNet's 16 x 16 and 32 x 32 maps), the six forward
 * launches zigzag_inproj -> mamba_pre_small -> selective scan (3 kernels) -> coords_outproj and their eleven backward
 * launches:
 *   src/UM_Net/MMUNet.py:176-188 (zig-zag flatten, self.mamba, inverse zig-zag, coordinate arithmetic),
 *   requirements/mamba_simple.py:201-205,303-318,365 (in_proj, uni-directional branch, out_proj),
 *   mamba_ssm/ops/selective_scan_interface.py:173-215 (MambaInnerFn.forward) and :238-289,387-394 (backward),
 *   csrc/selective_scan/selective_scan_{fwd,bwd}_kernel.cuh, causal-conv1d/csrc/causal_conv1d_{fwd,bwd}.cu.
 * `parts` workgroups per batch item, each scanning d_state / parts of the states (everything downstream of the scan is
 * a sum over the states); inside a workgroup a wave owns a channel, a lane a run of L / 64 tokens.  taps K in {1, 3},
 * inner width 2K, conv width 4, dt_rank 1, d_state <= 64, float32, every tensor contiguous.
 *   fwd: offset [B, 2K, H, W] (only channels 0..K-1 are read) -> y [parts][B, K, H, W]: partial row maps whose SUM is
 *        the row coordinate map (mmu_morph_sample_fwd adds them while it reads them: mmu_morph_params.y_parts).
 *        Nothing is saved for the backward (it recomputes the states by the same scan).
 *   bwd: dy [B, K, H, W] (gradient of the summed map) -> doffset [B, 2K, H, W] (channels K..2K-1 = 0) and
 *        dweights (mmu_mamba_small_grad_floats() floats: in_proj [4K][K] | conv weight [2K][4] | conv bias [2K] |
 *        x_proj [1+2N][2K] | dt_proj [2K] | dt bias [2K] | A [2K][N] | D [2K] | out_proj [K][2K] | altho);
 *        workspace: mmu_mamba_small_bwd_workspace_floats() floats (per-workgroup partials, summed in fixed order by a
 *        second small kernel: deterministic, no atomics, nothing to zero). */
typedef struct {
    int32_t batch, height, width, taps, dstate;
    int32_t parts;                 /* state-range parts per batch item, divides dstate (mmu_mamba_small_parts()) */
    float extend_scope;
    const float *offset;
    const float *in_proj_weight;   /* [4K][K] */
    const float *conv_weight;      /* [2K][4] */
    const float *conv_bias;        /* [2K] or NULL */
    const float *x_proj_weight;    /* [1 + 2N][2K] */
    const float *dt_proj_weight;   /* [2K] */
    const float *dt_bias;          /* [2K] or NULL */
    const float *A;                /* [2K][N], = -exp(A_log) */
    const float *D;                /* [2K] or NULL */
    const float *out_proj_weight;  /* [K][2K] */
    const float *altho;            /* scalar */
    float *y;
    const float *dy;
    float *doffset;
    float *workspace;
    float *dweights;
} mmu_mamba_small_params;

int mmu_mamba_small_supported(int taps, int height, int width, int dstate);
/* parts for the forward (backward = 0) or the backward call (backward = 1); the two need not agree */
int mmu_mamba_small_parts(int batch, int taps, int height, int width, int dstate, int backward);
size_t mmu_mamba_small_bwd_workspace_floats(int batch, int taps, int height, int width, int dstate, int parts);
size_t mmu_mamba_small_grad_floats(int taps, int dstate);
int mmu_mamba_small_fwd(const mmu_mamba_small_params *p, void *stream);
int mmu_mamba_small_bwd(const mmu_mamba_small_params *p, void *stream);

/* ---- test hooks (exercise the wave-level primitives on the GPU) -------- */
/* Runs the in-wave affine-pair scan on n_waves*64 (P,S) pairs, one wave per 64.
 * reverse=0: forward inclusive; reverse=1: reverse inclusive.  variant 0 = DPP intrinsics,
 * 1 = shuffle, 2 = hand-written fused DPP, 3/4 = second/first of two interleaved fused scans
 * (the first one runs on (0.5*P, -S)). */
int mmu_debug_wave_scan(const float *P, const float *S, float *outP, float *outS, int n_waves,
                        int reverse, int variant, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MMUNET_AMD_H */
