// causal_conv1d.hip -- depthwise causal conv1d (+SiLU) for gfx950, fwd / bwd / single-step update.
//
// Replaces the reference's causal_conv1d_cuda extension
//   fwd    : requirements/Mamba/causal-conv1d/csrc/causal_conv1d_fwd.cu:39-158
//   bwd    : requirements/Mamba/causal-conv1d/csrc/causal_conv1d_bwd.cu:46-270
//   update : requirements/Mamba/causal-conv1d/csrc/causal_conv1d_update.cu:26-96
// The reference walks L serially inside one block per (batch, channel); here the grid is
// (L tiles, channel, batch) so a 6-channel, 65,536-token call still fills the chip.  Pure
// streaming: each lane owns 4 consecutive tokens (16-B loads/stores), the 3-token causal halo
// comes from the neighbouring 16-B group (an L1/L2 hit), fp32 accumulate.
//
//   p_t  = bias + sum_w W[w] x[t-(width-1-w)]          out_t = silu ? p_t*sigmoid(p_t) : p_t
//   dp_t = dout_t * (silu ? sig(p)(1 + p(1-sig(p))) : 1)
//   dx_t = sum_w W[w] dp[t+width-1-w] ;  dW[w] += sum_{b,t} x[t-(width-1-w)] dp_t ;  db += sum dp_t
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

constexpr int CK = 4;                // tokens per lane
constexpr int CTHREADS = 256;        // threads per block
constexpr int CTILE = CK * CTHREADS; // tokens per block

struct ConvArgs {
    int batch, dim, seqlen, width, silu, vec;
    int rows_batch_fastest;   // grid (tiles, batch, dim) instead of (tiles, dim, batch): see conv1d_row_order
    const void *x, *dout;
    const float *weight, *bias;
    void *out, *dx;
    float *dweight, *dbias, *ws;   // ws: [batch][dim][gridDim.x][5] partials of (dW taps, db), or NULL -> float atomics
    long x_bs, x_ds, out_bs, out_ds, dout_bs, dout_ds, dx_bs, dx_ds, w_ds, w_ws;
};

__device__ __forceinline__ void load_weights(const ConvArgs &p, int d, float (&wr)[4], float &bv) {
    // right-aligned taps: wr[4-width+k] = W[k]
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int k = m - (4 - p.width);
        wr[m] = k >= 0 ? p.weight[(long)d * p.w_ds + (long)k * p.w_ws] : 0.f;
    }
    bv = p.bias ? p.bias[d] : 0.f;
}

template <typename io_t>
__global__ __launch_bounds__(CTHREADS) void conv1d_fwd_kernel(ConvArgs p) {
    const int d = p.rows_batch_fastest ? blockIdx.z : blockIdx.y, b = p.rows_batch_fastest ? blockIdx.y : blockIdx.z;
    const int t = blockIdx.x * CTILE + threadIdx.x * CK;
    const int L = p.seqlen;
    if (t >= L) return;
    float wr[4], bv;
    load_weights(p, d, wr, bv);
    const io_t *xr = (const io_t *)p.x + (long)b * p.x_bs + (long)d * p.x_ds;
    float cur[CK], prev[CK];
    load_k<io_t, CK>(xr + t, L - t, p.vec, cur);
    if (t >= CK) {
        load_k<io_t, CK>(xr + t - CK, CK, p.vec, prev);
    } else {
#pragma unroll
        for (int i = 0; i < CK; ++i) prev[i] = 0.f;
    }
    float xs[7] = {prev[1], prev[2], prev[3], cur[0], cur[1], cur[2], cur[3]};
    float o[CK];
#pragma unroll
    for (int i = 0; i < CK; ++i) {
        float acc = bv;
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = fmaf(wr[m], xs[i + m], acc);
        o[i] = p.silu ? acc * sigmoidf_(acc) : acc;
    }
    store_k<io_t, CK>((io_t *)p.out + (long)b * p.out_bs + (long)d * p.out_ds + t, L - t, p.vec, o);
}

// A block walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... of one (batch, channel) row and keeps the five
// weight-gradient sums (four taps + bias) in registers across its tiles: one block reduction per block, not per
// 1,024 tokens.  With a workspace the block sums are plain stores and conv1d_wgrad_reduce_kernel adds them in a
// fixed order (bit-reproducible, nothing to zero-fill); without one they are float atomics, as in the reference
// (causal_conv1d_bwd.cu:256-268).
template <typename io_t>
__global__ __launch_bounds__(CTHREADS) void conv1d_bwd_kernel(ConvArgs p) {
    __shared__ float red[CTHREADS / 64][5];
    const int d = p.rows_batch_fastest ? blockIdx.z : blockIdx.y, b = p.rows_batch_fastest ? blockIdx.y : blockIdx.z;
    const int L = p.seqlen;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float wr[4], bv;
    load_weights(p, d, wr, bv);
    float dwr[4] = {0.f, 0.f, 0.f, 0.f}, dbv = 0.f;
    const io_t *xr = (const io_t *)p.x + (long)b * p.x_bs + (long)d * p.x_ds;
    const io_t *gr = (const io_t *)p.dout + (long)b * p.dout_bs + (long)d * p.dout_ds;
    for (int t = blockIdx.x * CTILE + threadIdx.x * CK; t < L; t += gridDim.x * CTILE) {
        float cur[CK], prev[CK], next[CK], g0[CK], g1[CK];
        load_k<io_t, CK>(xr + t, L - t, p.vec, cur);
        load_k<io_t, CK>(xr + t + CK, L - t - CK, p.vec, next);
        if (t >= CK) {
            load_k<io_t, CK>(xr + t - CK, CK, p.vec, prev);
        } else {
#pragma unroll
            for (int i = 0; i < CK; ++i) prev[i] = 0.f;
        }
        load_k<io_t, CK>(gr + t, L - t, p.vec, g0);
        load_k<io_t, CK>(gr + t + CK, L - t - CK, p.vec, g1);
        // xs[j] = x[t-3+j], j = 0..9 ; go[j] = dout[t+j], j = 0..6
        float xs[10] = {prev[1], prev[2], prev[3], cur[0], cur[1], cur[2], cur[3], next[0], next[1], next[2]};
        float dp[7] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2]};
        if (p.silu) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                float acc = bv;
#pragma unroll
                for (int m = 0; m < 4; ++m) acc = fmaf(wr[m], xs[j + m], acc);
                const float sg = sigmoidf_(acc);
                dp[j] *= sg * (1.f + acc * (1.f - sg));
            }
        }
        float dxv[CK];
#pragma unroll
        for (int i = 0; i < CK; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = fmaf(wr[m], dp[i + 3 - m], acc);
            dxv[i] = acc;
            dbv += dp[i];
#pragma unroll
            for (int m = 0; m < 4; ++m) dwr[m] = fmaf(xs[i + m], dp[i], dwr[m]);
        }
        store_k<io_t, CK>((io_t *)p.dx + (long)b * p.dx_bs + (long)d * p.dx_ds + t, L - t, p.vec, dxv);
    }
    // block reduction of the 5 weight-gradient partials
    const float v4 = wave_sum4(dwr[0], dwr[1], dwr[2], dwr[3]);   // lanes 12..15: totals of taps 0..3
    const float vb = wave_sum(dbv);
    if (lane >= 12 && lane < 16) red[w][lane - 12] = v4;
    if (lane == 0) red[w][4] = vb;
    __syncthreads();
    if (threadIdx.x < 5) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CTHREADS / 64; ++i) s += red[i][threadIdx.x];
        const int m = threadIdx.x;
        if (p.ws) {
            p.ws[((((long)b * p.dim + d) * gridDim.x) + blockIdx.x) * 5 + m] = s;
        } else if (m < 4) {
            const int k = m - (4 - p.width);
            if (k >= 0) atomicAdd(&p.dweight[(long)d * p.width + k], s);
        } else if (p.dbias) {
            atomicAdd(&p.dbias[d], s);
        }
    }
}

// dW[d][k], db[d] = sum over (batch, block) of the partials, in a fixed order.  grid dim, block 64.
__global__ __launch_bounds__(64) void conv1d_wgrad_reduce_kernel(const float *__restrict__ ws, int batch, int dim, int nblk,
                                                                 int width, float *dweight, float *dbias) {
    const int d = blockIdx.x, lane = threadIdx.x;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int n = batch * nblk;           // partial rows of channel d: (b, block) pairs, strided over the lanes
    for (int i = lane; i < n; i += 64) {
        const int b = i / nblk, k = i - b * nblk;
        const float *r = ws + ((((long)b * dim + d) * nblk) + k) * 5;
#pragma unroll
        for (int m = 0; m < 5; ++m) acc[m] += r[m];
    }
    const float v4 = wave_sum4(acc[0], acc[1], acc[2], acc[3]);
    const float vb = wave_sum(acc[4]);
    if (lane >= 12 && lane < 16) {
        const int k = (lane - 12) - (4 - width);
        if (k >= 0) dweight[(long)d * width + k] = v4;
    }
    if (lane == 0 && dbias) dbias[d] = vb;
}

struct UpdArgs {
    int batch, dim, width, silu;
    const void *x;
    void *conv_state, *out;
    const float *weight, *bias;
    long x_bs, x_ds, cs_bs, cs_ds, cs_ws, out_bs, out_ds, w_ds, w_ws;
};

template <typename io_t>
__global__ __launch_bounds__(64) void conv1d_update_kernel(UpdArgs p) {
    const int d = blockIdx.y * 64 + threadIdx.x, b = blockIdx.x;
    if (d >= p.dim) return;
    io_t *cs = (io_t *)p.conv_state + (long)b * p.cs_bs + (long)d * p.cs_ds;
    const float xv = to_f32(((const io_t *)p.x)[(long)b * p.x_bs + (long)d * p.x_ds]);
    float acc = p.bias ? p.bias[d] : 0.f;
    // roll the window left by one, append x (causal_conv1d_update.cu:48-60)
    for (int k = 0; k < p.width; ++k) {
        const float sv = k + 1 < p.width ? to_f32(cs[(long)(k + 1) * p.cs_ws]) : xv;
        cs[(long)k * p.cs_ws] = from_f32<io_t>(sv);
        acc = fmaf(p.weight[(long)d * p.w_ds + (long)k * p.w_ws], sv, acc);
    }
    if (p.silu) acc = acc * sigmoidf_(acc);
    ((io_t *)p.out)[(long)b * p.out_bs + (long)d * p.out_ds] = from_f32<io_t>(acc);
}

inline bool al(const void *p, size_t a) { return p == nullptr || ((uintptr_t)p % a) == 0; }
// blocks per (batch, channel) row of the backward: enough blocks overall to fill the chip several times over,
// few enough that a block keeps its weight-gradient sums in registers over many tiles
inline int conv1d_bwd_blocks(int batch, int dim, int seqlen) {
    const long tiles = ((long)seqlen + CTILE - 1) / CTILE, rows = (long)batch * dim;
    long per_row = (4096 + rows - 1) / rows;
    if (per_row < 1) per_row = 1;
    return (int)(per_row < tiles ? per_row : tiles);
}
inline bool ml(long v) { return (v % CK) == 0; }

}  // namespace

// Workgroups are dispatched x-fastest, then y, then z: the rows that are in flight together should be NEIGHBOURS in
// memory.  mamba_inner hands the kernels channel-major storage ([D][B][L]: batch stride < channel stride); with the
// channel on y the concurrent rows were B * L * 4 bytes = 2 MiB apart (a power of two: the same HBM channels over and
// over) and the forward / backward ran at 57 % / 45 % of HBM instead of 64 % / 60 % on batch-major storage.
static inline int conv1d_row_order(long x_bs, long x_ds, int batch, int dim) { return x_bs < x_ds && batch > 1 && dim > 1; }

#define CONV_CHECKS(p, name)                                                                                     \
    MMU_CHECK((p) != nullptr, name ": null params");                                                             \
    MMU_CHECK((p)->dtype == MMU_DTYPE_F32 || (p)->dtype == MMU_DTYPE_BF16, name ": unsupported dtype %d",        \
              (p)->dtype);                                                                                       \
    MMU_CHECK((p)->width >= 2 && (p)->width <= 4, "causal_conv1d only supports width between 2 and 4 (got %d)",  \
              (p)->width);                                                                                       \
    MMU_CHECK((p)->batch > 0 && (p)->dim > 0, name ": empty tensor");                                            \
    MMU_CHECK((p)->batch <= 65535 && (p)->dim <= 65535, name ": batch/dim too large for the launch grid");

extern "C" int mmu_causal_conv1d_fwd(const mmu_conv1d_fwd_params *p, void *stream) {
    CONV_CHECKS(p, "causal_conv1d_fwd");
    MMU_CHECK(p->seqlen > 0 && p->x && p->weight && p->out, "causal_conv1d_fwd: x, weight, out are required");
    ConvArgs a = {};
    a.batch = p->batch; a.dim = p->dim; a.seqlen = p->seqlen; a.width = p->width; a.silu = p->silu;
    a.x = p->x; a.weight = p->weight; a.bias = p->bias; a.out = p->out;
    a.x_bs = p->x_bs; a.x_ds = p->x_ds; a.out_bs = p->out_bs; a.out_ds = p->out_ds; a.w_ds = p->w_ds; a.w_ws = p->w_ws;
    const size_t g = (p->dtype == MMU_DTYPE_F32 ? 4 : 2) * CK;
    a.vec = al(p->x, g) && al(p->out, g) && ml(p->x_bs) && ml(p->x_ds) && ml(p->out_bs) && ml(p->out_ds);
    a.rows_batch_fastest = conv1d_row_order(p->x_bs, p->x_ds, p->batch, p->dim);
    dim3 grid((p->seqlen + CTILE - 1) / CTILE, a.rows_batch_fastest ? p->batch : p->dim,
              a.rows_batch_fastest ? p->dim : p->batch);
    if (p->dtype == MMU_DTYPE_F32)
        conv1d_fwd_kernel<float><<<grid, CTHREADS, 0, (hipStream_t)stream>>>(a);
    else
        conv1d_fwd_kernel<bf16_t><<<grid, CTHREADS, 0, (hipStream_t)stream>>>(a);
    MMU_HIP_LAUNCH_CHECK("causal_conv1d_fwd");
    return 0;
}

extern "C" size_t mmu_causal_conv1d_bwd_workspace_floats(int batch, int dim, int seqlen) {
    return (size_t)batch * dim * conv1d_bwd_blocks(batch, dim, seqlen) * 5;
}

extern "C" int mmu_causal_conv1d_bwd(const mmu_conv1d_bwd_params *p, void *stream) {
    CONV_CHECKS(p, "causal_conv1d_bwd");
    MMU_CHECK(p->seqlen > 0 && p->x && p->weight && p->dout && p->dx && p->dweight,
              "causal_conv1d_bwd: x, weight, dout, dx, dweight are required");
    ConvArgs a = {};
    a.batch = p->batch; a.dim = p->dim; a.seqlen = p->seqlen; a.width = p->width; a.silu = p->silu;
    a.x = p->x; a.weight = p->weight; a.bias = p->bias; a.dout = p->dout; a.dx = p->dx;
    a.dweight = p->dweight; a.dbias = p->dbias; a.ws = p->workspace;
    a.x_bs = p->x_bs; a.x_ds = p->x_ds; a.dout_bs = p->dout_bs; a.dout_ds = p->dout_ds;
    a.dx_bs = p->dx_bs; a.dx_ds = p->dx_ds; a.w_ds = p->w_ds; a.w_ws = p->w_ws;
    const size_t g = (p->dtype == MMU_DTYPE_F32 ? 4 : 2) * CK;
    a.vec = al(p->x, g) && al(p->dout, g) && al(p->dx, g) && ml(p->x_bs) && ml(p->x_ds) && ml(p->dout_bs) &&
            ml(p->dout_ds) && ml(p->dx_bs) && ml(p->dx_ds);
    a.rows_batch_fastest = conv1d_row_order(p->x_bs, p->x_ds, p->batch, p->dim);
    dim3 grid(conv1d_bwd_blocks(p->batch, p->dim, p->seqlen), a.rows_batch_fastest ? p->batch : p->dim,
              a.rows_batch_fastest ? p->dim : p->batch);
    if (p->dtype == MMU_DTYPE_F32)
        conv1d_bwd_kernel<float><<<grid, CTHREADS, 0, (hipStream_t)stream>>>(a);
    else
        conv1d_bwd_kernel<bf16_t><<<grid, CTHREADS, 0, (hipStream_t)stream>>>(a);
    MMU_HIP_LAUNCH_CHECK("causal_conv1d_bwd");
    if (p->workspace) {
        const long job[8] = {2, (long)p->workspace, (long)p->dweight, (long)p->dbias, p->batch, p->dim, (long)grid.x, p->width};
        if (mmu_defer_job(job)) return 0;   // (deferred_reduce.hip)
        conv1d_wgrad_reduce_kernel<<<p->dim, 64, 0, (hipStream_t)stream>>>(p->workspace, p->batch, p->dim, (int)grid.x,
                                                                         p->width, p->dweight, p->dbias);
        MMU_HIP_LAUNCH_CHECK("causal_conv1d_bwd(reduce)");
    }
    return 0;
}

extern "C" int mmu_causal_conv1d_update(const mmu_conv1d_update_params *p, void *stream) {
    CONV_CHECKS(p, "causal_conv1d_update");
    MMU_CHECK(p->x && p->conv_state && p->weight && p->out, "causal_conv1d_update: x, conv_state, weight, out required");
    UpdArgs a = {};
    a.batch = p->batch; a.dim = p->dim; a.width = p->width; a.silu = p->silu;
    a.x = p->x; a.conv_state = p->conv_state; a.out = p->out; a.weight = p->weight; a.bias = p->bias;
    a.x_bs = p->x_bs; a.x_ds = p->x_ds; a.cs_bs = p->cs_bs; a.cs_ds = p->cs_ds; a.cs_ws = p->cs_ws;
    a.out_bs = p->out_bs; a.out_ds = p->out_ds; a.w_ds = p->w_ds; a.w_ws = p->w_ws;
    dim3 grid(p->batch, (p->dim + 63) / 64);
    if (p->dtype == MMU_DTYPE_F32)
        conv1d_update_kernel<float><<<grid, 64, 0, (hipStream_t)stream>>>(a);
    else
        conv1d_update_kernel<bf16_t><<<grid, 64, 0, (hipStream_t)stream>>>(a);
    MMU_HIP_LAUNCH_CHECK("causal_conv1d_update");
    return 0;
}
