// mamba_small_fused.hip -- the whole K-channel Mamba chain of an MMConv block on a SMALL map as ONE forward and ONE
// backward kernel (gfx950).  SURVEY.md section 8 rows f1 + f2 where they meet.
//
// Reference (per MMConv block, src/UM_Net/MMUNet.py:176-188 with requirements/mamba_simple.py:201-205,303-318,365 and
// mamba_ssm/ops/selective_scan_interface.py:173-215 forward, :238-289 + :387-394 backward):
//     y_off = offset[:, :K] -> two-row zig-zag flatten -> in_proj (K -> 4K) -> x | z
//     u     = silu(causal_conv1d(x, width 4))            x_dbl = W_x u  (rows: dt | B[N] | C[N], dt_rank 1)
//     delta = softplus(W_dt dt + dt_bias)                h_t = exp(delta A) h_{t-1} + delta u B ;  y = C.h + D u
//     out_z = y silu(z) -> out_proj (2K -> K) -> inverse zig-zag
//     rows  = max(softplus(altho), .01) * seq + h + extend_scope * cumsum-from-centre(y_off)
// 24 of MM-UNet's 38 three-tap blocks (and two of the three one-tap blocks) run on 16 x 16 or 32 x 32 maps: L = 256 or
// 1,024 tokens, 6 channels, 16 states.  As separate kernels (zigzag_inproj, mamba_pre_small, chunk_reduce8, chunk_carry,
// chunk_apply_fwd8, coords_outproj forward; eleven launches backward) every one of them is 1-8 workgroups at the
// 4.6 us dependent-node floor of a replayed graph: ~44 us forward and ~90 us backward per block for microseconds of
// arithmetic ("What's weak" 7).
//
// Here ONE workgroup owns ONE batch item (a scan has no parallelism across its tokens that is worth a second
// workgroup at L <= 2,048: the cross-workgroup hand-off costs more than the work).  A lane owns T consecutive
// zig-zag tokens, a wave 64 T of them, nw = L / (64 T) <= 8 waves the whole sequence; everything between the offset map
// and the row coordinates stays in registers:
//   * stage y_off in zig-zag order in LDS (coalesced reads along the rows); in_proj, conv1d (+3 halo tokens from LDS),
//     SiLU, the dt row of x_proj, dt_proj + softplus per lane;
//   * scan: for every state n the B_n / C_n rows are formed from u on the fly (12 FMAs per token: they never exist as
//     tensors), for every channel pair the lane composes its T tokens, wave_scan_affine_x2 (DPP) scans the 64 lanes,
//     and the carry between WAVES travels through LDS as a systolic chain: wave w spins on a progress word of wave
//     w - 1, adds its own aggregate and publishes -- no workgroup barrier inside the 96 (channel, state) steps, a wave
//     lags its predecessor by one LDS round trip in total, not per step;
//   * gate, out_proj, inverse zig-zag and the coordinate arithmetic in the epilogue.
// Forward saves the state entering every lane's token group (`hstate`, [B][2K N][L / T]); the backward kernel maps its
// lanes to the token groups in REVERSE order, so the adjoint recurrence g_t = C_t dy_t + a_{t+1} g_{t+1} is again a
// forward scan over lanes and waves (same DPP scan, same systolic chain), recomputes h from `hstate`, and carries the
// chain rule through x_proj / dt_proj / conv1d (neighbour tokens through LDS) / in_proj / the coordinate terms to
// d offset.  Weight gradients: lane sums -> wave_sum4 -> per-wave LDS slots -> one partial vector per batch item;
// mamba_small_reduce_kernel adds the batch items in fixed order (deterministic, no atomics, no zero fill).
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

// Diagnostic build only (-DMMU_SMALL_STAMPS, tools/dbg/small_stamps.sh): s_memtime at the phase boundaries of every wave of
// workgroup (0, 0), read back through mmu_debug_small_stamps.  In the product build no stamp executes.
#ifdef MMU_SMALL_STAMPS
__device__ unsigned long long g_small_stamps[2 * 8 * 16];   // [fwd / bwd][wave][slot]
#define SMALL_STAMP(dir, slot)                                                                                   \
    do {                                                                                                         \
        if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0)                                       \
            g_small_stamps[((dir) * 8 + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_amdgcn_s_memtime();       \
    } while (0)
#else
#define SMALL_STAMP(dir, slot)
#endif

namespace {

struct SmallArgs {
    int B, H, W, L, N, nw;
    int ns, npp;         // state-range parts per batch item (grid.y) and states per part (N = ns * npp)
    float scope;
    const float *off;    // [B, 2K, H, W]
    const float *win;    // [4K][K]
    const float *cw;     // [2K][4]
    const float *cb;     // [2K] or null
    const float *wx;     // [1 + 2N][2K]
    const float *wdt;    // [2K]
    const float *dtb;    // [2K] or null
    const float *A;      // [2K][N]  (= -exp(A_log))
    const float *Dp;     // [2K] or null
    const float *wout;   // [K][2K]
    const float *altho;  // scalar
    float *y;            // [ns][B, K, H, W]  partial row maps (their sum is the row map)
    float *hstate;       // [B][2K*N][L/T] or null (forward: written; backward: read)
    const float *dy;     // [B, K, H, W]
    float *doff;         // [ns][B][K][H*W]  partial d offset (first K channels), pixel order
    float *part;         // [B * ns][NV] weight-gradient partials
};

// The weights are separate __restrict__ kernel parameters, not members of SmallArgs: only then can the compiler prove
// that the kernel's own stores (hstate, y, d offset) do not clobber them and read them with scalar loads into SGPRs
// (as struct members they became per-lane global_load_dword in the scan loop, each behind the stores' vmcnt).
#define W_PARAMS                                                                                                      \
    const float *__restrict__ win, const float *__restrict__ cw, const float *__restrict__ cb,                       \
        const float *__restrict__ wx, const float *__restrict__ wdt, const float *__restrict__ dtb,                  \
        const float *__restrict__ Aw, const float *__restrict__ Dw, const float *__restrict__ wout,                  \
        const float *__restrict__ altho
#define W_ARGS win, cw, cb, wx, wdt, dtb, Aw, Dw, wout, altho

__device__ __forceinline__ int zig_of(int h, int w, int H, int W) {
    const int He = H & ~1;
    return h < He ? (h >> 1) * (2 * W) + 2 * w + (h & 1) : He * W + w;
}
__device__ __forceinline__ void unzig(int l, int H, int W, int &h, int &w) {
    const int He = H & ~1;
    if (l < He * W) {
        const int p = l / (2 * W), r = l - p * 2 * W;
        h = 2 * p + (r & 1);
        w = r >> 1;
    } else {
        h = He;
        w = l - He * W;
    }
}

__device__ __forceinline__ float coord_weight(float altho, float &dwgt_daltho) {
    const float sp = altho <= 20.f ? log1pf(expf(altho)) : altho;  // F.softplus (threshold 20)
    const float sg = 1.f / (1.f + expf(-altho));
    dwgt_daltho = sp >= 0.01f ? (altho <= 20.f ? sg : 1.f) : 0.f;  // d max(softplus, 0.01) / d altho
    return fmaxf(sp, 0.01f);
}

typedef float v2f_ __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float readlane63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// layout of the weight-gradient vector (one per batch item, then summed): offsets in floats
template <int K>
struct GradLayout {
    static constexpr int D = 2 * K;
    int win, cw, cb, wx, wdt, dtb, A, Dp, wout, altho, total;
    __host__ __device__ explicit GradLayout(int N) {
        int o = 0;
        win = o;   o += 4 * K * K;
        cw = o;    o += D * 4;
        cb = o;    o += D;
        wx = o;    o += (1 + 2 * N) * D;
        wdt = o;   o += D;
        dtb = o;   o += D;
        A = o;     o += D * N;
        Dp = o;    o += D;
        wout = o;  o += K * D;
        altho = o; o += 1;
        total = o;
    }
};

// ---- what both directions share: y_off in zig-zag order into LDS ------------------------------------------------
// K L = K T blockDim values: exactly K T per thread, all loads issued before the first LDS store (a loop with a
// runtime trip count waited for every load in turn: 2 us of the first version's 7).
template <int K, int T>
__device__ __forceinline__ void stage_yoff(const SmallArgs &p, int b, float *yoff) {
    const int L = p.L;
    const float *ob = p.off + (long)b * 2 * K * L;   // the first K channels of the batch item are contiguous
    float v[K * T];
#pragma unroll
    for (int it = 0; it < K * T; ++it) v[it] = ob[it * blockDim.x + threadIdx.x];
#pragma unroll
    for (int it = 0; it < K * T; ++it) {
        const int idx = it * blockDim.x + threadIdx.x;
        const int k = idx / L, r = idx - k * L;
        const int h = r / p.W, ww = r - h * p.W;
        yoff[k * L + zig_of(h, ww, p.H, p.W)] = v[it];
    }
}

// in_proj, conv1d + SiLU, dt row, dt_proj + softplus for the T tokens l0 .. l0 + T - 1 of this lane
template <int K, int T, bool KEEP_PRE>
__device__ __forceinline__ void pre_phase(const SmallArgs &p, W_PARAMS, const float *yoff, int l0, float (&xs)[2 * K][T + 3],
                                          float (&z)[2 * K][T], float (&pp)[2 * K][T], float (&u)[2 * K][T],
                                          float (&dl)[2 * K][T], float (&dt)[T]) {
    constexpr int D = 2 * K;
    const int L = p.L;
#pragma unroll
    for (int j = 0; j < T + 3; ++j) {
        const int l = l0 - 3 + j;
        float yo[K];
#pragma unroll
        for (int k = 0; k < K; ++k) yo[k] = l >= 0 ? yoff[k * L + l] : 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) a = fmaf(win[d * K + k], yo[k], a);
            xs[d][j] = a;
            if (j >= 3) {
                float c = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) c = fmaf(win[(D + d) * K + k], yo[k], c);
                z[d][j - 3] = c;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < T; ++i) dt[i] = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float bv = cb ? cb[d] : 0.f;
        const float w0 = wx[d];
#pragma unroll
        for (int i = 0; i < T; ++i) {
            float acc = bv;
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = fmaf(cw[d * 4 + m], xs[d][i + m], acc);
            if (KEEP_PRE) pp[d][i] = acc;
            u[d][i] = acc * sigmoidf_(acc);
            dt[i] = fmaf(w0, u[d][i], dt[i]);
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float wv = wdt[d], bv = dtb ? dtb[d] : 0.f;
#pragma unroll
        for (int i = 0; i < T; ++i) dl[d][i] = softplus_thr(fmaf(wv, dt[i], bv));
    }
}

// ---- the carry between waves ----------------------------------------------------------------------------------------
// Per state n every wave publishes ONE record: the scan state at its end for all D channels, then a progress word
// (the number of records it has published).  Wave w reads the progress word and the record of wave w - 1 in one LDS
// round trip (LDS operations of a wave execute in order: a record read behind a progress word that already says
// "published" is the published record), combines and publishes its own without waiting for the writes.
// Explicit ds_ instructions on LDS byte addresses: through `volatile` generic pointers the compiler emitted
// flat_load / flat_store ... sc0 sc1 with s_waitcnt vmcnt(0) each -- every step then waited for every global store
// issued before it (first version, a record per channel pair: 39 us forward, 86 us backward for a 32 x 32 map).
// The compiler's waitcnt pass does not see memory operations inside inline asm: each asm waits for its own results.
typedef float v4f_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr(const void *q) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)q;
}
template <int D>
struct CarryRec {
    static constexpr int CS = 2 * D;   // floats per record: P[D] | S[D]   (12 or 4: 16-byte multiples)
};
#define MMU_COMPILER_FENCE() asm volatile("" ::: "memory")

// in[d] = state entering wave w (0 for wave 0).  Every wave publishes its OWN aggregate (P, S) -- known right after
// its lane scan, independent of what enters it -- and composes the aggregates of the waves before it itself: the
// carry costs one LDS round trip, not a chain of nw - 1 (the chained form, where wave w published P in + S and wave
// w + 1 waited for it, measured 570 cycles per hop: with two states per workgroup the last wave idled 4,000 cycles).
template <int D>
__device__ __forceinline__ void carry_all(float *hcar, int *prog, int w, int nw, int npp, int nn, const float (&Pt)[D],
                                          const float (&St)[D], float (&in)[D]) {
    constexpr int CS = CarryRec<D>::CS;
    if (w + 1 < nw && (threadIdx.x & 63) == 0) {
        float *rec = hcar + (w * npp + nn) * CS;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            rec[d] = Pt[d];
            rec[D + d] = St[d];
        }
        MMU_COMPILER_FENCE();      // (the LDS executes one wave's operations in order: data, then the progress word)
        prog[w] = nn + 1;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) in[d] = 0.f;
    if (w > 0) {
        for (;;) {
            MMU_COMPILER_FENCE();
            int ok = 1;
#pragma unroll
            for (int v = 0; v < 7; ++v)
                if (v < w) ok &= __builtin_amdgcn_readfirstlane(prog[v]) > nn;
            if (ok) break;
            __builtin_amdgcn_s_sleep(1);
        }
        MMU_COMPILER_FENCE();
#pragma unroll
        for (int v = 0; v < 7; ++v) {
            if (v < w) {
                const float *rec = hcar + (v * npp + nn) * CS;
#pragma unroll
                for (int d = 0; d < D; ++d) in[d] = fmaf(rec[d], in[d], rec[D + d]);
            }
        }
    }
}

// Inclusive affine scans over the 64 lanes for all D channels at once: the D independent chains are interleaved, so
// every DPP read is >= D instructions behind the write of the register it reads (no s_nop, no dependent-issue stalls:
// two chains at a time left a single wave per SIMD waiting on every second instruction).
#define MMU_SCAN6_STEP(ctrl, mask)                                                              \
    "v_fmac_f32_dpp %1, %1, %0 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                    \
    "v_fmac_f32_dpp %3, %3, %2 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                    \
    "v_fmac_f32_dpp %5, %5, %4 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                    \
    "v_fmac_f32_dpp %7, %7, %6 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                    \
    "v_fmac_f32_dpp %9, %9, %8 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                    \
    "v_fmac_f32_dpp %11, %11, %10 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                 \
    "v_mul_f32_dpp %0, %0, %0 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                     \
    "v_mul_f32_dpp %2, %2, %2 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                     \
    "v_mul_f32_dpp %4, %4, %4 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                     \
    "v_mul_f32_dpp %6, %6, %6 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                     \
    "v_mul_f32_dpp %8, %8, %8 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                     \
    "v_mul_f32_dpp %10, %10, %10 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"

template <int D>
__device__ __forceinline__ void wave_scan_affine_all(float (&P)[D], float (&S)[D]) {
    if constexpr (D == 6) {
        asm volatile("s_nop 1\n\t"
                     MMU_SCAN6_STEP("row_shr:1", "0xf")
                     MMU_SCAN6_STEP("row_shr:2", "0xf")
                     MMU_SCAN6_STEP("row_shr:4", "0xf")
                     MMU_SCAN6_STEP("row_shr:8", "0xf")
                     MMU_SCAN6_STEP("row_bcast:15", "0xa")
                     MMU_SCAN6_STEP("row_bcast:31", "0xc")
                     "s_nop 1"
                     : "+v"(P[0]), "+v"(S[0]), "+v"(P[1]), "+v"(S[1]), "+v"(P[2]), "+v"(S[2]), "+v"(P[3]), "+v"(S[3]),
                       "+v"(P[4]), "+v"(S[4]), "+v"(P[5]), "+v"(S[5]));
    } else {
        wave_scan_affine_x2(P[0], S[0], P[1], S[1]);
    }
}

// per-state scalars in LDS: record n = wb[D] (x_proj row 1 + n) | wc[D] (row 1 + N + n) | A[.][n] | padding
template <int D>
struct WRec {
    static constexpr int RS = (3 * D + 3) & ~3;
};
template <int D>
__device__ __forceinline__ void stage_wl(float *wl, const float *__restrict__ wx, const float *__restrict__ Aw, int N) {
    constexpr int RS = WRec<D>::RS;
    for (int idx = threadIdx.x; idx < N * 3 * D; idx += blockDim.x) {
        const int n = idx / (3 * D), r = idx - n * 3 * D;
        wl[n * RS + r] = r < D ? wx[(1 + n) * D + r] : r < 2 * D ? wx[(1 + N + n) * D + r - D] : Aw[(r - 2 * D) * N + n];
    }
}

// ================================================================================================================
// forward
// ================================================================================================================
template <int K, int T>
__global__ __launch_bounds__(512) void mamba_small_fwd_kernel(SmallArgs p, W_PARAMS) {
    constexpr int D = 2 * K;
    extern __shared__ float smem[];
    const int L = p.L, N = p.N, DN = D * N, G = L / T;
    constexpr int RS = WRec<D>::RS, CS = CarryRec<D>::CS;
    float *yoff = smem;                                         // [K][L]
    float *wl = yoff + K * L;                                   // [N][RS]   per-state scalars
    float *hcar = wl + N * RS;                                  // [nw][npp][CS]
    int *prog = (int *)(hcar + p.nw * p.npp * CS);              // [nw]
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, part = blockIdx.y;
    const int n0 = part * p.npp;
    const bool first = part == 0;     // the terms that are not sums over the states belong to part 0
    SMALL_STAMP(0, 0);
    stage_yoff<K, T>(p, b, yoff);
    stage_wl<D>(wl, wx, Aw, N);
    if (tid < p.nw) prog[tid] = 0;
    __syncthreads();
    SMALL_STAMP(0, 1);

    const int l0 = tid * T;
    float xs[D][T + 3], z[D][T], pp[D][T], u[D][T], dl[D][T], dt[T];
    pre_phase<K, T, false>(p, W_ARGS, yoff, l0, xs, z, pp, u, dl, dt);
    SMALL_STAMP(0, 2);
    float dlu[D][T], yacc[D][T];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float Dv = (Dw && first) ? Dw[d] : 0.f;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            dlu[d][i] = dl[d][i] * u[d][i];
            yacc[d][i] = Dv * u[d][i];
        }
    }
    float *hs = p.hstate ? p.hstate + (long)b * DN * G + tid : nullptr;

    for (int nn = 0; nn < p.npp; ++nn) {
        const int n = n0 + nn;
        const float *wr = wl + n * RS;
        float Bn[T], Cn[T];
#pragma unroll
        for (int i = 0; i < T; ++i) Bn[i] = Cn[i] = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float wb = wr[d], wc = wr[D + d];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                Bn[i] = fmaf(wb, u[d][i], Bn[i]);
                Cn[i] = fmaf(wc, u[d][i], Cn[i]);
            }
        }
        float P[D], S[D], pl[D][T], hl[D][T];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float A2 = wr[2 * D + d] * MMU_LOG2E;
            P[d] = 1.f;
            S[d] = 0.f;
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const float a = fast_exp2(dl[d][i] * A2);
                S[d] = fmaf(a, S[d], dlu[d][i] * Bn[i]);
                P[d] *= a;
                pl[d][i] = P[d];
                hl[d][i] = S[d];
            }
        }
        wave_scan_affine_all<D>(P, S);
        float Pt[D], St[D], in[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            Pt[d] = readlane63(P[d]);
            St[d] = readlane63(S[d]);
        }
        carry_all<D>(hcar, prog, w, p.nw, p.npp, nn, Pt, St, in);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            // state entering this lane's tokens
            const float h0 = fmaf(wave_shift_up1(P[d], 1.f), in[d], wave_shift_up1(S[d], 0.f));
            if (hs) hs[(long)(n * D + d) * G] = h0;
#pragma unroll
            for (int i = 0; i < T; ++i) yacc[d][i] = fmaf(Cn[i], fmaf(pl[d][i], h0, hl[d][i]), yacc[d][i]);
        }
    }

    SMALL_STAMP(0, 3);
    // gate, out_proj, inverse zig-zag, coordinates
    float dummy;
    const float wgt = coord_weight(altho[0], dummy);
    constexpr int c = K / 2;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int l = l0 + i;
        int h, ww;
        unzig(l, p.H, p.W, h, ww);
        float oz[D];
#pragma unroll
        for (int d = 0; d < D; ++d) oz[d] = yacc[d][i] * z[d][i] * sigmoidf_(z[d][i]);
        float off[K], cum[K];
#pragma unroll
        for (int k = 0; k < K; ++k) off[k] = yoff[k * L + l];
        cum[c] = 0.f;
#pragma unroll
        for (int j = 1; j <= c; ++j) {
            cum[c + j] = cum[c + j - 1] + off[c + j];
            cum[c - j] = cum[c - j + 1] + off[c - j];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float sq = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) sq = fmaf(wout[k * D + d], oz[d], sq);
            p.y[((((long)part * p.B + b) * K + k) * p.H + h) * p.W + ww] =
                fmaf(wgt, sq, first ? (float)h + p.scope * cum[k] : 0.f);
        }
    }
    SMALL_STAMP(0, 4);
}

// ================================================================================================================
// backward
// ================================================================================================================
// Sums NV per-lane values over the wave, four at a time, into slot[0 .. NV) (one per value; the lanes 12..15 that
// hold a batch's results write them).
template <int NV>
__device__ __forceinline__ void wave_sums_to(const float (&v)[NV], float *slot) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; i += 4) {
        const float r = wave_sum4_swap(v[i], i + 1 < NV ? v[i + 1] : 0.f, i + 2 < NV ? v[i + 2] : 0.f,
                                       i + 3 < NV ? v[i + 3] : 0.f);
        const int k = i + lane - 12;
        if (lane >= 12 && lane < 16 && k < NV) slot[k] = r;
    }
}

template <int K, int T>
__global__ __launch_bounds__(512) void mamba_small_bwd_kernel(SmallArgs p, W_PARAMS) {
    constexpr int D = 2 * K;
    extern __shared__ float smem[];
    const int L = p.L, N = p.N, DN = D * N, G = L / T;
    const GradLayout<K> lay(N);
    const int NV = lay.total;
    constexpr int RS = WRec<D>::RS, CS = CarryRec<D>::CS;
    float *yoff = smem;                                          // [K][L]
    float *dpl = smem + K * L;                                   // [D][L + 4]   conv1d backward exchange
    float *wl = dpl + D * (L + 4);                               // [N][RS]      per-state scalars (16-byte aligned)
    float *hcar = wl + N * RS;                                   // [nw][npp][CS]
    float *wpart = hcar + p.nw * p.npp * CS;                     // [nw][NV]
    int *prog = (int *)(wpart + p.nw * NV);                      // [nw]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, part = blockIdx.y;
    const int n0 = part * p.npp;
    const bool first = part == 0;     // the terms that are not sums over the states belong to part 0
    SMALL_STAMP(1, 0);
    // lanes take the token groups in REVERSE order: the adjoint scan runs forward over lanes and waves
    const int gr = G - 1 - tid;
    const int l0 = gr * T;
    float dyr[K][T];     // incoming gradient of the row coordinates (issued first: in flight during the staging)
#pragma unroll
    for (int i = 0; i < T; ++i) {
        int h, ww;
        unzig(l0 + i, p.H, p.W, h, ww);
#pragma unroll
        for (int k = 0; k < K; ++k) dyr[k][i] = p.dy[(((long)b * K + k) * p.H + h) * p.W + ww];
    }
    stage_yoff<K, T>(p, b, yoff);
    stage_wl<D>(wl, wx, Aw, N);
    if (tid < p.nw) prog[tid] = 0;
    for (int i = tid; i < p.nw * NV; i += blockDim.x) wpart[i] = 0.f;   // (the other parts' rows of dW_x / dA stay 0)
    for (int i = tid; i < D * 4; i += blockDim.x) dpl[(i >> 2) * (L + 4) + L + (i & 3)] = 0.f;   // tokens past the end
    __syncthreads();

    float xs[D][T + 3], z[D][T], pp[D][T], u[D][T], dl[D][T], dt[T];
    SMALL_STAMP(1, 1);
    pre_phase<K, T, true>(p, W_ARGS, yoff, l0, xs, z, pp, u, dl, dt);
    SMALL_STAMP(1, 2);

    float dwgt_da;
    const float wgt = coord_weight(altho[0], dwgt_da);
    // the incoming gradient through out_proj and the gate
    float doz[D][T], dyv[D][T], sz[D][T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) s = fmaf(wout[k * D + d], dyr[k][i], s);
            doz[d][i] = wgt * s;
            sz[d][i] = sigmoidf_(z[d][i]);
            dyv[d][i] = doz[d][i] * z[d][i] * sz[d][i];     // d out_z * silu(z) = gradient of y
        }
    }
    float dlu[D][T], du[D][T], ddl[D][T], yacc[D][T];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float Dv = (Dw && first) ? Dw[d] : 0.f;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            dlu[d][i] = dl[d][i] * u[d][i];
            du[d][i] = Dv * dyv[d][i];
            ddl[d][i] = 0.f;
            yacc[d][i] = Dv * u[d][i];
        }
    }
    const float *hs = p.hstate + (long)b * DN * G + gr;
    float *myw = wpart + w * NV;
    SMALL_STAMP(1, 3);

    for (int nn = 0; nn < p.npp; ++nn) {
        const int n = n0 + nn;
        const float *wr = wl + n * RS;
        float hin[D];     // saved states entering this lane's tokens: in flight during (1) and the carry
#pragma unroll
        for (int d = 0; d < D; ++d) hin[d] = hs[(long)(n * D + d) * G];
        float Bn[T], Cn[T], dBn[T], dCn[T];
#pragma unroll
        for (int i = 0; i < T; ++i) Bn[i] = Cn[i] = dBn[i] = dCn[i] = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float wb = wr[d], wc = wr[D + d];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                Bn[i] = fmaf(wb, u[d][i], Bn[i]);
                Cn[i] = fmaf(wc, u[d][i], Cn[i]);
            }
        }
        // (1) the adjoint recurrence gh_t = a_t g_t, g_t = c_t + gh_{t+1}, c_t = C_t dy_t: every channel's T tokens, last
        //     first, as one affine map (Q, R); only (Q, R) is kept -- (2) rebuilds a and c per channel (one exp and one
        //     multiply per token: cheaper than 4 T registers per channel across the scan)
        float Q[D], R[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float A2 = wr[2 * D + d] * MMU_LOG2E;
            Q[d] = 1.f;
            R[d] = 0.f;
#pragma unroll
            for (int i = T - 1; i >= 0; --i) {
                const float a = fast_exp2(dl[d][i] * A2);
                R[d] = a * fmaf(Cn[i], dyv[d][i], R[d]);
                Q[d] *= a;
            }
        }
        wave_scan_affine_all<D>(Q, R);
        float Qt[D], Rt[D], in[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            Qt[d] = readlane63(Q[d]);
            Rt[d] = readlane63(R[d]);
        }
        carry_all<D>(hcar, prog, w, p.nw, p.npp, nn, Qt, Rt, in);
        // (2) per channel: h forward from the saved state, then the adjoint backward with every gradient that needs it
        float dAv[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float Ar = wr[2 * D + d], A2 = Ar * MMU_LOG2E;
            float gh = fmaf(wave_shift_up1(Q[d], 1.f), in[d], wave_shift_up1(R[d], 0.f));   // gh of the token after this lane's last
            float hp = hin[d];                                                               // state entering its first token
            float a[T], hm[T];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                a[i] = fast_exp2(dl[d][i] * A2);
                hm[i] = a[i] * hp;
                hp = fmaf(dlu[d][i], Bn[i], hm[i]);
                yacc[d][i] = fmaf(Cn[i], hp, yacc[d][i]);
                dCn[i] = fmaf(dyv[d][i], hp, dCn[i]);
            }
            float da = 0.f;
#pragma unroll
            for (int i = T - 1; i >= 0; --i) {
                const float g = fmaf(Cn[i], dyv[d][i], gh);
                gh = a[i] * g;
                const float t0 = g * dl[d][i];
                dBn[i] = fmaf(t0, u[d][i], dBn[i]);
                du[d][i] = fmaf(t0, Bn[i], du[d][i]);
                da = fmaf(t0, hm[i], da);
                ddl[d][i] = fmaf(g, fmaf(u[d][i], Bn[i], Ar * hm[i]), ddl[d][i]);
            }
            dAv[d] = da;
        }
        // d x_dbl rows 1 + n (B_n) and 1 + N + n (C_n): back into u, and their x_proj weight gradients
        float wv[3 * D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float wb = wr[d], wc = wr[D + d];
            float sb = 0.f, sc = 0.f;
#pragma unroll
            for (int i = 0; i < T; ++i) {
                du[d][i] = fmaf(wb, dBn[i], fmaf(wc, dCn[i], du[d][i]));
                sb = fmaf(dBn[i], u[d][i], sb);
                sc = fmaf(dCn[i], u[d][i], sc);
            }
            wv[d] = sb;
            wv[D + d] = sc;
            wv[2 * D + d] = dAv[d];
        }
        // (three destinations: two rows of dW_x and one column of dA)
        {
            const int ln = lane;
#pragma unroll
            for (int i = 0; i < 3 * D; i += 4) {
                const float r = wave_sum4_swap(wv[i], i + 1 < 3 * D ? wv[i + 1] : 0.f, i + 2 < 3 * D ? wv[i + 2] : 0.f,
                                               i + 3 < 3 * D ? wv[i + 3] : 0.f);
                const int k = i + ln - 12;
                if (ln >= 12 && ln < 16 && k < 3 * D) {
                    const int which = k / D, d = k - which * D;
                    const int dst = which == 0 ? lay.wx + (1 + n) * D + d
                                  : which == 1 ? lay.wx + (1 + N + n) * D + d
                                               : lay.A + d * N + n;
                    myw[dst] = r;
                }
            }
        }
    }

    SMALL_STAMP(1, 4);
    // ---- behind the scan: gate, dt row, softplus, conv1d, in_proj, coordinates ---------------------------------
    float wsm[K * D + 1 + 4 * D];   // dWout [K*D], dwgt, dD [D], dWdt [D], dbias [D], dWx row 0 [D]
#pragma unroll
    for (int i = 0; i < K * D + 1 + 4 * D; ++i) wsm[i] = 0.f;
    float dzv[D][T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        float oz[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float zs = z[d][i] * sz[d][i];                        // silu(z)
            oz[d] = yacc[d][i] * zs;
            dzv[d][i] = doz[d][i] * yacc[d][i] * sz[d][i] * (1.f + z[d][i] * (1.f - sz[d][i]));
            if (first) wsm[K * D + 1 + d] = fmaf(dyv[d][i], u[d][i], wsm[K * D + 1 + d]);   // dD
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float sq = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                sq = fmaf(wout[k * D + d], oz[d], sq);
                wsm[k * D + d] = fmaf(wgt * dyr[k][i], oz[d], wsm[k * D + d]);        // dWout
            }
            wsm[K * D] = fmaf(dyr[k][i], sq, wsm[K * D]);                              // d wgt
        }
        // softplus, dt_proj, the dt row of x_proj
        float ddt = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float raw = fmaf(wdt[d], dt[i], dtb ? dtb[d] : 0.f);
            const float draw = ddl[d][i] * (raw <= 20.f ? sigmoidf_(raw) : 1.f);   // softplus' (threshold 20)
            ddt = fmaf(wdt[d], draw, ddt);
            wsm[K * D + 1 + D + d] = fmaf(draw, dt[i], wsm[K * D + 1 + D + d]);        // dWdt
            wsm[K * D + 1 + 2 * D + d] += draw;                                        // d dt_bias
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            du[d][i] = fmaf(wx[d], ddt, du[d][i]);
            wsm[K * D + 1 + 3 * D + d] = fmaf(ddt, u[d][i], wsm[K * D + 1 + 3 * D + d]);   // dWx row 0
        }
    }
    // conv1d backward: dp = du * silu'(pre); neighbours' dp through LDS
    float wcv[5 * D];   // dcw [D][4], dcb [D]
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f}, sb = 0.f;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const float sg = sigmoidf_(pp[d][i]);
            const float dp = du[d][i] * sg * (1.f + pp[d][i] * (1.f - sg));
            dpl[d * (L + 4) + l0 + i] = dp;
            sb += dp;
#pragma unroll
            for (int m = 0; m < 4; ++m) s4[m] = fmaf(xs[d][i + m], dp, s4[m]);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) wcv[d * 4 + m] = s4[m];
        wcv[4 * D + d] = sb;
    }
    SMALL_STAMP(1, 5);
    __syncthreads();
    SMALL_STAMP(1, 6);
    float wiv[4 * K * K];
#pragma unroll
    for (int i = 0; i < 4 * K * K; ++i) wiv[i] = 0.f;
    constexpr int c = K / 2;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int l = l0 + i;
        float dxz[2 * D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) s = fmaf(cw[d * 4 + m], dpl[d * (L + 4) + l + 3 - m], s);
            dxz[d] = s;
            dxz[D + d] = dzv[d][i];
        }
        float yo[K], g[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            yo[k] = yoff[k * L + l];
            g[k] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 2 * D; ++j) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                g[k] = fmaf(win[j * K + k], dxz[j], g[k]);
                wiv[j * K + k] = fmaf(dxz[j], yo[k], wiv[j * K + k]);
            }
        }
        // the coordinate terms: tap j > c feeds cum[k] for k >= j, tap j < c for k <= j
        if (first) {
            float run = 0.f;
#pragma unroll
            for (int j = K - 1; j > c; --j) {
                run += dyr[j][i];
                g[j] = fmaf(p.scope, run, g[j]);
            }
            run = 0.f;
#pragma unroll
            for (int j = 0; j < c; ++j) {
                run += dyr[j][i];
                g[j] = fmaf(p.scope, run, g[j]);
            }
        }
        int h, ww;
        unzig(l, p.H, p.W, h, ww);
#pragma unroll
        for (int k = 0; k < K; ++k) p.doff[((((long)part * p.B + b) * K + k) * p.H + h) * p.W + ww] = g[k];
    }
    SMALL_STAMP(1, 7);
    // weight gradients: wave sums into this wave's LDS slots, then the waves are added and the batch item's partial
    // vector goes out
    wave_sums_to<4 * K * K>(wiv, myw + lay.win);
    wave_sums_to<5 * D>(wcv, myw + lay.cw);                  // cw and cb are adjacent in the layout
    {
        float tmp[K * D];
#pragma unroll
        for (int i = 0; i < K * D; ++i) tmp[i] = wsm[i];
        wave_sums_to<K * D>(tmp, myw + lay.wout);
        float one[1] = {wsm[K * D]};
        wave_sums_to<1>(one, myw + lay.altho);
        float t4[D];
#pragma unroll
        for (int i = 0; i < D; ++i) t4[i] = wsm[K * D + 1 + i];
        wave_sums_to<D>(t4, myw + lay.Dp);
#pragma unroll
        for (int i = 0; i < D; ++i) t4[i] = wsm[K * D + 1 + D + i];
        wave_sums_to<D>(t4, myw + lay.wdt);
#pragma unroll
        for (int i = 0; i < D; ++i) t4[i] = wsm[K * D + 1 + 2 * D + i];
        wave_sums_to<D>(t4, myw + lay.dtb);
#pragma unroll
        for (int i = 0; i < D; ++i) t4[i] = wsm[K * D + 1 + 3 * D + i];
        wave_sums_to<D>(t4, myw + lay.wx);
    }
    SMALL_STAMP(1, 8);
    __syncthreads();
    SMALL_STAMP(1, 9);
    for (int i = tid; i < NV; i += blockDim.x) {
        float s = 0.f;
        for (int ww = 0; ww < p.nw; ++ww) s += wpart[ww * NV + i];
        if (i == lay.altho) s *= dwgt_da;
        p.part[((long)b * p.ns + part) * NV + i] = s;
    }
    SMALL_STAMP(1, 10);
}

// Workgroups [0, nbw): dweights[i] = sum over the B * ns slots of part[slot][i].  The rest: d offset [B, 2K, H, W] =
// sum over the ns parts of dparts[part][b][k][pixel] for the first K channels, 0 for the others.  Fixed order.
__global__ __launch_bounds__(256) void mamba_small_reduce_kernel(const float *__restrict__ part, float *__restrict__ out,
                                                                 int slots, int NV, int nbw,
                                                                 const float *__restrict__ dparts,
                                                                 float *__restrict__ doff, int B, int K, int L, int ns) {
    if ((int)blockIdx.x < nbw) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i >= NV) return;
        float s = 0.f;
        for (int b = 0; b < slots; ++b) s += part[(long)b * NV + i];
        out[i] = s;
        return;
    }
    const long idx = (long)(blockIdx.x - nbw) * 256 + threadIdx.x;
    if (idx >= (long)B * 2 * K * L) return;
    const int b = (int)(idx / (2L * K * L));
    const long r = idx - (long)b * 2 * K * L;
    float s = 0.f;
    if (r < (long)K * L) {
        const long ps = (long)B * K * L;
        const float *src = dparts + (long)b * K * L + r;
        for (int j = 0; j < ns; ++j) s += src[j * ps];
    }
    doff[idx] = s;
}

// tokens per lane / waves for a sequence length: the smallest T in {1, 2, 4} with L = 64 T nw, nw <= 8
bool plan(int L, int &T, int &nw) {
    for (int t = 1; t <= 4; t *= 2) {
        if (L % (64 * t) == 0 && L / (64 * t) <= 8) {
            T = t;
            nw = L / (64 * t);
            return true;
        }
    }
    return false;
}

int grad_total(int K, int N) { return K == 3 ? GradLayout<3>(N).total : GradLayout<1>(N).total; }

// State-range parts per batch item: a scan kernel with one workgroup per batch item keeps 8 of 256 CUs busy and is
// bound by the instruction issue of that one CU; the states are independent up to the final sums over n, so part j takes
// states [j N / ns, (j + 1) N / ns) and the partial results are added downstream (the sampler adds the row maps while it
// reads them, mamba_small_reduce_kernel adds the gradients).  MMU_SMALL_PARTS overrides (tuning / tests).
int default_parts(int batch, int N) {
    static const int forced = []() { const char *e = getenv("MMU_SMALL_PARTS"); return e ? atoi(e) : 0; }();
    int ns = forced > 0 ? forced : 8;
    while (ns > 1 && (N % ns != 0 || (long)batch * ns > 512)) ns >>= 1;
    if (N % ns != 0) ns = 1;
    return ns;
}

int check(const mmu_mamba_small_params *p, const char *name, int &T, int &nw) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->taps == 1 || p->taps == 3, "%s: 1 or 3 taps supported (got %d)", name, p->taps);
    MMU_CHECK(p->batch > 0 && p->height > 0 && p->width > 0, "%s: empty tensor", name);
    MMU_CHECK(p->dstate >= 1 && p->dstate <= 64, "%s: d_state must be in 1..64 (got %d)", name, p->dstate);
    MMU_CHECK(plan(p->height * p->width, T, nw),
              "%s: height * width must be a multiple of 64, at most 2048, and 64 * {1,2,4} * (<= 8 waves) (got %d)", name,
              p->height * p->width);
    MMU_CHECK(p->parts >= 1 && p->dstate % p->parts == 0, "%s: parts (%d) must divide d_state (%d)", name, p->parts,
              p->dstate);
    MMU_CHECK(p->offset && p->in_proj_weight && p->conv_weight && p->x_proj_weight && p->dt_proj_weight && p->A &&
                  p->out_proj_weight && p->altho,
              "%s: offset, in_proj / conv / x_proj / dt_proj / out_proj weights, A and altho are required", name);
    return 0;
}

SmallArgs to_args(const mmu_mamba_small_params *p, int nw) {
    SmallArgs a = {};
    a.B = p->batch; a.H = p->height; a.W = p->width; a.L = p->height * p->width; a.N = p->dstate; a.nw = nw;
    a.ns = p->parts; a.npp = p->dstate / p->parts;
    a.scope = p->extend_scope;
    a.off = p->offset; a.win = p->in_proj_weight; a.cw = p->conv_weight; a.cb = p->conv_bias;
    a.wx = p->x_proj_weight; a.wdt = p->dt_proj_weight; a.dtb = p->dt_bias; a.A = p->A; a.Dp = p->D;
    a.wout = p->out_proj_weight; a.altho = p->altho; a.y = p->y; a.hstate = p->hstate; a.dy = p->dy;
    a.doff = p->doffset; a.part = p->workspace;
    return a;
}

template <typename F>
int set_lds_attr(F kernel, size_t bytes, const char *name) {
    if (bytes > 160 * 1024) return mmu_fail("%s: needs %zu B of LDS (> 160 KiB)", name, bytes);
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return mmu_fail("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
    }
    return 0;
}

#define SMALL_DISPATCH(KERNEL, K_, T_, ...)                         \
    do {                                                            \
        if (K_ == 3) {                                              \
            if (T_ == 1) { KERNEL(3, 1, __VA_ARGS__); }             \
            else if (T_ == 2) { KERNEL(3, 2, __VA_ARGS__); }        \
            else { KERNEL(3, 4, __VA_ARGS__); }                     \
        } else {                                                    \
            if (T_ == 1) { KERNEL(1, 1, __VA_ARGS__); }             \
            else if (T_ == 2) { KERNEL(1, 2, __VA_ARGS__); }        \
            else { KERNEL(1, 4, __VA_ARGS__); }                     \
        }                                                           \
    } while (0)

}  // namespace

#ifdef MMU_SMALL_STAMPS
extern "C" int mmu_debug_small_stamps(unsigned long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_small_stamps), sizeof(g_small_stamps)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int mmu_mamba_small_supported(int taps, int height, int width, int dstate) {
    int T, nw;
    return (taps == 1 || taps == 3) && dstate >= 1 && dstate <= 64 && height > 0 && width > 0 &&
           plan(height * width, T, nw);
}

extern "C" int mmu_mamba_small_tokens_per_lane(int height, int width) {
    int T, nw;
    return plan(height * width, T, nw) ? T : 0;
}

// floats of hstate: batch * 2 * taps * dstate * (L / T)
extern "C" size_t mmu_mamba_small_state_floats(int batch, int taps, int height, int width, int dstate) {
    int T, nw;
    if (!plan(height * width, T, nw)) return 0;
    return (size_t)batch * 2 * taps * dstate * (size_t)(height * width / T);
}

extern "C" int mmu_mamba_small_parts(int batch, int dstate) { return default_parts(batch, dstate); }

// floats of the backward workspace: batch * parts weight-gradient partial vectors + parts partial d offset maps
extern "C" size_t mmu_mamba_small_bwd_workspace_floats(int batch, int taps, int height, int width, int dstate, int parts) {
    if ((taps != 1 && taps != 3) || parts < 1) return 0;
    return (size_t)batch * parts * grad_total(taps, dstate) + (size_t)parts * batch * taps * height * width;
}

// floats of the weight-gradient vector (its layout: in_proj [4K][K] | conv weight [2K][4] | conv bias [2K] |
// x_proj [1+2N][2K] | dt_proj [2K] | dt bias [2K] | A [2K][N] | D [2K] | out_proj [K][2K] | altho)
extern "C" size_t mmu_mamba_small_grad_floats(int taps, int dstate) {
    if (taps != 1 && taps != 3) return 0;
    return (size_t)grad_total(taps, dstate);
}

extern "C" int mmu_mamba_small_fwd(const mmu_mamba_small_params *p, void *stream) {
    int T, nw;
    if (int r = check(p, "mamba_small_fwd", T, nw)) return r;
    MMU_CHECK(p->y != nullptr, "mamba_small_fwd: y is required");
    const SmallArgs a = to_args(p, nw);
    const int K = p->taps, D = 2 * K;
    const int RS = (3 * D + 3) & ~3, CS = 2 * D;
    const size_t lds = sizeof(float) * ((size_t)K * a.L + (size_t)a.N * RS + (size_t)nw * a.npp * CS + nw);
    hipStream_t st = (hipStream_t)stream;
#define SMALL_FWD(K_, T_, a_)                                                                         \
    if (int r = set_lds_attr(mamba_small_fwd_kernel<K_, T_>, lds, "mamba_small_fwd")) return r;       \
    mamba_small_fwd_kernel<K_, T_><<<dim3(a_.B, a_.ns), 64 * nw, lds, st>>>(a_, a_.win, a_.cw, a_.cb, a_.wx, a_.wdt, a_.dtb, a_.A, a_.Dp, a_.wout, a_.altho)
    SMALL_DISPATCH(SMALL_FWD, K, T, a);
#undef SMALL_FWD
    MMU_HIP_LAUNCH_CHECK("mamba_small_fwd");
    return 0;
}

extern "C" int mmu_mamba_small_bwd(const mmu_mamba_small_params *p, void *stream) {
    int T, nw;
    if (int r = check(p, "mamba_small_bwd", T, nw)) return r;
    MMU_CHECK(p->hstate && p->dy && p->doffset && p->workspace && p->dweights,
              "mamba_small_bwd: hstate, dy, doffset, workspace and dweights are required");
    SmallArgs a = to_args(p, nw);
    // workspace: [B * parts][NV] weight-gradient partials, then [parts][B][K][L] partial d offset
    a.part = p->workspace;
    a.doff = p->workspace + (size_t)a.B * a.ns * grad_total(p->taps, a.N);
    const int K = p->taps, D = 2 * K;
    const int NV = grad_total(K, a.N);
    const int RS = (3 * D + 3) & ~3, CS = 2 * D;
    const size_t lds = sizeof(float) * ((size_t)K * a.L + (size_t)D * (a.L + 4) + (size_t)a.N * RS +
                                        (size_t)nw * a.npp * CS + (size_t)nw * NV + nw);
    hipStream_t st = (hipStream_t)stream;
#define SMALL_BWD(K_, T_, a_)                                                                         \
    if (int r = set_lds_attr(mamba_small_bwd_kernel<K_, T_>, lds, "mamba_small_bwd")) return r;       \
    mamba_small_bwd_kernel<K_, T_><<<dim3(a_.B, a_.ns), 64 * nw, lds, st>>>(a_, a_.win, a_.cw, a_.cb, a_.wx, a_.wdt, a_.dtb, a_.A, a_.Dp, a_.wout, a_.altho)
    SMALL_DISPATCH(SMALL_BWD, K, T, a);
#undef SMALL_BWD
    MMU_HIP_LAUNCH_CHECK("mamba_small_bwd");
    const int nbw = (NV + 255) / 256;
    const long nd = (long)a.B * 2 * K * a.L;
    mamba_small_reduce_kernel<<<nbw + (unsigned)((nd + 255) / 256), 256, 0, st>>>(a.part, p->dweights, a.B * a.ns, NV, nbw,
                                                                                 a.doff, p->doffset, a.B, K, a.L, a.ns);
    MMU_HIP_LAUNCH_CHECK("mamba_small_reduce");
    return 0;
}
