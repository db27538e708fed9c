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
// dependent-node floor of a replayed graph ("What's weak" 7).
//
// Decomposition (the third one built; the first two are in HISTORY.md):
//   * grid (batch, parts): a workgroup owns one batch item and a RANGE OF STATES n.  Everything behind the scan is a
//     sum over n (y, and in the backward du / d delta / dz / every weight gradient), so the parts' results are partial
//     sums that are added downstream: the sampler adds the partial row maps while it reads them
//     (mmu_morph_params.y_parts), mamba_small_reduce_kernel adds the partial gradients in fixed order.  One workgroup per
//     batch item kept 8 of 256 CUs busy and was bound by the instruction issue of that one CU.
//   * inside a workgroup a WAVE owns a CHANNEL and a lane a contiguous run of TL = L / 64 tokens (<= 16).  A channel's
//     scan then never leaves its wave: the lane composes its TL tokens, one DPP scan over the 64 lanes (two states
//     interleaved), done -- no carry between waves, no barrier in the forward state loop, and the scan's fixed cost is
//     spread over TL tokens.  (Lanes = token groups of ALL channels with a chain of carries between the waves: 33
//     lane-instructions per (token, channel, state), 570 cycles per hop; this form: ~7.)
//   * what crosses channels goes through LDS tables between workgroup barriers: u -> (dt, B_n, C_n) by all threads,
//     out_z -> out_proj + coordinates by all threads; backward: per-channel dB_n / dC_n -> their sums, d dt, d xz ->
//     d y_off.  A wave owns its channel's rows of every weight gradient: 18 + 3 per state wave sums instead of 110.
//   * the backward recomputes h by the same lane scan (nothing is saved by the forward) and runs the adjoint
//     recurrence g_t = C_t dy_t + a_{t+1} g_{t+1} as a scan over the lanes in reverse order (wave_reverse + the same DPP
//     scan, interleaved with the forward one).
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
    int B, H, W, lw, N;  // lw = log2(W) when W is a power of two (shifts instead of divisions), -1 otherwise
    int Lr;              // H * W: the real tokens; the kernels' tables hold L = 64 * TL >= Lr slots, the rest are padding
                         // (y_off = d rows = 0 there: behind the last real token in the forward scan, g = 0 in the adjoint)
    int ns, npp;         // state-range parts per batch item (grid.y) and states per part (N = ns * npp)
    float scope;
    const float *off;    // [B, 2K, H, W]
    float *y;            // [ns][B, K, H, W]  partial row maps (their sum is the row map)
    const float *dy;     // [B, K, H, W]
    float *doff;         // [ns][B][K][H*W]  partial d offset (first K channels), pixel order
    float *part;         // [B * ns][NV] weight-gradient partials
};

// The weights are separate __restrict__ kernel parameters, not members of SmallArgs: only then can the compiler prove
// that the kernel's own stores do not clobber them and read them with scalar loads into SGPRs (as struct members they
// became per-lane global_load_dword in the middle of the kernel, each behind the stores' vmcnt).
#define W_PARAMS                                                                                                      \
    const float *__restrict__ win, const float *__restrict__ cw, const float *__restrict__ cb,                       \
        const float *__restrict__ wx, const float *__restrict__ wdt, const float *__restrict__ dtb,                  \
        const float *__restrict__ Aw, const float *__restrict__ Dw, const float *__restrict__ wout,                  \
        const float *__restrict__ altho

// zig-zag token of pixel (h, w) and back (two-row zig-zag, MMUNet.py:196-242); lw >= 0: W = 1 << lw
__device__ __forceinline__ int zig_of(int h, int w, int H, int W, int lw) {
    const int He = H & ~1;
    if (lw >= 0) return h < He ? ((h >> 1) << (lw + 1)) + 2 * w + (h & 1) : (He << lw) + w;
    return h < He ? (h >> 1) * (2 * W) + 2 * w + (h & 1) : He * W + w;
}
__device__ __forceinline__ void unzig(int l, int H, int W, int lw, int &h, int &w) {
    const int He = H & ~1;
    if (l < He * W) {
        int p, r;
        if (lw >= 0) {
            p = l >> (lw + 1);
            r = l & ((2 << lw) - 1);
        } else {
            p = (int)((unsigned)l / (unsigned)(2 * W));
            r = l - p * 2 * W;
        }
        h = 2 * p + (r & 1);
        w = r >> 1;
    } else {
        h = He;
        w = l - He * W;
    }
}

// max(softplus(altho), .01) and its derivative (MMUNet.py:186); hardware exp2 / log2 (softplus_thr)
__device__ __forceinline__ float coord_weight(float altho, float &dwgt_daltho) {
    const float sp = softplus_thr(altho);
    dwgt_daltho = sp >= 0.01f ? (altho <= 20.f ? sigmoidf_(altho) : 1.f) : 0.f;
    return fmaxf(sp, 0.01f);
}

// layout of the weight-gradient vector (one per workgroup, then summed): offsets in floats
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

// ---- a lane's run of TL consecutive tokens of an LDS row, as 16-byte accesses where TL allows (the dynamic LDS base is
// declared 16-byte aligned and every table offset is a multiple of 4 floats).  Lane stride = TL floats: scalar accesses
// would hit 64 / TL distinct banks only (16-way conflicts at TL = 16); the 16-byte form is 4-way at worst.
template <int TL>
__device__ __forceinline__ void run_load(const float *q, float (&v)[TL]) {
    if constexpr (TL % 4 == 0) {
#pragma unroll
        for (int i = 0; i < TL; i += 4) {
            const float4 f = *reinterpret_cast<const float4 *>(q + i);
            v[i] = f.x; v[i + 1] = f.y; v[i + 2] = f.z; v[i + 3] = f.w;
        }
    } else if constexpr (TL % 2 == 0) {
#pragma unroll
        for (int i = 0; i < TL; i += 2) {
            const float2 f = *reinterpret_cast<const float2 *>(q + i);
            v[i] = f.x; v[i + 1] = f.y;
        }
    } else {
        static_assert(TL == 1, "odd runs other than 1 are not instantiated");
        v[0] = q[0];
    }
}
template <int TL>
__device__ __forceinline__ void run_store(float *q, const float (&v)[TL]) {
    if constexpr (TL % 4 == 0) {
#pragma unroll
        for (int i = 0; i < TL; i += 4) *reinterpret_cast<float4 *>(q + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
    } else if constexpr (TL % 2 == 0) {
#pragma unroll
        for (int i = 0; i < TL; i += 2) *reinterpret_cast<float2 *>(q + i) = make_float2(v[i], v[i + 1]);
    } else {
        q[0] = v[0];
    }
}

// ---- LDS staging: a [K][H][W] map of one batch item in zig-zag token order, all loads in flight before the first store
template <int K, int TL>
__device__ __forceinline__ void stage_zig(const SmallArgs &p, const float *__restrict__ src, float *dst) {
    constexpr int NT = 128 * K;                    // threads: 64 per channel, 2K channels
    constexpr int IT = (K * 64 * TL + NT - 1) / NT;  // values per thread
    constexpr int L = 64 * TL;
    const int Lr = p.Lr, n = K * Lr;               // src: [K][H][W] of one batch item, Lr = H * W <= L values per channel
    float v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = it * NT + threadIdx.x;
        v[it] = idx < n ? src[idx] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = it * NT + threadIdx.x;
        if (idx < n) {
            int k, r, h, ww;
            if (p.lw >= 0 && Lr == L) {
                k = idx / L; r = idx & (L - 1);
                h = r >> p.lw; ww = r & (p.W - 1);
            } else {
                k = (int)((unsigned)idx / (unsigned)Lr); r = idx - k * Lr;
                h = (int)((unsigned)r / (unsigned)p.W); ww = r - h * p.W;
            }
            dst[k * L + zig_of(h, ww, p.H, p.W, p.lw)] = v[it];
        }
    }
    // the padding slots [Lr, L) of every channel row (the zig-zag order maps the Lr pixels onto [0, Lr))
    const int pad = L - Lr;
    for (int i = threadIdx.x; i < K * pad; i += NT) {
        const int k = i / pad;
        dst[k * L + Lr + (i - k * pad)] = 0.f;
    }
}

#define MMU_COMPILER_FENCE() asm volatile("" ::: "memory")

// per-state scalars in LDS: record nn = wb[D] (x_proj row 1 + n) | wc[D] (row 1 + N + n) | A[.][n] | padding
template <int D>
struct WRec {
    static constexpr int RS = (3 * D + 3) & ~3;
};
template <int D>
__device__ __forceinline__ void stage_wl(float *wl, const float *__restrict__ wx, const float *__restrict__ Aw, int N,
                                         int n0, int npp) {
    constexpr int RS = WRec<D>::RS;
    for (int idx = threadIdx.x; idx < npp * 3 * D; idx += blockDim.x) {
        const int nn = idx / (3 * D), r = idx - nn * 3 * D, n = n0 + nn;
        wl[nn * RS + r] = r < D ? wx[(1 + n) * D + r] : r < 2 * D ? wx[(1 + N + n) * D + r - D] : Aw[(r - 2 * D) * N + n];
    }
}

// The wave's scalars (its channel's rows of the small weights), read at kernel entry: one batch of scalar loads whose
// latency (~1 us from a cold scalar cache) overlaps the staging loads instead of stalling each phase at first use.
template <int K>
struct ChanW {
    float wi[K], wz[K];        // in_proj rows d and 2K + d
    float c0, c1, c2, c3, cb;  // conv1d taps and bias
    float w0[2 * K];           // x_proj row 0 (the dt row), all channels
    float wdt, dtb, Dv;        // dt_proj, its bias, D
    float wo[K];               // out_proj column d
    float altho;
};
template <int K>
__device__ __forceinline__ ChanW<K> load_chan(int d, bool first, W_PARAMS) {
    constexpr int D = 2 * K;
    ChanW<K> c;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        c.wi[k] = win[d * K + k];
        c.wz[k] = win[(D + d) * K + k];
        c.wo[k] = wout[k * D + d];
    }
    c.c0 = cw[d * 4]; c.c1 = cw[d * 4 + 1]; c.c2 = cw[d * 4 + 2]; c.c3 = cw[d * 4 + 3];
    c.cb = cb ? cb[d] : 0.f;
#pragma unroll
    for (int dd = 0; dd < D; ++dd) c.w0[dd] = wx[dd];
    c.wdt = wdt[d];
    c.dtb = dtb ? dtb[d] : 0.f;
    c.Dv = (Dw && first) ? Dw[d] : 0.f;
    c.altho = altho[0];
    (void)Aw;
    return c;
}
#define W_ARGS win, cw, cb, wx, wdt, dtb, Aw, Dw, wout, altho

// own channel d of wave: in_proj (x and z rows), conv1d (width 4, halo from the tokens before the lane's run), SiLU
template <int K, int TL>
__device__ __forceinline__ void pre_channel(const ChanW<K> &cwv, const float *yoff, int L, int l0,
                                            float (&x)[TL + 3], float (&z)[TL], float (&pp)[TL], float (&u)[TL]) {
    const float (&wi)[K] = cwv.wi;
    const float (&wz)[K] = cwv.wz;
    // y_off of the tokens l0 - 3 .. l0 + TL - 1: the lane's own run plus the three before it (the previous lane's tail)
    float yo[K][TL + 3];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float own[TL];
        run_load<TL>(yoff + k * L + l0, own);
#pragma unroll
        for (int i = 0; i < TL; ++i) yo[k][3 + i] = own[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) yo[k][j] = l0 - 3 + j >= 0 ? yoff[k * L + l0 - 3 + j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < TL + 3; ++j) {
        float a = 0.f, c = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            a = fmaf(wi[k], yo[k][j], a);
            c = fmaf(wz[k], yo[k][j], c);
        }
        x[j] = a;
        if (j >= 3) z[j - 3] = c;
    }
    const float bv = cwv.cb, c0 = cwv.c0, c1 = cwv.c1, c2 = cwv.c2, c3 = cwv.c3;
#pragma unroll
    for (int i = 0; i < TL; ++i) {
        const float acc = fmaf(c3, x[i + 3], fmaf(c2, x[i + 2], fmaf(c1, x[i + 1], fmaf(c0, x[i], bv))));
        pp[i] = acc;
        u[i] = acc * sigmoidf_(acc);
    }
}

// dt row and the B_n / C_n rows of a state from the u table, by all threads (token l per thread and step)
template <int D>
__device__ __forceinline__ void coop_dt(const float *ut, float *dtt, const float (&w0)[D], int L) {
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) s = fmaf(w0[d], ut[d * L + l], s);
        dtt[l] = s;
    }
}
template <int D>
__device__ __forceinline__ void coop_bc(const float *ut, float *bc /* [2][L] */, const float *wr, int L) {
    float wb[D], wc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        wb[d] = wr[d];
        wc[d] = wr[D + d];
    }
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        float sb = 0.f, sc = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float uu = ut[d * L + l];
            sb = fmaf(wb[d], uu, sb);
            sc = fmaf(wc[d], uu, sc);
        }
        bc[l] = sb;
        bc[L + l] = sc;
    }
}

// ================================================================================================================
// forward
// ================================================================================================================
template <int K, int TL>
__global__ __launch_bounds__(128 * K) void mamba_small_fwd_kernel(SmallArgs p, W_PARAMS) {
    constexpr int D = 2 * K, RS = WRec<D>::RS;
    extern __shared__ __align__(16) float smem[];
    constexpr int L = 64 * TL;
    const int N = p.N, npp = p.npp;
    float *yoff = smem;              // [K][L]      y_off, zig-zag order
    float *ut = yoff + K * L;        // [D][L]      u; after the scan: out_z
    float *dtt = ut + D * L;         // [L]         dt row
    float *bct = dtt + L;            // [npp][2][L] B_n, C_n
    float *wl = bct + npp * 2 * L;   // [npp][RS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int d = __builtin_amdgcn_readfirstlane(tid >> 6);     // this wave's channel
    const int b = blockIdx.x, part = blockIdx.y;
    const int n0 = part * npp;
    const bool first = part == 0;     // the terms that are not sums over the states belong to part 0
    SMALL_STAMP(0, 0);
    const ChanW<K> cwv = load_chan<K>(d, first, W_ARGS);
    stage_zig<K, TL>(p, p.off + (long)b * 2 * K * p.Lr, yoff);
    stage_wl<D>(wl, wx, Aw, N, n0, npp);
    __syncthreads();
    SMALL_STAMP(0, 1);

    const int l0 = lane * TL;
    float x[TL + 3], z[TL], pp[TL], u[TL];
    pre_channel<K, TL>(cwv, yoff, L, l0, x, z, pp, u);
    run_store<TL>(ut + d * L + l0, u);
    __syncthreads();
    coop_dt<D>(ut, dtt, cwv.w0, L);
    for (int nn = 0; nn < npp; ++nn) coop_bc<D>(ut, bct + nn * 2 * L, wl + nn * RS, L);
    __syncthreads();
    SMALL_STAMP(0, 2);

    float dl[TL], dlu[TL], yacc[TL];
    {
        const float wv = cwv.wdt, bv = cwv.dtb, Dv = cwv.Dv;
        float dtv[TL];
        run_load<TL>(dtt + l0, dtv);
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            dl[i] = softplus_thr(fmaf(wv, dtv[i], bv));
            dlu[i] = dl[i] * u[i];
            yacc[i] = Dv * u[i];
        }
    }
    // two states per round: their two lane scans are interleaved (every DPP read two instructions behind its write)
    for (int nn = 0; nn < npp; nn += 2) {
        const bool two = nn + 1 < npp;
        const int n1 = two ? nn + 1 : nn;
        const float A0 = wl[nn * RS + 2 * D + d] * MMU_LOG2E, A1 = wl[n1 * RS + 2 * D + d] * MMU_LOG2E;
        float B0[TL], B1[TL], C0[TL], C1[TL];
        run_load<TL>(bct + nn * 2 * L + l0, B0);
        run_load<TL>(bct + n1 * 2 * L + l0, B1);
        run_load<TL>(bct + nn * 2 * L + L + l0, C0);
        run_load<TL>(bct + n1 * 2 * L + L + l0, C1);
        float P0 = 1.f, S0 = 0.f, P1 = 1.f, S1 = 0.f;
        float pl0[TL], hl0[TL], pl1[TL], hl1[TL];
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            const float a0 = fast_exp2(dl[i] * A0), a1 = fast_exp2(dl[i] * A1);
            S0 = fmaf(a0, S0, dlu[i] * B0[i]);
            S1 = fmaf(a1, S1, dlu[i] * B1[i]);
            P0 *= a0;
            P1 *= a1;
            pl0[i] = P0; hl0[i] = S0;
            pl1[i] = P1; hl1[i] = S1;
        }
        wave_scan_affine_x2(P0, S0, P1, S1);
        const float h0 = wave_shift_up1(S0, 0.f), h1 = wave_shift_up1(S1, 0.f);   // state entering this lane's tokens
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            yacc[i] = fmaf(C0[i], fmaf(pl0[i], h0, hl0[i]), yacc[i]);
            if (two) yacc[i] = fmaf(C1[i], fmaf(pl1[i], h1, hl1[i]), yacc[i]);
        }
    }
    SMALL_STAMP(0, 3);
    // (every wave has left the cooperative reads of ut behind the last barrier: its own row is free for out_z)
    {
        float oz[TL];
#pragma unroll
        for (int i = 0; i < TL; ++i) oz[i] = yacc[i] * z[i] * sigmoidf_(z[i]);
        run_store<TL>(ut + d * L + l0, oz);
    }
    __syncthreads();

    // out_proj, inverse zig-zag, coordinates: a token per thread and step
    float dummy;
    const float wgt = coord_weight(cwv.altho, dummy);
    constexpr int c = K / 2;
    float *yb = p.y + ((long)part * p.B + b) * K * p.Lr;
    for (int l = tid; l < p.Lr; l += blockDim.x) {
        int h, ww;
        unzig(l, p.H, p.W, p.lw, h, ww);
        float oz[D], off[K], cum[K];
#pragma unroll
        for (int dd = 0; dd < D; ++dd) oz[dd] = ut[dd * L + l];
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
            for (int dd = 0; dd < D; ++dd) sq = fmaf(wout[k * D + dd], oz[dd], sq);
            yb[(k * p.H + h) * p.W + ww] = fmaf(wgt, sq, first ? (float)h + p.scope * cum[k] : 0.f);
        }
    }
    SMALL_STAMP(0, 4);
}

// ================================================================================================================
// backward
// ================================================================================================================
// sums NV per-lane values over the wave, four at a time; dst[i] receives value i (written by one of lanes 12..15)
template <int NV>
__device__ __forceinline__ void wave_sums_scatter(const float (&v)[NV], float *wpart, const int (&dst)[NV]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; i += 4) {
        const float r = wave_sum4_swap(v[i], i + 1 < NV ? v[i + 1] : 0.f, i + 2 < NV ? v[i + 2] : 0.f,
                                       i + 3 < NV ? v[i + 3] : 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (i + q < NV && lane == 12 + q) wpart[dst[i + q]] = r;
    }
}

template <int K, int TL>
__global__ __launch_bounds__(128 * K) void mamba_small_bwd_kernel(SmallArgs p, W_PARAMS) {
    constexpr int D = 2 * K, RS = WRec<D>::RS;
    extern __shared__ __align__(16) float smem[];
    constexpr int L = 64 * TL;
    const int N = p.N, npp = p.npp;
    const GradLayout<K> lay(N);
    const int NV = lay.total;
    float *yoff = smem;                 // [K][L]      y_off, zig-zag order
    float *dyz = yoff + K * L;          // [K][L]      d rows, zig-zag order
    float *ut = dyz + K * L;            // [D][L]      u
    float *dtt = ut + D * L;            // [L]         dt row
    float *bct = dtt + L;               // [2][2][L]   B_n, C_n, double-buffered over the states
    float *slab = bct + 4 * L;          // [2D][L + 4] per-channel exchange: dB_d | dC_d, then draw, dp, d xz
    float *sums = slab + 2 * D * (L + 4);   // [2][L]  dB_n, dC_n summed over the channels; later d dt
    float *wl = sums + 2 * L;           // [npp][RS]
    float *wpart = wl + npp * RS;       // [NV + 1]    this workgroup's weight-gradient vector (+ a scratch word)
    const int tid = threadIdx.x, lane = tid & 63;
    const int d = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, part = blockIdx.y;
    const int n0 = part * npp;
    const bool first = part == 0;
    SMALL_STAMP(1, 0);
    const ChanW<K> cwv = load_chan<K>(d, first, W_ARGS);
    stage_zig<K, TL>(p, p.off + (long)b * 2 * K * p.Lr, yoff);
    stage_zig<K, TL>(p, p.dy + (long)b * K * p.Lr, dyz);
    stage_wl<D>(wl, wx, Aw, N, n0, npp);
    for (int i = tid; i < NV; i += blockDim.x) wpart[i] = 0.f;   // (the other parts' rows of dW_x / dA stay 0)
    __syncthreads();
    SMALL_STAMP(1, 1);

    const int l0 = lane * TL;
    float dwgt_da;
    const float wgt = coord_weight(cwv.altho, dwgt_da);
    const float wdv = cwv.wdt, dbv = cwv.dtb, Dv = cwv.Dv;
    // Across the state loop a lane keeps six values per token (u, delta, dy, and the three accumulators); x, z, the
    // conv pre-activation and d out_z are formed again behind the loop (TL = 16: 256 registers would not hold them)
    float u[TL], dl[TL], dyv[TL], du[TL], ddl[TL], yacc[TL];
    auto d_out_z = [&](float (&doz)[TL]) {
#pragma unroll
        for (int i = 0; i < TL; ++i) doz[i] = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float g[TL];
            run_load<TL>(dyz + k * L + l0, g);
            const float wo = wgt * cwv.wo[k];
#pragma unroll
            for (int i = 0; i < TL; ++i) doz[i] = fmaf(wo, g[i], doz[i]);
        }
    };
    {
        float x[TL + 3], z[TL], pp[TL], doz[TL];
        pre_channel<K, TL>(cwv, yoff, L, l0, x, z, pp, u);
        run_store<TL>(ut + d * L + l0, u);
        d_out_z(doz);
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            dyv[i] = doz[i] * z[i] * sigmoidf_(z[i]);            // d y = d out_z * silu(z)
            du[i] = Dv * dyv[i];
            ddl[i] = 0.f;
            yacc[i] = Dv * u[i];
        }
    }
    __syncthreads();
    coop_dt<D>(ut, dtt, cwv.w0, L);
    coop_bc<D>(ut, bct, wl, L);                 // state 0 of this part; state nn + 1 is formed during round nn
    __syncthreads();
    SMALL_STAMP(1, 2);
    {
        float dtv[TL];
        run_load<TL>(dtt + l0, dtv);
#pragma unroll
        for (int i = 0; i < TL; ++i) dl[i] = softplus_thr(fmaf(wdv, dtv[i], dbv));
    }
    SMALL_STAMP(1, 3);

    for (int nn = 0; nn < npp; ++nn) {
        const int n = n0 + nn;
        const float *wr = wl + nn * RS;
        const float Ar = wr[2 * D + d], A2 = Ar * MMU_LOG2E;
        float Bn[TL], Cn[TL], a[TL];
        run_load<TL>(bct + (nn & 1) * 2 * L + l0, Bn);
        run_load<TL>(bct + (nn & 1) * 2 * L + L + l0, Cn);
        // (1) the lane's tokens as two affine maps: h forward (P, S), and the adjoint gh_t = a_t g_t,
        //     g_t = C_t dy_t + gh_{t+1} backward (Q, R)
        float P = 1.f, S = 0.f, Q = 1.f, R = 0.f;
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            a[i] = fast_exp2(dl[i] * A2);
            S = fmaf(a[i], S, dl[i] * u[i] * Bn[i]);
            P *= a[i];
        }
#pragma unroll
        for (int i = TL - 1; i >= 0; --i) {
            R = a[i] * fmaf(Cn[i], dyv[i], R);
            Q *= a[i];
        }
        // the adjoint runs from the last lane to the first: reverse the lanes, scan forward, reverse back
        Q = wave_reverse(Q);
        R = wave_reverse(R);
        wave_scan_affine_x2(P, S, Q, R);
        float hp = wave_shift_up1(S, 0.f);                    // state entering this lane's first token
        float gh = wave_reverse(wave_shift_up1(R, 0.f));      // gh of the token after this lane's last
        // (2) h forward, then the adjoint backward with everything that needs it
        //     dB_n / dC_n are sums over the channels: this channel's rows go to LDS, all threads add them up in (3)
        float hm[TL];
        {
            float dCn[TL];
#pragma unroll
            for (int i = 0; i < TL; ++i) {
                hm[i] = a[i] * hp;
                hp = fmaf(dl[i] * u[i], Bn[i], hm[i]);
                yacc[i] = fmaf(Cn[i], hp, yacc[i]);
                dCn[i] = dyv[i] * hp;
            }
            run_store<TL>(slab + (D + d) * (L + 4) + l0, dCn);
        }
        float da = 0.f;
        {
            float dBn[TL];
#pragma unroll
            for (int i = TL - 1; i >= 0; --i) {
                const float g = fmaf(Cn[i], dyv[i], gh);
                gh = a[i] * g;
                const float t0 = g * dl[i];
                dBn[i] = t0 * u[i];
                du[i] = fmaf(t0, Bn[i], du[i]);
                da = fmaf(t0, hm[i], da);
                ddl[i] = fmaf(g, fmaf(u[i], Bn[i], Ar * hm[i]), ddl[i]);
            }
            run_store<TL>(slab + d * (L + 4) + l0, dBn);
        }
        // (3)
        __syncthreads();
        for (int l = tid; l < L; l += blockDim.x) {
            float sb = 0.f, sc = 0.f;
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                sb += slab[dd * (L + 4) + l];
                sc += slab[(D + dd) * (L + 4) + l];
            }
            sums[l] = sb;
            sums[L + l] = sc;
        }
        if (nn + 1 < npp) coop_bc<D>(ut, bct + ((nn + 1) & 1) * 2 * L, wl + (nn + 1) * RS, L);
        __syncthreads();
        // (4) back into u through the x_proj rows 1 + n and 1 + N + n, and this channel's entries of their gradients
        {
            const float wb = wr[d], wc = wr[D + d];
            float sb = 0.f, sc = 0.f, gb[TL], gc[TL];
            run_load<TL>(sums + l0, gb);
            run_load<TL>(sums + L + l0, gc);
#pragma unroll
            for (int i = 0; i < TL; ++i) {
                du[i] = fmaf(wb, gb[i], fmaf(wc, gc[i], du[i]));
                sb = fmaf(gb[i], u[i], sb);
                sc = fmaf(gc[i], u[i], sc);
            }
            const float v3[3] = {sb, sc, da};
            const int dst[3] = {lay.wx + (1 + n) * D + d, lay.wx + (1 + N + n) * D + d, lay.A + d * N + n};
            wave_sums_scatter<3>(v3, wpart, dst);
        }
    }
    SMALL_STAMP(1, 4);

    // ---- behind the scan -------------------------------------------------------------------------------------------
    // gate: out_z (for d out_proj), dz; softplus -> d raw delta; the dt row needs the sum over the channels
    float dz[TL], wv[3 * K + 10];
#pragma unroll
    for (int i = 0; i < 3 * K + 10; ++i) wv[i] = 0.f;
    float x[TL + 3], z[TL], pp[TL], dtv[TL];
    {
        float u2[TL], doz[TL];
        MMU_COMPILER_FENCE();          // (formed again, not kept alive across the loop)
        pre_channel<K, TL>(cwv, yoff, L, l0, x, z, pp, u2);
        run_load<TL>(dtt + l0, dtv);
        d_out_z(doz);
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            const float sg = sigmoidf_(z[i]);
            dz[i] = doz[i] * yacc[i] * sg * (1.f + z[i] * (1.f - sg));
        }
    }
    // wv: [0,K) dWout[.][d] | K dD | K+1 dWdt | K+2 dbias | K+3 dWx row 0 | K+4..K+7 dcw | K+8 dcb |
    //     K+9..2K+8 dWin row d | 2K+9..3K+8 dWin row D+d | 3K+9 unused
    {
        float ozv[TL], wdr[TL];
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            ozv[i] = yacc[i] * z[i] * sigmoidf_(z[i]);
            if (first) wv[K] = fmaf(dyv[i], u[i], wv[K]);
            const float raw = fmaf(wdv, dtv[i], dbv);
            const float draw = ddl[i] * (raw <= 20.f ? sigmoidf_(raw) : 1.f);   // softplus' (threshold 20)
            wv[K + 1] = fmaf(draw, dtv[i], wv[K + 1]);
            wv[K + 2] += draw;
            wdr[i] = wdv * draw;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float g[TL];
            run_load<TL>(dyz + k * L + l0, g);
#pragma unroll
            for (int i = 0; i < TL; ++i) wv[k] = fmaf(wgt * g[i], ozv[i], wv[k]);
        }
        run_store<TL>(slab + d * (L + 4) + l0, wdr);
    }
    __syncthreads();
    float *ddtt = sums;                // d dt [L]
    for (int l = tid; l < L; l += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int dd = 0; dd < D; ++dd) s += slab[dd * (L + 4) + l];
        ddtt[l] = s;
    }
    __syncthreads();
    SMALL_STAMP(1, 5);
    // dt row of x_proj back into u; conv1d backward (this channel's dp: the next lane's first three through LDS)
    {
        const float w0 = cwv.w0[d];
        const float c0 = cwv.c0, c1 = cwv.c1, c2 = cwv.c2, c3 = cwv.c3;
        float *dpr = slab + d * (L + 4);
        if (lane < 4) dpr[L + lane] = 0.f;      // tokens past the end
        float dp[TL + 3], ddt[TL];
        run_load<TL>(ddtt + l0, ddt);
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            du[i] = fmaf(w0, ddt[i], du[i]);
            wv[K + 3] = fmaf(ddt[i], u[i], wv[K + 3]);
            const float sg = sigmoidf_(pp[i]);
            dp[i] = du[i] * sg * (1.f + pp[i] * (1.f - sg));
            wv[K + 8] += dp[i];
            wv[K + 4] = fmaf(x[i], dp[i], wv[K + 4]);
            wv[K + 5] = fmaf(x[i + 1], dp[i], wv[K + 5]);
            wv[K + 6] = fmaf(x[i + 2], dp[i], wv[K + 6]);
            wv[K + 7] = fmaf(x[i + 3], dp[i], wv[K + 7]);
        }
        {
            float own[TL];
#pragma unroll
            for (int i = 0; i < TL; ++i) own[i] = dp[i];
            run_store<TL>(dpr + l0, own);
        }
        // the next lane's first three.  The wave reads what the wave wrote: the LDS executes a wave's operations in
        // order -- but the compiler sees a thread's stores to [l0, l0 + TL) and loads from [l0 + TL, ...) as independent
        // and is free to hoist the loads (it did, for some TL: d offset off by a few per cent); the fence pins the order.
        MMU_COMPILER_FENCE();
#pragma unroll
        for (int j = 0; j < 3; ++j) dp[TL + j] = dpr[l0 + TL + j];
        MMU_COMPILER_FENCE();     // ... and the stores of d x below (same row) stay behind these loads
        float dx[TL];
#pragma unroll
        for (int i = 0; i < TL; ++i) dx[i] = fmaf(c3, dp[i], fmaf(c2, dp[i + 1], fmaf(c1, dp[i + 2], c0 * dp[i + 3])));
        // in_proj: this channel's two rows of d xz; their weight gradients; d y_off is a sum over all rows
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float yo[TL];
            run_load<TL>(yoff + k * L + l0, yo);
#pragma unroll
            for (int i = 0; i < TL; ++i) {
                wv[K + 9 + k] = fmaf(dx[i], yo[i], wv[K + 9 + k]);
                wv[2 * K + 9 + k] = fmaf(dz[i], yo[i], wv[2 * K + 9 + k]);
            }
        }
        // (row d holds this wave's own dp, row D + d nothing that is still read: no barrier needed before the stores)
        run_store<TL>(slab + d * (L + 4) + l0, dx);
        run_store<TL>(slab + (D + d) * (L + 4) + l0, dz);
    }
    __syncthreads();
    SMALL_STAMP(1, 6);
    {
        constexpr int c = K / 2;
        float *db = p.doff + ((long)part * p.B + b) * K * p.Lr;
        for (int l = tid; l < p.Lr; l += blockDim.x) {
            float g[K];
#pragma unroll
            for (int k = 0; k < K; ++k) g[k] = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * D; ++j) {
                const float v = slab[j * (L + 4) + l];
#pragma unroll
                for (int k = 0; k < K; ++k) g[k] = fmaf(win[j * K + k], v, g[k]);
            }
            // the coordinate terms: tap j > c feeds cum[k] for k >= j, tap j < c for k <= j
            if (first) {
                float run = 0.f;
#pragma unroll
                for (int j = K - 1; j > c; --j) {
                    run += dyz[j * L + l];
                    g[j] = fmaf(p.scope, run, g[j]);
                }
                run = 0.f;
#pragma unroll
                for (int j = 0; j < c; ++j) {
                    run += dyz[j * L + l];
                    g[j] = fmaf(p.scope, run, g[j]);
                }
            }
            int h, ww;
            unzig(l, p.H, p.W, p.lw, h, ww);
#pragma unroll
            for (int k = 0; k < K; ++k) db[(k * p.H + h) * p.W + ww] = g[k];
        }
    }
    SMALL_STAMP(1, 7);
    // this channel's rows of the weight gradients
    {
        int dst[3 * K + 10];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            dst[k] = lay.wout + k * D + d;
            dst[K + 9 + k] = lay.win + d * K + k;
            dst[2 * K + 9 + k] = lay.win + (D + d) * K + k;
        }
        dst[K] = first ? lay.Dp + d : NV;      // (NV: the scratch word behind the vector)
        dst[K + 1] = lay.wdt + d;
        dst[K + 2] = lay.dtb + d;
        dst[K + 3] = lay.wx + d;
#pragma unroll
        for (int m = 0; m < 4; ++m) dst[K + 4 + m] = lay.cw + d * 4 + m;
        dst[K + 8] = lay.cb + d;
        dst[3 * K + 9] = NV;
        wave_sums_scatter<3 * K + 10>(wv, wpart, dst);
    }
    SMALL_STAMP(1, 8);
    __syncthreads();
    // d altho: rows = wgt * seq + ..., seq = out_proj(out_z)  =>  d wgt = sum_{k,d} Wout[k][d] * (dWout[k][d] / wgt)
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < K * D; ++i) s = fmaf(wout[i], wpart[lay.wout + i], s);
        wpart[lay.altho] = s / wgt * dwgt_da;
    }
    __syncthreads();
    float *pb = p.part + ((long)b * p.ns + part) * NV;
    for (int i = tid; i < NV; i += blockDim.x) pb[i] = wpart[i];
    SMALL_STAMP(1, 9);
}

// Workgroups [0, nbw): dweights[i] = sum over the B * ns slots of part[slot][i] -- 16 outputs per workgroup, the slots
// dealt to 16 groups of threads whose partial sums meet in LDS in fixed order (a single thread walking 64+ slots waited
// for every load in turn: 10 us).  The rest: d offset [B, 2K, H, W] = sum over the ns parts of dparts[part][b][k][pixel]
// for the first K channels, 0 for the others.  Fixed order: bit-reproducible.
__global__ __launch_bounds__(256) void mamba_small_reduce_kernel(const float *__restrict__ part, float *__restrict__ out,
                                                                 int slots, int NV, int nbw,
                                                                 const float *__restrict__ dparts,
                                                                 float *__restrict__ doff, int B, int K, int L, int ns) {
    __shared__ float red[16][17];
    if ((int)blockIdx.x < nbw) {
        const int x = threadIdx.x & 15, g = threadIdx.x >> 4;
        const int i = blockIdx.x * 16 + x;
        float s = 0.f;
        if (i < NV) {
#pragma unroll 8
            for (int b = g; b < slots; b += 16) s += part[(long)b * NV + i];
        }
        red[g][x] = s;
        __syncthreads();
        if (g == 0 && i < NV) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += red[q][x];
            out[i] = t;
        }
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
#pragma unroll 8
        for (int j = 0; j < ns; ++j) s += src[j * ps];
    }
    doff[idx] = s;
}

// tokens per lane: the smallest TL in {1, 2, 4, 6, 8, 12, 16} with 64 * TL >= L (L = height * width <= 1,024: any
// map, power-of-two sides or not -- 19 x 19 = 361 tokens of the reference's 608 x 608 training resolution run as
// TL = 6 with 23 padding slots)
bool plan(int L, int &TL) {
    static const int tls[] = {1, 2, 4, 6, 8, 12, 16};
    for (int t : tls) {
        if (L <= 64 * t) {
            TL = t;
            return L >= 1;
        }
    }
    return false;
}

int grad_total(int K, int N) { return K == 3 ? GradLayout<3>(N).total : GradLayout<1>(N).total; }

size_t fwd_lds_floats(int K, int L, int npp) {
    const int D = 2 * K, RS = (3 * D + 3) & ~3;
    return (size_t)K * L + (size_t)D * L + L + (size_t)npp * 2 * L + (size_t)npp * RS;
}
size_t bwd_lds_floats(int K, int L, int npp, int NV) {
    const int D = 2 * K, RS = (3 * D + 3) & ~3;
    return (size_t)2 * K * L + (size_t)D * L + L + (size_t)4 * L + (size_t)2 * D * (L + 4) + (size_t)2 * L +
           (size_t)npp * RS + NV + 4;
}

// State-range parts per batch item: the states are independent up to the final sums over n, so part j takes states
// [j N / ns, (j + 1) N / ns).  Enough workgroups to spread over the chip, two states per part at least where possible
// (the forward interleaves two lane scans).  MMU_SMALL_PARTS overrides (tuning / tests).
int default_parts(int batch, int K, int L, int N, int backward) {
    static const int forced = []() { const char *e = getenv("MMU_SMALL_PARTS"); return e ? atoi(e) : 0; }();
    static const int forced_b = []() { const char *e = getenv("MMU_SMALL_PARTS_BWD"); return e ? atoi(e) : 0; }();
    // forward: two states per part (its two lane scans are interleaved); backward: a round per state with two
    // workgroup barriers each, so as few states per part as the chip has room for (measured at batch 8, 32 x 32:
    // 2 / 4 / 8 / 16 parts = 46 / 34 / 28 / 25 us)
    int ns = backward ? (forced_b > 0 ? forced_b : 16) : (forced > 0 ? forced : 8);
    if (ns > N) ns = N;
    while (ns > 1 && (N % ns != 0 || (long)batch * ns > 1024)) --ns;
    if (ns < 1) ns = 1;
    // the forward keeps the B_n / C_n rows of all its states in LDS: more parts until they fit
    while (ns < N && (N % ns != 0 || fwd_lds_floats(K, L, N / ns) * sizeof(float) > 160 * 1024)) ++ns;
    return ns;
}

int check(const mmu_mamba_small_params *p, const char *name, int &TL) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->taps == 1 || p->taps == 3, "%s: 1 or 3 taps supported (got %d)", name, p->taps);
    MMU_CHECK(p->batch > 0 && p->height > 0 && p->width > 0, "%s: empty tensor", name);
    MMU_CHECK(p->dstate >= 1 && p->dstate <= 64, "%s: d_state must be in 1..64 (got %d)", name, p->dstate);
    MMU_CHECK(plan(p->height * p->width, TL), "%s: height * width must be at most 1,024 (got %d x %d)", name, p->height,
              p->width);
    MMU_CHECK(p->parts >= 1 && p->dstate % p->parts == 0, "%s: parts (%d) must divide d_state (%d)", name, p->parts,
              p->dstate);
    MMU_CHECK(p->offset && p->in_proj_weight && p->conv_weight && p->x_proj_weight && p->dt_proj_weight && p->A &&
                  p->out_proj_weight && p->altho,
              "%s: offset, in_proj / conv / x_proj / dt_proj / out_proj weights, A and altho are required", name);
    return 0;
}

SmallArgs to_args(const mmu_mamba_small_params *p) {
    SmallArgs a = {};
    a.B = p->batch; a.H = p->height; a.W = p->width; a.N = p->dstate;
    a.lw = -1;
    if ((p->width & (p->width - 1)) == 0) {
        a.lw = 0;
        while ((1 << a.lw) < p->width) ++a.lw;
    }
    a.Lr = p->height * p->width;
    a.ns = p->parts; a.npp = p->dstate / p->parts;
    a.scope = p->extend_scope;
    a.off = p->offset; a.y = p->y; a.dy = p->dy;
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

#define SMALL_TL(KERNEL, K_, ...)                        \
    switch (TL) {                                        \
        case 1: KERNEL(K_, 1, __VA_ARGS__); break;       \
        case 2: KERNEL(K_, 2, __VA_ARGS__); break;       \
        case 4: KERNEL(K_, 4, __VA_ARGS__); break;       \
        case 6: KERNEL(K_, 6, __VA_ARGS__); break;       \
        case 8: KERNEL(K_, 8, __VA_ARGS__); break;       \
        case 12: KERNEL(K_, 12, __VA_ARGS__); break;     \
        default: KERNEL(K_, 16, __VA_ARGS__); break;     \
    }
#define SMALL_DISPATCH(KERNEL, ...)                      \
    do {                                                 \
        if (p->taps == 3) {                              \
            SMALL_TL(KERNEL, 3, __VA_ARGS__)             \
        } else {                                         \
            SMALL_TL(KERNEL, 1, __VA_ARGS__)             \
        }                                                \
    } while (0)
#define W_CALL p->in_proj_weight, p->conv_weight, p->conv_bias, p->x_proj_weight, p->dt_proj_weight, p->dt_bias, p->A, \
               p->D, p->out_proj_weight, p->altho

}  // namespace

#ifdef MMU_SMALL_STAMPS
extern "C" int mmu_debug_small_stamps(unsigned long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_small_stamps), sizeof(g_small_stamps)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int mmu_mamba_small_supported(int taps, int height, int width, int dstate) {
    int TL;
    return (taps == 1 || taps == 3) && dstate >= 1 && dstate <= 64 && height > 0 && width > 0 && plan(height * width, TL);
}

extern "C" int mmu_mamba_small_parts(int batch, int taps, int height, int width, int dstate, int backward) {
    int TL = 16;
    plan(height * width, TL);
    return default_parts(batch, taps, 64 * TL, dstate, backward);
}

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
    int TL;
    if (int r = check(p, "mamba_small_fwd", TL)) return r;
    MMU_CHECK(p->y != nullptr, "mamba_small_fwd: y is required");
    const SmallArgs a = to_args(p);
    const size_t lds = sizeof(float) * fwd_lds_floats(p->taps, 64 * TL, a.npp);
    hipStream_t st = (hipStream_t)stream;
#define SMALL_FWD(K_, T_, a_)                                                                         \
    if (int r = set_lds_attr(mamba_small_fwd_kernel<K_, T_>, lds, "mamba_small_fwd")) return r;       \
    mamba_small_fwd_kernel<K_, T_><<<dim3(a_.B, a_.ns), 128 * K_, lds, st>>>(a_, W_CALL)
    SMALL_DISPATCH(SMALL_FWD, a);
#undef SMALL_FWD
    MMU_HIP_LAUNCH_CHECK("mamba_small_fwd");
    return 0;
}

extern "C" int mmu_mamba_small_bwd(const mmu_mamba_small_params *p, void *stream) {
    int TL;
    if (int r = check(p, "mamba_small_bwd", TL)) return r;
    MMU_CHECK(p->dy && p->doffset && p->workspace && p->dweights,
              "mamba_small_bwd: dy, doffset, workspace and dweights are required");
    SmallArgs a = to_args(p);
    const int K = p->taps;
    const int NV = grad_total(K, a.N);
    // workspace: [B * parts][NV] weight-gradient partials, then [parts][B][K][L] partial d offset
    a.part = p->workspace;
    a.doff = p->workspace + (size_t)a.B * a.ns * NV;
    const size_t lds = sizeof(float) * bwd_lds_floats(K, 64 * TL, a.npp, NV);
    hipStream_t st = (hipStream_t)stream;
#define SMALL_BWD(K_, T_, a_)                                                                         \
    if (int r = set_lds_attr(mamba_small_bwd_kernel<K_, T_>, lds, "mamba_small_bwd")) return r;       \
    mamba_small_bwd_kernel<K_, T_><<<dim3(a_.B, a_.ns), 128 * K_, lds, st>>>(a_, W_CALL)
    SMALL_DISPATCH(SMALL_BWD, a);
#undef SMALL_BWD
    MMU_HIP_LAUNCH_CHECK("mamba_small_bwd");
    const int nbw = (NV + 15) / 16;
    const int L = p->height * p->width;
    const long nd = (long)a.B * 2 * K * L;
    mamba_small_reduce_kernel<<<nbw + (unsigned)((nd + 255) / 256), 256, 0, st>>>(a.part, p->dweights, a.B * a.ns, NV, nbw,
                                                                                 a.doff, p->doffset, a.B, K, L, a.ns);
    MMU_HIP_LAUNCH_CHECK("mamba_small_reduce");
    return 0;
}
