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

namespace {

struct SmallArgs {
    int B, H, W, L, N, nw;
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
    float *y;            // [B, K, H, W]
    float *hstate;       // [B][2K*N][L/T] or null (forward: written; backward: read)
    const float *dy;     // [B, K, H, W]
    float *doff;         // [B, 2K, H, W]
    float *part;         // [B][NV] weight-gradient partials
};

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
template <int K>
__device__ __forceinline__ void stage_yoff(const SmallArgs &p, int b, float *yoff) {
    const int L = p.L;
    const float *ob = p.off + (long)b * 2 * K * L;   // the first K channels of the batch item are contiguous
    for (int idx = threadIdx.x; idx < K * L; idx += blockDim.x) {
        const int k = idx / L, r = idx - k * L;
        const int h = r / p.W, ww = r - h * p.W;
        yoff[k * L + zig_of(h, ww, p.H, p.W)] = ob[idx];
    }
}

// in_proj, conv1d + SiLU, dt row, dt_proj + softplus for the T tokens l0 .. l0 + T - 1 of this lane
template <int K, int T, bool KEEP_PRE>
__device__ __forceinline__ void pre_phase(const SmallArgs &p, const float *yoff, int l0, float (&xs)[2 * K][T + 3],
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
            for (int k = 0; k < K; ++k) a = fmaf(p.win[d * K + k], yo[k], a);
            xs[d][j] = a;
            if (j >= 3) {
                float c = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) c = fmaf(p.win[(D + d) * K + k], yo[k], c);
                z[d][j - 3] = c;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < T; ++i) dt[i] = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float bv = p.cb ? p.cb[d] : 0.f;
        const float w0 = p.wx[d];
#pragma unroll
        for (int i = 0; i < T; ++i) {
            float acc = bv;
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = fmaf(p.cw[d * 4 + m], xs[d][i + m], acc);
            if (KEEP_PRE) pp[d][i] = acc;
            u[d][i] = acc * sigmoidf_(acc);
            dt[i] = fmaf(w0, u[d][i], dt[i]);
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float wv = p.wdt[d], bv = p.dtb ? p.dtb[d] : 0.f;
#pragma unroll
        for (int i = 0; i < T; ++i) dl[d][i] = softplus_thr(fmaf(wv, dt[i], bv));
    }
}

// The carry between waves: wave w waits until wave w - 1 has published steps s and s + 1, reads them, adds its own
// aggregates and publishes.  LDS operations of one wave execute in order, so "data, then progress word" needs no
// fence beyond the wait for the data write; a reader that sees the progress word sees the data.
__device__ __forceinline__ void carry_pair(volatile float *hcar, volatile int *prog, int w, int nw, int DN, int s,
                                           float Pt0, float St0, float Pt1, float St1, float &in0, float &in1) {
    in0 = 0.f;
    in1 = 0.f;
    if (w > 0) {
        while (prog[w - 1] < s + 2) __builtin_amdgcn_s_sleep(1);
        in0 = hcar[(w - 1) * DN + s];
        in1 = hcar[(w - 1) * DN + s + 1];
    }
    if (w + 1 < nw) {
        if ((threadIdx.x & 63) == 0) {
            hcar[w * DN + s] = fmaf(Pt0, in0, St0);
            hcar[w * DN + s + 1] = fmaf(Pt1, in1, St1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            prog[w] = s + 2;
        }
    }
}

// ================================================================================================================
// forward
// ================================================================================================================
template <int K, int T>
__global__ __launch_bounds__(512) void mamba_small_fwd_kernel(SmallArgs p) {
    constexpr int D = 2 * K;
    extern __shared__ float smem[];
    const int L = p.L, N = p.N, DN = D * N, G = L / T;
    float *yoff = smem;                                         // [K][L]
    volatile float *hcar = smem + K * L;                        // [nw][DN]
    volatile int *prog = (volatile int *)(smem + K * L + p.nw * DN);   // [nw]
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    stage_yoff<K>(p, b, yoff);
    if (tid < p.nw) prog[tid] = 0;
    __syncthreads();

    const int l0 = tid * T;
    float xs[D][T + 3], z[D][T], pp[D][T], u[D][T], dl[D][T], dt[T];
    pre_phase<K, T, false>(p, yoff, l0, xs, z, pp, u, dl, dt);
    float dlu[D][T], yacc[D][T];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float Dv = p.Dp ? p.Dp[d] : 0.f;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            dlu[d][i] = dl[d][i] * u[d][i];
            yacc[d][i] = Dv * u[d][i];
        }
    }
    float *hs = p.hstate ? p.hstate + (long)b * DN * G + tid : nullptr;

    for (int n = 0; n < N; ++n) {
        float Bn[T], Cn[T];
#pragma unroll
        for (int i = 0; i < T; ++i) Bn[i] = Cn[i] = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float wb = p.wx[(1 + n) * D + d], wc = p.wx[(1 + N + n) * D + d];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                Bn[i] = fmaf(wb, u[d][i], Bn[i]);
                Cn[i] = fmaf(wc, u[d][i], Cn[i]);
            }
        }
#pragma unroll
        for (int dq = 0; dq < D; dq += 2) {
            const float A0 = p.A[dq * N + n] * MMU_LOG2E, A1 = p.A[(dq + 1) * N + n] * MMU_LOG2E;
            float P0 = 1.f, S0 = 0.f, P1 = 1.f, S1 = 0.f;
            float pl0[T], hl0[T], pl1[T], hl1[T];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const float a0 = fast_exp2(dl[dq][i] * A0), a1 = fast_exp2(dl[dq + 1][i] * A1);
                S0 = fmaf(a0, S0, dlu[dq][i] * Bn[i]);
                S1 = fmaf(a1, S1, dlu[dq + 1][i] * Bn[i]);
                P0 *= a0;
                P1 *= a1;
                pl0[i] = P0; hl0[i] = S0;
                pl1[i] = P1; hl1[i] = S1;
            }
            wave_scan_affine_x2(P0, S0, P1, S1);
            const float Pt0 = readlane63(P0), St0 = readlane63(S0), Pt1 = readlane63(P1), St1 = readlane63(S1);
            const float Pe0 = wave_shift_up1(P0, 1.f), Se0 = wave_shift_up1(S0, 0.f);
            const float Pe1 = wave_shift_up1(P1, 1.f), Se1 = wave_shift_up1(S1, 0.f);
            const int s = n * D + dq;
            float in0, in1;
            carry_pair(hcar, prog, w, p.nw, DN, s, Pt0, St0, Pt1, St1, in0, in1);
            const float h0 = fmaf(Pe0, in0, Se0), h1 = fmaf(Pe1, in1, Se1);   // state entering this lane's tokens
            if (hs) {
                hs[(long)s * G] = h0;
                hs[(long)(s + 1) * G] = h1;
            }
#pragma unroll
            for (int i = 0; i < T; ++i) {
                yacc[dq][i] = fmaf(Cn[i], fmaf(pl0[i], h0, hl0[i]), yacc[dq][i]);
                yacc[dq + 1][i] = fmaf(Cn[i], fmaf(pl1[i], h1, hl1[i]), yacc[dq + 1][i]);
            }
        }
    }

    // gate, out_proj, inverse zig-zag, coordinates
    float dummy;
    const float wgt = coord_weight(p.altho[0], dummy);
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
            for (int d = 0; d < D; ++d) sq = fmaf(p.wout[k * D + d], oz[d], sq);
            p.y[(((long)b * K + k) * p.H + h) * p.W + ww] = fmaf(wgt, sq, (float)h + p.scope * cum[k]);
        }
    }
}

// ================================================================================================================
// backward
// ================================================================================================================
// Sums NV per-lane values over the wave, four at a time, into slot[0 .. NV) (one per value; the lanes 12..15 that
// hold a batch's results write them).
template <int NV>
__device__ __forceinline__ void wave_sums_to(const float (&v)[NV], volatile float *slot) {
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
__global__ __launch_bounds__(512) void mamba_small_bwd_kernel(SmallArgs p) {
    constexpr int D = 2 * K;
    extern __shared__ float smem[];
    const int L = p.L, N = p.N, DN = D * N, G = L / T;
    const GradLayout<K> lay(N);
    const int NV = lay.total;
    float *yoff = smem;                                          // [K][L]
    float *dpl = smem + K * L;                                   // [D][L + 4]   conv1d backward exchange
    volatile float *hcar = dpl + D * (L + 4);                    // [nw][DN]
    volatile float *wpart = hcar + p.nw * DN;                    // [nw][NV]
    volatile int *prog = (volatile int *)(wpart + p.nw * NV);    // [nw]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    stage_yoff<K>(p, b, yoff);
    if (tid < p.nw) prog[tid] = 0;
    for (int i = tid; i < D * 4; i += blockDim.x) dpl[(i >> 2) * (L + 4) + L + (i & 3)] = 0.f;   // tokens past the end
    __syncthreads();

    // lanes take the token groups in REVERSE order: the adjoint scan runs forward over lanes and waves
    const int gr = G - 1 - tid;
    const int l0 = gr * T;
    float xs[D][T + 3], z[D][T], pp[D][T], u[D][T], dl[D][T], dt[T];
    pre_phase<K, T, true>(p, yoff, l0, xs, z, pp, u, dl, dt);

    float dwgt_da;
    const float wgt = coord_weight(p.altho[0], dwgt_da);
    // incoming gradient of the row coordinates, through out_proj and the gate
    float dyr[K][T], doz[D][T], dyv[D][T], sz[D][T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        int h, ww;
        unzig(l0 + i, p.H, p.W, h, ww);
#pragma unroll
        for (int k = 0; k < K; ++k) dyr[k][i] = p.dy[(((long)b * K + k) * p.H + h) * p.W + ww];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) s = fmaf(p.wout[k * D + d], dyr[k][i], s);
            doz[d][i] = wgt * s;
            sz[d][i] = sigmoidf_(z[d][i]);
            dyv[d][i] = doz[d][i] * z[d][i] * sz[d][i];     // d out_z * silu(z) = gradient of y
        }
    }
    float dlu[D][T], du[D][T], ddl[D][T], yacc[D][T];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float Dv = p.Dp ? p.Dp[d] : 0.f;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            dlu[d][i] = dl[d][i] * u[d][i];
            du[d][i] = Dv * dyv[d][i];
            ddl[d][i] = 0.f;
            yacc[d][i] = Dv * u[d][i];
        }
    }
    const float *hs = p.hstate + (long)b * DN * G + gr;
    volatile float *myw = wpart + w * NV;

    for (int n = 0; n < N; ++n) {
        float Bn[T], Cn[T], dBn[T], dCn[T];
#pragma unroll
        for (int i = 0; i < T; ++i) Bn[i] = Cn[i] = dBn[i] = dCn[i] = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float wb = p.wx[(1 + n) * D + d], wc = p.wx[(1 + N + n) * D + d];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                Bn[i] = fmaf(wb, u[d][i], Bn[i]);
                Cn[i] = fmaf(wc, u[d][i], Cn[i]);
            }
        }
        float dAv[D];
#pragma unroll
        for (int dq = 0; dq < D; dq += 2) {
            const float Ar0 = p.A[dq * N + n], Ar1 = p.A[(dq + 1) * N + n];
            const float A0 = Ar0 * MMU_LOG2E, A1 = Ar1 * MMU_LOG2E;
            const int s = n * D + dq;
            float hp0 = hs[(long)s * G], hp1 = hs[(long)(s + 1) * G];   // state entering this lane's first token
            float a0[T], a1[T], hm0[T], hm1[T], c0[T], c1[T];
            // forward over the lane's tokens: h_i, a_i h_{i-1}; gathers what needs h
#pragma unroll
            for (int i = 0; i < T; ++i) {
                a0[i] = fast_exp2(dl[dq][i] * A0);
                a1[i] = fast_exp2(dl[dq + 1][i] * A1);
                hm0[i] = a0[i] * hp0;
                hm1[i] = a1[i] * hp1;
                hp0 = fmaf(dlu[dq][i], Bn[i], hm0[i]);
                hp1 = fmaf(dlu[dq + 1][i], Bn[i], hm1[i]);
                yacc[dq][i] = fmaf(Cn[i], hp0, yacc[dq][i]);
                yacc[dq + 1][i] = fmaf(Cn[i], hp1, yacc[dq + 1][i]);
                dCn[i] = fmaf(dyv[dq][i], hp0, fmaf(dyv[dq + 1][i], hp1, dCn[i]));
                c0[i] = Cn[i] * dyv[dq][i];
                c1[i] = Cn[i] * dyv[dq + 1][i];
            }
            // adjoint, gh_t = a_t g_t with g_t = c_t + gh_{t+1}: the lane's tokens, last first, as one affine map
            float Q0 = 1.f, R0 = 0.f, Q1 = 1.f, R1 = 0.f;
#pragma unroll
            for (int i = T - 1; i >= 0; --i) {
                R0 = a0[i] * (c0[i] + R0);
                R1 = a1[i] * (c1[i] + R1);
                Q0 *= a0[i];
                Q1 *= a1[i];
            }
            wave_scan_affine_x2(Q0, R0, Q1, R1);
            const float Qt0 = readlane63(Q0), Rt0 = readlane63(R0), Qt1 = readlane63(Q1), Rt1 = readlane63(R1);
            const float Qe0 = wave_shift_up1(Q0, 1.f), Re0 = wave_shift_up1(R0, 0.f);
            const float Qe1 = wave_shift_up1(Q1, 1.f), Re1 = wave_shift_up1(R1, 0.f);
            float in0, in1;
            carry_pair(hcar, prog, w, p.nw, DN, s, Qt0, Rt0, Qt1, Rt1, in0, in1);
            float gh0 = fmaf(Qe0, in0, Re0), gh1 = fmaf(Qe1, in1, Re1);   // gh of the token after this lane's last
            float da0 = 0.f, da1 = 0.f;
#pragma unroll
            for (int i = T - 1; i >= 0; --i) {
                const float g0 = c0[i] + gh0, g1 = c1[i] + gh1;
                gh0 = a0[i] * g0;
                gh1 = a1[i] * g1;
                const float t0 = g0 * dl[dq][i], t1 = g1 * dl[dq + 1][i];
                dBn[i] = fmaf(t0, u[dq][i], fmaf(t1, u[dq + 1][i], dBn[i]));
                du[dq][i] = fmaf(t0, Bn[i], du[dq][i]);
                du[dq + 1][i] = fmaf(t1, Bn[i], du[dq + 1][i]);
                da0 = fmaf(t0, hm0[i], da0);
                da1 = fmaf(t1, hm1[i], da1);
                ddl[dq][i] = fmaf(g0, fmaf(u[dq][i], Bn[i], Ar0 * hm0[i]), ddl[dq][i]);
                ddl[dq + 1][i] = fmaf(g1, fmaf(u[dq + 1][i], Bn[i], Ar1 * hm1[i]), ddl[dq + 1][i]);
            }
            dAv[dq] = da0;
            dAv[dq + 1] = da1;
        }
        // d x_dbl rows 1 + n (B_n) and 1 + N + n (C_n): back into u, and their x_proj weight gradients
        float wv[3 * D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float wb = p.wx[(1 + n) * D + d], wc = p.wx[(1 + N + n) * D + d];
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
            wsm[K * D + 1 + d] = fmaf(dyv[d][i], u[d][i], wsm[K * D + 1 + d]);        // dD
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float sq = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                sq = fmaf(p.wout[k * D + d], oz[d], sq);
                wsm[k * D + d] = fmaf(wgt * dyr[k][i], oz[d], wsm[k * D + d]);        // dWout
            }
            wsm[K * D] = fmaf(dyr[k][i], sq, wsm[K * D]);                              // d wgt
        }
        // softplus, dt_proj, the dt row of x_proj
        float ddt = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float raw = fmaf(p.wdt[d], dt[i], p.dtb ? p.dtb[d] : 0.f);
            const float draw = ddl[d][i] * (raw <= 20.f ? sigmoidf_(raw) : 1.f);   // softplus' (threshold 20)
            ddt = fmaf(p.wdt[d], draw, ddt);
            wsm[K * D + 1 + D + d] = fmaf(draw, dt[i], wsm[K * D + 1 + D + d]);        // dWdt
            wsm[K * D + 1 + 2 * D + d] += draw;                                        // d dt_bias
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            du[d][i] = fmaf(p.wx[d], ddt, du[d][i]);
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
    __syncthreads();
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
            for (int m = 0; m < 4; ++m) s = fmaf(p.cw[d * 4 + m], dpl[d * (L + 4) + l + 3 - m], s);
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
                g[k] = fmaf(p.win[j * K + k], dxz[j], g[k]);
                wiv[j * K + k] = fmaf(dxz[j], yo[k], wiv[j * K + k]);
            }
        }
        // the coordinate terms: tap j > c feeds cum[k] for k >= j, tap j < c for k <= j
        {
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
        for (int k = 0; k < K; ++k) {
            p.doff[(((long)b * 2 * K + k) * p.H + h) * p.W + ww] = g[k];
            p.doff[(((long)b * 2 * K + K + k) * p.H + h) * p.W + ww] = 0.f;
        }
    }
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
    __syncthreads();
    for (int i = tid; i < NV; i += blockDim.x) {
        float s = 0.f;
        for (int ww = 0; ww < p.nw; ++ww) s += wpart[ww * NV + i];
        if (i == lay.altho) s *= dwgt_da;
        p.part[(long)b * NV + i] = s;
    }
}

// out[i] = sum over the batch items of part[b][i], fixed order
__global__ __launch_bounds__(256) void mamba_small_reduce_kernel(const float *__restrict__ part, float *__restrict__ out,
                                                                 int B, int NV) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NV) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[(long)b * NV + i];
    out[i] = s;
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

int check(const mmu_mamba_small_params *p, const char *name, int &T, int &nw) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->taps == 1 || p->taps == 3, "%s: 1 or 3 taps supported (got %d)", name, p->taps);
    MMU_CHECK(p->batch > 0 && p->height > 0 && p->width > 0, "%s: empty tensor", name);
    MMU_CHECK(p->dstate >= 1 && p->dstate <= 64, "%s: d_state must be in 1..64 (got %d)", name, p->dstate);
    MMU_CHECK(plan(p->height * p->width, T, nw),
              "%s: height * width must be a multiple of 64, at most 2048, and 64 * {1,2,4} * (<= 8 waves) (got %d)", name,
              p->height * p->width);
    MMU_CHECK(p->offset && p->in_proj_weight && p->conv_weight && p->x_proj_weight && p->dt_proj_weight && p->A &&
                  p->out_proj_weight && p->altho,
              "%s: offset, in_proj / conv / x_proj / dt_proj / out_proj weights, A and altho are required", name);
    return 0;
}

SmallArgs to_args(const mmu_mamba_small_params *p, int nw) {
    SmallArgs a = {};
    a.B = p->batch; a.H = p->height; a.W = p->width; a.L = p->height * p->width; a.N = p->dstate; a.nw = nw;
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
    const size_t lds = sizeof(float) * ((size_t)K * a.L + (size_t)nw * D * a.N + nw);
    hipStream_t st = (hipStream_t)stream;
#define SMALL_FWD(K_, T_, a_)                                                                         \
    if (int r = set_lds_attr(mamba_small_fwd_kernel<K_, T_>, lds, "mamba_small_fwd")) return r;       \
    mamba_small_fwd_kernel<K_, T_><<<a_.B, 64 * nw, lds, st>>>(a_)
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
    const SmallArgs a = to_args(p, nw);
    const int K = p->taps, D = 2 * K;
    const int NV = grad_total(K, a.N);
    const size_t lds = sizeof(float) * ((size_t)K * a.L + (size_t)D * (a.L + 4) + (size_t)nw * D * a.N +
                                        (size_t)nw * NV + nw);
    hipStream_t st = (hipStream_t)stream;
#define SMALL_BWD(K_, T_, a_)                                                                         \
    if (int r = set_lds_attr(mamba_small_bwd_kernel<K_, T_>, lds, "mamba_small_bwd")) return r;       \
    mamba_small_bwd_kernel<K_, T_><<<a_.B, 64 * nw, lds, st>>>(a_)
    SMALL_DISPATCH(SMALL_BWD, K, T, a);
#undef SMALL_BWD
    MMU_HIP_LAUNCH_CHECK("mamba_small_bwd");
    mamba_small_reduce_kernel<<<(NV + 255) / 256, 256, 0, st>>>(p->workspace, p->dweights, a.B, NV);
    MMU_HIP_LAUNCH_CHECK("mamba_small_reduce");
    return 0;
}
