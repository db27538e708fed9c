// selective_scan.hip -- chunk-parallel selective scan for gfx950 (MI355X), fwd + bwd.
//
// Replaces the reference's selective_scan_cuda extension
//   fwd: requirements/Mamba/mamba/csrc/selective_scan/selective_scan_fwd_kernel.cuh:67-345
//   bwd: requirements/Mamba/mamba/csrc/selective_scan/selective_scan_bwd_kernel.cuh:75-531
// with a different decomposition.  The reference runs one thread block per
// (batch, channel) and walks L serially; MM-UNet has 6-channel scans over 65,536
// tokens, so here the sequence is cut into chunks of T = 64*K tokens that are all
// processed in parallel, and the carry between chunks is a tiny second kernel:
//
//   forward
//     K1 chunk_reduce<FWD>  per (b, chunk): for every channel d and state n the affine
//                           map of the chunk  h_out = P*h_in + S   (P = prod a_t, S = local scan)
//     K2 chunk_carry        per (b, d, n): H_c = P_c*H_{c-1} + S_c over chunks (x <- (P, H))
//     K3 chunk_apply_fwd    per (b, chunk): local scan started from H_{c-1}; y, out, out_z
//   backward (same shape, mirrored in time)
//     K1 chunk_reduce<BWD>  reverse aggregates (Q, R) of the adjoint recurrence
//     K2 chunk_carry(reverse)
//     K4 chunk_apply_bwd    recompute h from x, reverse scan for g, all gradients;
//                           dB/dC are reduced over channels inside the workgroup (LDS float
//                           atomics) and stored once -- no global atomics
//     K5 reduce_partials    dA, dD, ddelta_bias partials summed over (batch, chunk)
//
// One workgroup = one (batch, chunk, group); its waves loop over the channels of the
// group, so the B/C tile of the chunk is staged in LDS once and shared by all
// channels (the reference re-reads B/C once per channel block).
// Inside a wave: lane l owns K consecutive tokens; per state the K-token serial
// recurrence runs in registers, lanes are combined with a DPP affine-pair scan.
//
// Math (per b, d, n, t), identical to the reference kernels:
//   dl = softplus(delta + bias) (threshold 20);  a = exp2(dl * A * log2e);  b = dl*u*B
//   h_t = a h_{t-1} + b ;  y = sum_n C h + D u ;  out_z = y * silu(z)
//   adjoint: gamma_t = a_t (C_t dy_t + gamma_{t+1});  g_t = C_t dy_t + gamma_{t+1}
#include <algorithm>
#include "mmu_common.h"
#include "scan_common.h"
#include "../../include/mmunet_amd.h"

namespace {


// Stage one [N][T] tile of B or C (tokens [t0, t0+T) of batch b, group g) into LDS as fp32.
template <typename io_t, int K, bool FULL>
__device__ __forceinline__ void stage_tile(float *__restrict__ s, const io_t *__restrict__ g, long row_stride,
                                           int N, int t0, int L, bool vec) {
    constexpr int T = 64 * K;
    for (int idx = threadIdx.x; idx < N * 64; idx += blockDim.x) {
        const int n = idx >> 6, j = idx & 63;
        const int t = t0 + j * K;
        float v[K];
        load_k<io_t, K, FULL>(g + (long)n * row_stride + t, L - t, vec, v);
        float *dst = s + n * T + j * K;
#pragma unroll
        for (int i = 0; i < K; ++i) dst[i] = v[i];
    }
}

// ---------------------------------------------------------------------------
// K1: per-chunk aggregates.  BWD=false: (P, S) of the state recurrence.
//                            BWD=true : (Q, R) of the adjoint recurrence.
// grid (n_chunks, batch, ngroups), block W*64.  LDS: tile[N][T] | A2[W][N] | res[W][2N]
// ---------------------------------------------------------------------------
template <typename io_t, int K, bool BWD, bool FULL>
__global__ __launch_bounds__(1024) void chunk_reduce_kernel(ScanArgs p) {
    constexpr int T = 64 * K;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = p.dstate, L = p.seqlen;
    const int c = blockIdx.x, b = blockIdx.y, g = blockIdx.z;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, W = blockDim.x >> 6;
    const int t0 = c * T;
    float *sT = smem;
    float *sA = sT + N * T + w * N;
    float *sR = smem + N * T + W * N + w * 2 * N;

    if (!BWD)
        stage_tile<io_t, K, FULL>(sT, (const io_t *)p.B + (long)b * p.B_bs + (long)g * p.B_gs, p.B_ns, N, t0, L, p.vec_bc);
    else
        stage_tile<io_t, K, FULL>(sT, (const io_t *)p.C + (long)b * p.C_bs + (long)g * p.C_gs, p.C_ns, N, t0, L, p.vec_bc);
    __syncthreads();

    const int dpg = p.dim / p.ngroups;
    const int tl = lane * K;            // first local token of this lane
    const int nvalid = L - (t0 + tl);   // may be <= 0
    float *dst_base = BWD ? p.gx : p.x;

    for (int d = g * dpg + w; d < (g + 1) * dpg; d += W) {
        for (int n = lane; n < N; n += 64) sA[n] = p.A[(long)d * p.A_ds + (long)n * p.A_ns] * MMU_LOG2E;
        const float bias = p.delta_bias ? p.delta_bias[d] : 0.f;
        float dl[K], wv[K];  // wv: fwd = dl*u ; bwd = dout*silu(z)
        load_k<io_t, K, FULL>((const io_t *)p.delta + (long)b * p.delta_bs + (long)d * p.delta_ds + t0 + tl, nvalid,
                        p.vec_io, dl);
#pragma unroll
        for (int i = 0; i < K; ++i) {
            float v = dl[i] + bias;
            if (p.softplus) v = softplus_thr(v);
            dl[i] = (FULL || i < nvalid) ? v : 0.f;  // identity element beyond L: a = 1, b = 0
        }
        if (!BWD) {
            load_k<io_t, K, FULL>((const io_t *)p.u + (long)b * p.u_bs + (long)d * p.u_ds + t0 + tl, nvalid, p.vec_io, wv);
#pragma unroll
            for (int i = 0; i < K; ++i) wv[i] *= dl[i];
        } else {
            load_k<io_t, K, FULL>((const io_t *)p.dout + (long)b * p.dout_bs + (long)d * p.dout_ds + t0 + tl, nvalid,
                            p.vec_io, wv);
            if (p.z) {
                float zv[K];
                load_k<io_t, K, FULL>((const io_t *)p.z + (long)b * p.z_bs + (long)d * p.z_ds + t0 + tl, nvalid, p.vec_io,
                                zv);
#pragma unroll
                for (int i = 0; i < K; ++i) wv[i] *= zv[i] * sigmoidf_(zv[i]);
            }
        }
        // cumulative delta over the chunk: fwd needs the exclusive SUFFIX sum, bwd the inclusive PREFIX sum
        float lane_tot = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) lane_tot += dl[i];
        const float incl = wave_scan_add(lane_tot);
        const float tot = wave_bcast_last(incl);
        float cum[K];
        if (!BWD) {
            float run = tot - incl;  // sum over lanes > this one
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                cum[i] = run;
                run += dl[i];
            }
        } else {
            float run = incl - lane_tot;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                run += dl[i];
                cum[i] = run;
            }
        }
        for (int n = 0; n < N; ++n) {
            const float a2 = sA[n];
            const float *row = sT + n * T + tl;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) s = fmaf(fast_exp2(a2 * cum[i]) * wv[i], row[i], s);
            s = wave_scan_add(s);
            if (lane == 63) {
                sR[2 * n] = fast_exp2(a2 * tot);
                sR[2 * n + 1] = s;
            }
        }
        float *dst = dst_base + (((long)b * p.dim + d) * p.n_chunks + c) * 2 * N;
        for (int j = lane; j < 2 * N; j += 64) dst[j] = sR[j];
    }
}

// ---- 8-tokens-per-lane tiles, two states per register pair -----------------------------------------
// A [N][512]-token tile whose consumer lane l owns tokens 8l..8l+7, stored by STATE PAIR so that the
// arithmetic runs on v_pk_{mul,fma}_f32 (2 states per instruction; these kernels are VALU-issue bound,
// measured 71 % VALU-busy): [pair][4][64] float4, quarter q of lane l =
//     (row 2p [8l+2q], row 2p+1 [8l+2q], row 2p [8l+2q+1], row 2p+1 [8l+2q+1]).
// Every ds_read_b128 of a wave is 1 KiB contiguous (conflict-free; the natural [n][512] order put lanes
// 32 B apart = 2-way conflicts, 42 % of the LDS cycles of the first version).  An odd dstate gets a
// zero partner row: B = C = 0, A = 0, h0 = 0 contribute nothing.
template <typename io_t, bool FULL>
__device__ __forceinline__ void stage_pair8(float *__restrict__ s, const io_t *__restrict__ g, long row_stride,
                                            int N, int t0, int L, bool vec) {
    const int NP = (N + 1) >> 1;
    for (int idx = threadIdx.x; idx < NP * 64; idx += blockDim.x) {
        const int pr = idx >> 6, j = idx & 63;
        const int t = t0 + j * 8;
        float v0[8], v1[8];
        load_k<io_t, 8, FULL>(g + (long)(2 * pr) * row_stride + t, L - t, vec, v0);
        if (2 * pr + 1 < N) {
            load_k<io_t, 8, FULL>(g + (long)(2 * pr + 1) * row_stride + t, L - t, vec, v1);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v1[i] = 0.f;
        }
        float4 *dst = reinterpret_cast<float4 *>(s) + pr * 256 + j;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q * 64] = make_float4(v0[2 * q], v1[2 * q], v0[2 * q + 1], v1[2 * q + 1]);
    }
}


// ---------------------------------------------------------------------------
// K1 (fast form, 128-token chunks): per-chunk aggregates with 8 tokens per lane, so one wave covers
// FOUR chunks (one per 16-lane DPP row) and every cross-lane step is a 4-step row operation.
// grid (ceil(n_chunks / 4), batch, ngroups), block W*64.  LDS: tile[N][512] | A2[W][N] | res[W][4][2N]
// ---------------------------------------------------------------------------
template <typename io_t, bool BWD, bool HAS_Z>
__global__ __launch_bounds__(512, 4) void chunk_reduce8_kernel(ScanArgs p) {
    constexpr unsigned ES = sizeof(io_t);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = p.dstate, L = p.seqlen;
    const int NP = (N + 1) >> 1, NE = 2 * NP;
    const int b = blockIdx.y, g = blockIdx.z;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, W = blockDim.x >> 6;
    const int t0 = blockIdx.x * 512;
    const int c0 = blockIdx.x * 4;
    float *sT = smem;
    float *sA = sT + NE * 512 + w * 2 * NE;  // [2][NE]: the next channel's A row is fetched one channel ahead
    float *sR = smem + NE * 512 + W * 2 * NE + w * 8 * NE;

    if (!BWD)
        stage_pair8<io_t, true>(sT, (const io_t *)p.B + (long)b * p.B_bs + (long)g * p.B_gs, p.B_ns, N, t0, L, true);
    else
        stage_pair8<io_t, true>(sT, (const io_t *)p.C + (long)b * p.C_bs + (long)g * p.C_gs, p.C_ns, N, t0, L, true);

    // channel loop: same branch-free software pipeline as chunk_apply_fwd8_kernel (see there)
    const int dpg = p.dim / p.ngroups;
    const int dend = (g + 1) * dpg;
    const int q = lane >> 4;  // which of the wave's 4 chunks
    const rsrc_t r_delta = make_rsrc((const io_t *)p.delta + (long)b * p.delta_bs + t0);
    const rsrc_t r_w = make_rsrc(BWD ? (const io_t *)p.dout + (long)b * p.dout_bs + t0 : (const io_t *)p.u + (long)b * p.u_bs + t0);
    const rsrc_t r_z = make_rsrc(HAS_Z ? (const io_t *)p.z + (long)b * p.z_bs + t0 : (const io_t *)p.u);
    const rsrc_t r_A = make_rsrc(p.A);
    const rsrc_t r_bias = make_rsrc(p.delta_bias ? p.delta_bias : p.A);
    const rsrc_t r_dst = make_rsrc((BWD ? p.gx : p.x) + ((long)b * p.dim * p.n_chunks + c0) * 2 * N);
    const float has_bias = p.delta_bias ? 1.f : 0.f;
    const unsigned w_ds = BWD ? p.dout_ds : p.u_ds;
    const unsigned voff = lane * 8 * ES;
    const int ln = lane < N ? lane : N - 1;
    const unsigned voff_A = ln * (unsigned)p.A_ns * 4u;
    if (lane < 2 * NE) sA[lane] = 0.f;  // the zero state that pads an odd dstate

    float dl_n[8], wv_n[8], a_n, bias_n;
    auto fetch = [&](int dc) {
        dc = __builtin_amdgcn_readfirstlane(dc);
        buf_load8<io_t>(r_delta, voff, dc * (unsigned)p.delta_ds * ES, dl_n);
        buf_load8<io_t>(r_w, voff, dc * w_ds * ES, wv_n);
        a_n = buf_load1(r_A, voff_A, dc * (unsigned)p.A_ds * 4u);
        bias_n = buf_load1(r_bias, 0, dc * 4u) * has_bias;
    };
    int d = g * dpg + w;
    fetch(d);
    asm volatile("" : "+v"(bias_n));  // consume the whole group before the loop (see chunk_apply_fwd8_kernel)
    if (lane < N) sA[lane] = a_n * MMU_LOG2E;
    __syncthreads();
    int buf = 0;
    for (; d < dend; d += W, buf ^= 1) {
        const float bias = bias_n;
        float dl[8], wv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dl[i] = dl_n[i] + bias;
            wv[i] = wv_n[i];
        }
        fetch(d + W < dend ? d + W : d);
        const unsigned dcur = __builtin_amdgcn_readfirstlane(d);
        float zv[8];
        if constexpr (BWD && HAS_Z) buf_load8<io_t>(r_z, voff, dcur * (unsigned)p.z_ds * ES, zv);
        if (p.softplus) {
#pragma unroll
            for (int i = 0; i < 8; ++i) dl[i] = softplus_thr(dl[i]);
        }
        float lane_tot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) lane_tot += dl[i];
        float cum[8], tot;
        if (!BWD) {  // exclusive suffix sums inside the chunk; chunk total lands on the row's lane 0
            const float incl = row_scan_add_down(lane_tot);
            tot = incl;
            float run = incl - lane_tot;
#pragma unroll
            for (int i = 7; i >= 0; --i) {
                cum[i] = run;
                run += dl[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[i] *= dl[i];
        } else {     // inclusive prefix sums; chunk total lands on the row's lane 15
            const float incl = row_scan_add_up(lane_tot);
            tot = incl;
            float run = incl - lane_tot;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                run += dl[i];
                cum[i] = run;
            }
            if constexpr (HAS_Z) {
#pragma unroll
                for (int i = 0; i < 8; ++i) wv[i] *= zv[i] * sigmoidf_(zv[i]);
            }
        }
        v2f cum2[4], wv2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cum2[k] = v2f{cum[2 * k], cum[2 * k + 1]};
            wv2[k] = v2f{wv[2 * k], wv[2 * k + 1]};
        }
        const bool writer = (lane & 15) == (BWD ? 15 : 0);
        const float *cA = sA + buf * NE;
        v2f a2n = *reinterpret_cast<const v2f *>(cA);
        for (int pr = 0; pr < NP; ++pr) {
            const v2f a2 = a2n;
            v2f row[8], wr[8], e[8];
            pair8_read(sT, pr, lane, row);
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // the tile read sits behind the 16 exps
                e[2 * k] = exp2_2(mul_bcast<0>(cum2[k], a2));
                e[2 * k + 1] = exp2_2(mul_bcast<1>(cum2[k], a2));
            }
            a2n = *reinterpret_cast<const v2f *>(cA + 2 * (pr + 1 < NP ? pr + 1 : pr));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                wr[2 * k] = mul_bcast<0>(wv2[k], row[2 * k]);
                wr[2 * k + 1] = mul_bcast<1>(wv2[k], row[2 * k + 1]);
            }
            v2f s = v2f{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) s = fma2(e[i], wr[i], s);
            const float s0 = BWD ? row_scan_add_up(s.x) : row_scan_add_down(s.x);
            const float s1 = BWD ? row_scan_add_up(s.y) : row_scan_add_down(s.y);
            if (writer) {
                float *r = sR + q * 2 * N + 4 * pr;
                r[1] = s0;
                if (2 * pr + 1 < N) r[3] = s1;
            }
        }
        // the chunks' decay products exp2(A2[n] * sum of the chunk's deltas): ONE exp per lane for the whole channel -- lane
        // l takes state l & 15 (l & 15 + 16 k for larger N) of its own row's chunk -- instead of two per pair and lane (every
        // lane of a row used to compute the same sixteen values, 11 % of the kernel's transcendentals)
        {
            // (the chunk total stands on the row's last lane (backward) / first lane (forward): row_newbcast hands it to the row)
            const float tot_row = BWD ? dpp_mov<0x150 + 15, 0xf>(0.f, tot) : dpp_mov<0x150, 0xf>(0.f, tot);
            for (int n = lane & 15; n < N; n += 16) sR[q * 2 * N + 2 * n] = fast_exp2(cA[n] * tot_row);
        }
        // drain the prefetch group BEFORE the record stores: the loop header would otherwise have to wait for
        // its youngest load (bias) with vmcnt(0), i.e. for this channel's stores to be acknowledged
        asm volatile("" : "+v"(bias_n), "+v"(a_n));
        if (lane < N) sA[(buf ^ 1) * NE + lane] = a_n * MMU_LOG2E;
        // the 4 chunks' (P, S) records are contiguous in memory
        const unsigned so = dcur * (unsigned)(p.n_chunks * 2 * N) * 4u;
        for (int j = lane; j < 8 * N; j += 64)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sR[j]), r_dst, j * 4u, so, 0);
    }
}

// ---------------------------------------------------------------------------
// K2: carry over chunks, in place on buf[bd][c][n] = (P, S) -> (P, H).
// reverse=0: H_c = P_c H_{c-1} + S_c ; reverse=1: H_c = P_c H_{c+1} + S_c.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chunk_carry_kernel(float *__restrict__ buf, long total /*B*D*N*/, int n_chunks,
                                                          int N, int reverse) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long bd = idx / N;
    const int n = (int)(idx % N);
    float2 *p = reinterpret_cast<float2 *>(buf) + bd * n_chunks * N + n;
    float h = 0.f;
    constexpr int U = 32;  // latency-bound: 32 independent 8-B loads in flight per thread
    for (int c0 = 0; c0 < n_chunks; c0 += U) {
        float2 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int c = c0 + j;
            const int cc = reverse ? n_chunks - 1 - c : c;
            v[j] = (c < n_chunks) ? p[(long)cc * N] : make_float2(1.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            h = fmaf(v[j].x, h, v[j].y);
            v[j].y = h;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int c = c0 + j;
            const int cc = reverse ? n_chunks - 1 - c : c;
            if (c < n_chunks) p[(long)cc * N] = v[j];
        }
    }
}

// K2, parallel form for power-of-two dstate <= 64: one 256-thread workgroup per (batch, channel); thread =
// (segment of 32 consecutive chunks, state).  Each thread composes its segment in registers, the 256/N
// segment maps meet in LDS, every thread picks up the state entering its segment and rewrites its 32
// records -- one read and one write of the records (134 MB at the headline shape) with 4 waves per SIMD,
// where the serial form has B*D*N threads = one wave per CU walking 512 dependent steps.
template <int N>
__global__ __launch_bounds__(256) void chunk_carry_par_kernel(float *__restrict__ buf, int n_chunks, int reverse) {
    constexpr int SEG = 256 / N, CPT = 32;
    __shared__ float2 agg[SEG][N];
    const int n = threadIdx.x % N, seg = threadIdx.x / N;
    float2 *p = reinterpret_cast<float2 *>(buf) + (long)blockIdx.x * n_chunks * N + n;
    float Hblk = 0.f;  // state entering the current super-block of SEG * CPT chunks
    for (int base = 0; base < n_chunks; base += SEG * CPT) {
        const int cs = base + seg * CPT;
        float2 v[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = cs + j;
            const int cc = reverse ? n_chunks - 1 - c : c;
            v[j] = c < n_chunks ? p[(long)cc * N] : make_float2(1.f, 0.f);
        }
        float P = 1.f, S = 0.f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            S = fmaf(v[j].x, S, v[j].y);
            P *= v[j].x;
        }
        agg[seg][n] = make_float2(P, S);
        __syncthreads();
        float H = Hblk, Hnext = 0.f;
        for (int sgm = 0; sgm < SEG; ++sgm) {
            if (sgm == seg) Hnext = H;  // (H entering this thread's segment)
            const float2 a = agg[sgm][n];
            H = fmaf(a.x, H, a.y);
        }
        Hblk = H;  // state leaving the super-block
        H = Hnext;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = cs + j;
            const int cc = reverse ? n_chunks - 1 - c : c;
            H = fmaf(v[j].x, H, v[j].y);
            if (c < n_chunks) p[(long)cc * N] = make_float2(v[j].x, H);
        }
        __syncthreads();
    }
}

int launch_carry(float *buf, int batch, int dim, int N, int n_chunks, int reverse, hipStream_t st) {
    const long rows = (long)batch * dim;
    if (n_chunks >= 64 && rows < (1L << 31) && (N == 4 || N == 8 || N == 16 || N == 32 || N == 64)) {
        switch (N) {
            case 4: chunk_carry_par_kernel<4><<<(unsigned)rows, 256, 0, st>>>(buf, n_chunks, reverse); break;
            case 8: chunk_carry_par_kernel<8><<<(unsigned)rows, 256, 0, st>>>(buf, n_chunks, reverse); break;
            case 16: chunk_carry_par_kernel<16><<<(unsigned)rows, 256, 0, st>>>(buf, n_chunks, reverse); break;
            case 32: chunk_carry_par_kernel<32><<<(unsigned)rows, 256, 0, st>>>(buf, n_chunks, reverse); break;
            default: chunk_carry_par_kernel<64><<<(unsigned)rows, 256, 0, st>>>(buf, n_chunks, reverse); break;
        }
    } else {
        const long total = rows * N;
        chunk_carry_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(buf, total, n_chunks, N, reverse);
    }
    MMU_HIP_LAUNCH_CHECK(reverse ? "chunk_carry(reverse)" : "chunk_carry");
    return 0;
}

// ---------------------------------------------------------------------------
// K3: forward apply.  grid (ceil(L / (64*KX)), batch, ngroups), block W*64.
// KX tokens per lane (tile of 64*KX tokens); chunk carries live at 64*KC-token granularity
// (KX is a multiple of KC, so a tile always starts on a chunk boundary).
// LDS: B[N][TT] | C[N][TT] | A2[W][N] | H0[W][N]
// ---------------------------------------------------------------------------
// One or two states at a time: K-token serial recurrence in registers, lanes joined by the DPP
// affine scan.  The chunk carry h0 is folded into lane 0's element before the scan, so the
// inclusive scan directly yields the state at the end of every lane.
template <int KX, int NS>
__device__ __forceinline__ void fwd_states(const float (&dl)[KX], const float (&du)[KX], float (&y)[KX],
                                           const float *rowB, const float *rowC, int TT, const float *a2,
                                           const float *h0, int lane) {
    // rowB / rowC: start of state n's row in the B / C tile
    float a[NS][KX], bb[NS][KX], P[NS], S[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float Bv[KX];
#pragma unroll
        for (int i = 0; i < KX; ++i) Bv[i] = rowB[s * TT + lane * KX + i];
#pragma unroll
        for (int i = 0; i < KX; ++i) {
            a[s][i] = fast_exp2(dl[i] * a2[s]);
            bb[s][i] = du[i] * Bv[i];
        }
        P[s] = a[s][0];
        S[s] = bb[s][0];
#pragma unroll
        for (int i = 1; i < KX; ++i) {
            S[s] = fmaf(a[s][i], S[s], bb[s][i]);
            P[s] *= a[s][i];
        }
        S[s] = lane == 0 ? fmaf(P[s], h0[s], S[s]) : S[s];
    }
    if constexpr (NS == 2)
        wave_scan_affine_x2(P[0], S[0], P[1], S[1]);
    else
        wave_scan_affine(P[0], S[0]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float Cv[KX];
#pragma unroll
        for (int i = 0; i < KX; ++i) Cv[i] = rowC[s * TT + lane * KX + i];
        float h = wave_shift_up1(S[s], h0[s]);  // state entering this lane's tokens
#pragma unroll
        for (int i = 0; i < KX; ++i) {
            h = fmaf(a[s][i], h, bb[s][i]);
            y[i] = fmaf(Cv[i], h, y[i]);
        }
    }
}

// The same for the 8-tokens-per-lane pair tiles: both states of a pair advance in one packed instruction.
// The lane's decay product is exp2(A * sum of its deltas) (one exp instead of 7 multiplies per state).
// LDS latency is kept off the critical path without spending registers on it: a2 / h0 arrive holding THIS
// pair's values and leave holding the next pair's (read a whole scan ahead of use); the B tile read is
// issued first and lands behind the 16 exps (which do not need it), the C tile read is issued before the
// cross-lane scan and lands behind it (so B and C values share registers).
__device__ __forceinline__ void fwd_pair8(const v2f (&dl)[4], const v2f (&du)[4], float dlsum, v2f (&yp)[8],
                                          v2f &a2, v2f &h0, const float *tileB, const float *tileC,
                                          const float *cA, const float *cH, int pr, int npairs, int lane) {
    // dl / du: this lane's 8 tokens, two per register pair
    v2f a[8], bb[8], Bv[8], Cv[8];
    pair8_read(tileB, pr, lane, Bv);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a[2 * q] = exp2_2(mul_bcast<0>(dl[q], a2));
        a[2 * q + 1] = exp2_2(mul_bcast<1>(dl[q], a2));
    }
    const v2f P = exp2_2(a2 * dlsum);
    const v2f h_in = h0;
    const int nx = pr + 1 < npairs ? pr + 1 : pr;  // the last pair re-reads itself (branch-free)
    a2 = *reinterpret_cast<const v2f *>(cA + 2 * nx);
    h0 = *reinterpret_cast<const v2f *>(cH + 2 * nx);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bb[2 * q] = mul_bcast<0>(du[q], Bv[2 * q]);
        bb[2 * q + 1] = mul_bcast<1>(du[q], Bv[2 * q + 1]);
    }
    v2f S = bb[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) S = fma2(a[i], S, bb[i]);
    const v2f S_in = fma2(P, h_in, S);
    S = lane == 0 ? S_in : S;
    float P0 = P.x, S0 = S.x, P1 = P.y, S1 = S.y;
    __builtin_amdgcn_sched_barrier(0);
    pair8_read(tileC, pr, lane, Cv);  // lands during the scan, in the registers the B values just left
    wave_scan_affine_x2(P0, S0, P1, S1);
    v2f h = v2f{wave_shift_up1(S0, h_in.x), wave_shift_up1(S1, h_in.y)};  // states entering this lane's tokens
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h = fma2(a[i], h, bb[i]);
        yp[i] = fma2(Cv[i], h, yp[i]);
    }
}

template <typename io_t, int KX, bool FULL>
__global__ __launch_bounds__(1024) void chunk_apply_fwd_kernel(ScanArgs p) {
    constexpr int TT = 64 * KX;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = p.dstate, L = p.seqlen;
    const int b = blockIdx.y, g = blockIdx.z;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, W = blockDim.x >> 6;
    const int t0 = blockIdx.x * TT;
    const int cprev = blockIdx.x - 1;  // chunk whose end state enters this tile
    float *sB = smem;
    float *sC = sB + N * TT;
    float *sA = sC + N * TT + w * N;
    float *sH = smem + 2 * N * TT + W * N + w * N;

    stage_tile<io_t, KX, FULL>(sB, (const io_t *)p.B + (long)b * p.B_bs + (long)g * p.B_gs, p.B_ns, N, t0, L, p.vec_bc);
    stage_tile<io_t, KX, FULL>(sC, (const io_t *)p.C + (long)b * p.C_bs + (long)g * p.C_gs, p.C_ns, N, t0, L, p.vec_bc);
    __syncthreads();

    const int dpg = p.dim / p.ngroups;
    const int tl = lane * KX;
    const int nvalid = L - (t0 + tl);
    const int dend = (g + 1) * dpg;
    const io_t *pdelta = (const io_t *)p.delta + (long)b * p.delta_bs + t0 + tl;
    const io_t *pu = (const io_t *)p.u + (long)b * p.u_bs + t0 + tl;

    // software pipeline over channels: the next channel's u / delta are in flight during this one's scan
    float dl_n[KX], du_n[KX];
    int d = g * dpg + w;
    if (d < dend) {
        load_k<io_t, KX, FULL>(pdelta + (long)d * p.delta_ds, nvalid, p.vec_io, dl_n);
        load_k<io_t, KX, FULL>(pu + (long)d * p.u_ds, nvalid, p.vec_io, du_n);
    }
    for (; d < dend; d += W) {
        const float *xprev = p.x + (((long)b * p.dim + d) * p.n_chunks + cprev) * 2 * N;
        for (int n = lane; n < N; n += 64) {
            sA[n] = p.A[(long)d * p.A_ds + (long)n * p.A_ns] * MMU_LOG2E;
            sH[n] = (cprev >= 0) ? xprev[2 * n + 1] : 0.f;
        }
        const float bias = p.delta_bias ? p.delta_bias[d] : 0.f;
        const float Dv = p.D ? p.D[d] : 0.f;
        float dl[KX], du[KX], y[KX], zv[KX];
#pragma unroll
        for (int i = 0; i < KX; ++i) {
            dl[i] = dl_n[i];
            du[i] = du_n[i];
        }
        if (d + W < dend) {
            load_k<io_t, KX, FULL>(pdelta + (long)(d + W) * p.delta_ds, nvalid, p.vec_io, dl_n);
            load_k<io_t, KX, FULL>(pu + (long)(d + W) * p.u_ds, nvalid, p.vec_io, du_n);
        }
        if (p.z)  // consumed after the state loop
            load_k<io_t, KX, FULL>((const io_t *)p.z + (long)b * p.z_bs + (long)d * p.z_ds + t0 + tl, nvalid, p.vec_io,
                                   zv);
#pragma unroll
        for (int i = 0; i < KX; ++i) {
            float v = dl[i] + bias;
            if (p.softplus) v = softplus_thr(v);
            dl[i] = (FULL || i < nvalid) ? v : 0.f;
            y[i] = Dv * du[i];
            du[i] *= dl[i];
        }
        int n = 0;
        for (; n + 1 < N; n += 2)
            fwd_states<KX, 2>(dl, du, y, sB + n * TT, sC + n * TT, TT, sA + n, sH + n, lane);
        if (n < N) fwd_states<KX, 1>(dl, du, y, sB + n * TT, sC + n * TT, TT, sA + n, sH + n, lane);
        if (p.out)
            store_k<io_t, KX, FULL>((io_t *)p.out + (long)b * p.out_bs + (long)d * p.out_ds + t0 + tl, nvalid,
                                    p.vec_io, y);
        if (p.z) {
#pragma unroll
            for (int i = 0; i < KX; ++i) y[i] *= zv[i] * sigmoidf_(zv[i]);
            store_k<io_t, KX, FULL>((io_t *)p.out_z + (long)b * p.out_z_bs + (long)d * p.out_z_ds + t0 + tl, nvalid,
                                    p.vec_io, y);
        }
    }
}

// K3 for dstate <= 16 on full, aligned tiles: 512-token tiles (8 tokens per lane) over 128-token chunk
// carries, state pairs on packed math.  grid (L / 512, batch, ngroups), block W*64 with W <= 8 so that two
// workgroups share a CU (one stages its tile while the other scans).
// LDS: B[NE][512] | C[NE][512] | A2[W][2][NE] | H0[W][2][NE].
//
// The channel loop is a software pipeline with NO control flow around its memory operations and no
// per-stream address registers: s_waitcnt vmcnt counts in issue order, and at a control-flow merge (or a
// scratch reload) the compiler emits vmcnt(0) -- with `if (p.z)` / `if (more)` around the loads and spilled
// 64-bit row pointers every channel drained the whole queue (its own z, the next channel's prefetch, the
// previous channel's stores) before its first softplus.  So: rows are addressed as buffer resource + scalar
// row offset + one shared lane offset, z is a template flag, the prefetch always runs (the last iteration
// re-reads its own rows, L2 hits), A / h0 / bias / D are fetched with the prefetch by every lane.
template <typename io_t, bool HAS_Z>
__global__ __launch_bounds__(512, 4) void chunk_apply_fwd8_kernel(ScanArgs p) {
    constexpr int TT = 512;
    constexpr unsigned ES = sizeof(io_t);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = p.dstate, L = p.seqlen;
    const int b = blockIdx.y, g = blockIdx.z;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, W = blockDim.x >> 6;
    const int t0 = blockIdx.x * TT;
    const int cprev = blockIdx.x * 4 - 1;  // 128-token chunk whose end state enters this tile
    const int NE = (N + 1) & ~1;           // rows held (an odd dstate is padded with a zero state)
    float *sB = smem;
    float *sC = sB + NE * TT;
    float *sA = sC + NE * TT + w * 2 * NE;
    float *sH = smem + 2 * NE * TT + W * 2 * NE + w * 2 * NE;

    stage_pair8<io_t, true>(sB, (const io_t *)p.B + (long)b * p.B_bs + (long)g * p.B_gs, p.B_ns, N, t0, L, true);
    stage_pair8<io_t, true>(sC, (const io_t *)p.C + (long)b * p.C_bs + (long)g * p.C_gs, p.C_ns, N, t0, L, true);

    const int dpg = p.dim / p.ngroups;
    const int dend = (g + 1) * dpg;
    const rsrc_t r_delta = make_rsrc((const io_t *)p.delta + (long)b * p.delta_bs + t0);
    const rsrc_t r_u = make_rsrc((const io_t *)p.u + (long)b * p.u_bs + t0);
    const rsrc_t r_z = make_rsrc(HAS_Z ? (const io_t *)p.z + (long)b * p.z_bs + t0 : (const io_t *)p.u);
    const rsrc_t r_oz = make_rsrc(HAS_Z ? (io_t *)p.out_z + (long)b * p.out_z_bs + t0 : (io_t *)p.out);
    const rsrc_t r_out = make_rsrc(p.out ? (io_t *)p.out + (long)b * p.out_bs + t0 : nullptr);
    const rsrc_t r_A = make_rsrc(p.A);
    const rsrc_t r_x = make_rsrc(p.x + (long)b * p.dim * p.n_chunks * 2 * N);
    const rsrc_t r_bias = make_rsrc(p.delta_bias ? p.delta_bias : p.A);  // a missing vector reads as 0 (flag)
    const rsrc_t r_D = make_rsrc(p.D ? p.D : p.A);
    const float has_bias = p.delta_bias ? 1.f : 0.f, has_D = p.D ? 1.f : 0.f;
    const unsigned voff = lane * 8 * ES;
    const int ln = lane < N ? lane : N - 1;  // clamped state index: every lane loads, only lanes < N keep
    const unsigned voff_A = ln * (unsigned)p.A_ns * 4u;
    const unsigned voff_x = ((cprev >= 0 ? cprev : 0) * 2 * N + 2 * ln + 1) * 4u;
    if (lane < 2 * NE) {  // the zero state that pads an odd dstate, and the zero entering state of the first tile
        sA[lane] = 0.f;
        sH[lane] = 0.f;
    }

    // everything one channel needs from global memory, as ONE group of loads issued a channel ahead
    float dl_n[8], du_n[8], a_n, h_n, bias_n, D_n;
    auto fetch = [&](int dc) {
        dc = __builtin_amdgcn_readfirstlane(dc);
        buf_load8<io_t>(r_delta, voff, dc * (unsigned)p.delta_ds * ES, dl_n);
        buf_load8<io_t>(r_u, voff, dc * (unsigned)p.u_ds * ES, du_n);
        a_n = buf_load1(r_A, voff_A, dc * (unsigned)p.A_ds * 4u);
        h_n = buf_load1(r_x, voff_x, dc * (unsigned)(p.n_chunks * 2 * N) * 4u);
        bias_n = buf_load1(r_bias, 0, dc * 4u) * has_bias;
        D_n = buf_load1(r_D, 0, dc * 4u) * has_D;
    };
    int d = g * dpg + w;  // W <= channels per group: every wave owns at least one channel
    fetch(d);
    // consume the whole group here: otherwise the loop header inherits "bias was the second-youngest load"
    // from this path and waits, every iteration, for all but one of the previous channel's STORES
    asm volatile("" : "+v"(bias_n), "+v"(D_n));
    if (lane < N) {
        sA[lane] = a_n * MMU_LOG2E;
        if (cprev >= 0) sH[lane] = h_n;
    }
    __syncthreads();
    int buf = 0;
    for (; d < dend; d += W, buf ^= 1) {
        const float bias = bias_n, Dv = D_n;
        float dl[8], du[8], zv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dl[i] = dl_n[i] + bias;
            du[i] = du_n[i];
        }
        fetch(d + W < dend ? d + W : d);  // next channel (the last iteration re-reads its own rows: L2 hits)
        const unsigned dcur = __builtin_amdgcn_readfirstlane(d);
        if constexpr (HAS_Z) buf_load8<io_t>(r_z, voff, dcur * (unsigned)p.z_ds * ES, zv);  // used after the state loop
        if (p.softplus) {
#pragma unroll
            for (int i = 0; i < 8; ++i) dl[i] = softplus_thr(dl[i]);
        }
        float dlsum = 0.f;
        v2f yp[8], dl2[4], du2[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dlsum += dl[i];
            yp[i] = v2f{Dv * du[i], 0.f};
            du[i] *= dl[i];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            dl2[q] = v2f{dl[2 * q], dl[2 * q + 1]};
            du2[q] = v2f{du[2 * q], du[2 * q + 1]};
        }
        const float *cA = sA + buf * NE, *cH = sH + buf * NE;
        v2f a2 = *reinterpret_cast<const v2f *>(cA), h0 = *reinterpret_cast<const v2f *>(cH);
        for (int pr = 0; pr < NE / 2; ++pr)
            fwd_pair8(dl2, du2, dlsum, yp, a2, h0, sB, sC, cA, cH, pr, NE / 2, lane);
        if (lane < N) {
            sA[(buf ^ 1) * NE + lane] = a_n * MMU_LOG2E;
            if (cprev >= 0) sH[(buf ^ 1) * NE + lane] = h_n;
        }
        float y[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = yp[i].x + yp[i].y;
        if (p.out) buf_store8<io_t>(r_out, voff, dcur * (unsigned)p.out_ds * ES, y);
        if constexpr (HAS_Z) {
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] *= zv[i] * sigmoidf_(zv[i]);
            buf_store8<io_t>(r_oz, voff, dcur * (unsigned)p.out_z_ds * ES, y);
        }
    }
}

// Backward work of one or two states for this lane's K tokens (see the math block at the top).
// dB/dC contributions of this channel go to accB/accC: REG=true -> the caller's register accumulators
// [NS][K] (summed over the wave's channel loop, stored once per chunk); REG=false -> LDS float atomics
// into tiles laid out [n][i][lane] (generic-dstate fallback; the LDS atomic unit retires only ~1 lane
// per 3.5 cycles per CU, measured, so that path is slow).  dAtot[s] = wave-uniform sum of the dA terms.
template <int K, int NS, bool REG>
__device__ __forceinline__ void bwd_states(const float (&dl)[K], const float (&uv)[K], const float (&dy)[K],
                                           float (&y)[K], float (&duv)[K], float (&ddl)[K], const float *rb,
                                           const float *rc, float *accB, float *accC, int T, const float *a2,
                                           const float *h0, const float *g0, float (&dAtot)[NS], int lane) {
    float a[NS][K], bb[NS][K], hh[NS][K], cc[NS][K], P[NS], S[NS], Q[NS], R[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            a[s][i] = fast_exp2(dl[i] * a2[s]);
            bb[s][i] = dl[i] * uv[i] * rb[s * T + i];
            cc[s][i] = rc[s * T + i] * dy[i];
        }
        P[s] = a[s][0];
        S[s] = bb[s][0];
#pragma unroll
        for (int i = 1; i < K; ++i) {
            S[s] = fmaf(a[s][i], S[s], bb[s][i]);
            P[s] *= a[s][i];
        }
        // adjoint map of this lane, composed right-to-left: gamma_out = a_i (c_i + gamma_in)
        Q[s] = P[s];
        R[s] = 0.f;
#pragma unroll
        for (int i = K - 1; i >= 0; --i) R[s] = a[s][i] * (cc[s][i] + R[s]);
        S[s] = lane == 0 ? fmaf(P[s], h0[s], S[s]) : S[s];
        R[s] = lane == 63 ? fmaf(Q[s], g0[s], R[s]) : R[s];
        Q[s] = wave_reverse(Q[s]);
        R[s] = wave_reverse(R[s]);
    }
    if constexpr (NS == 2) {
        wave_scan_affine_x2(P[0], S[0], P[1], S[1]);
        wave_scan_affine_x2(Q[0], R[0], Q[1], R[1]);
    } else {
        wave_scan_affine(P[0], S[0]);
        wave_scan_affine(Q[0], R[0]);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float An = a2[s] * MMU_LN2;
        float h = wave_shift_up1(S[s], h0[s]);
#pragma unroll
        for (int i = 0; i < K; ++i) {
            h = fmaf(a[s][i], h, bb[s][i]);
            hh[s][i] = h;
        }
        // gamma entering this lane from the right = reversed-order inclusive R of lane+1
        float gam = __builtin_bit_cast(
            float, __builtin_amdgcn_ds_bpermute((62 - lane) << 2, __builtin_bit_cast(int, R[s])));
        gam = lane == 63 ? g0[s] : gam;
        float dAp = 0.f;
#pragma unroll
        for (int i = K - 1; i >= 0; --i) {
            const float gt = cc[s][i] + gam;
            gam = a[s][i] * gt;
            const float ahp = hh[s][i] - bb[s][i];  // a_t * h_{t-1}
            const float gdl = gt * dl[i];
            const float Bv = rb[s * T + i];
            duv[i] = fmaf(gdl, Bv, duv[i]);
            ddl[i] += gt * fmaf(uv[i], Bv, An * ahp);
            dAp = fmaf(gdl, ahp, dAp);
            y[i] = fmaf(rc[s * T + i], hh[s][i], y[i]);
            if constexpr (REG) {
                accB[s * K + i] = fmaf(gdl, uv[i], accB[s * K + i]);
                accC[s * K + i] = fmaf(dy[i], hh[s][i], accC[s * K + i]);
            } else {
                atomicAdd(accB + s * T + i * 64, gdl * uv[i]);
                atomicAdd(accC + s * T + i * 64, dy[i] * hh[s][i]);
            }
        }
        dAtot[s] = wave_sum(dAp);
    }
}

// ---------------------------------------------------------------------------
// K4: backward apply.  grid (n_chunks, batch, ngroups), block W*64 (W <= 8).
// LDS: B[N][T] | C[N][T] | dB[N][T] | dC[N][T] | A2[W][N] | H0[W][N] | G0[W][N] | dAp[W][N]
// ---------------------------------------------------------------------------
template <typename io_t, int K, bool FULL>
__global__ __launch_bounds__(512) void chunk_apply_bwd_kernel(ScanArgs p) {
    constexpr int T = 64 * K;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = p.dstate, L = p.seqlen;
    const int c = blockIdx.x, b = blockIdx.y, g = blockIdx.z;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, W = blockDim.x >> 6;
    const int t0 = c * T;
    float *sB = smem;
    float *sC = sB + N * T;
    float *sdB = sC + N * T;
    float *sdC = sdB + N * T;
    float *scr = sdC + N * T;
    float *sA = scr + w * N;
    float *sH = scr + W * N + w * N;
    float *sG = scr + 2 * W * N + w * N;
    float *sdA = scr + 3 * W * N + w * N;

    stage_tile<io_t, K, FULL>(sB, (const io_t *)p.B + (long)b * p.B_bs + (long)g * p.B_gs, p.B_ns, N, t0, L, p.vec_bc);
    stage_tile<io_t, K, FULL>(sC, (const io_t *)p.C + (long)b * p.C_bs + (long)g * p.C_gs, p.C_ns, N, t0, L, p.vec_bc);
    for (int i = threadIdx.x; i < 2 * N * T; i += blockDim.x) sdB[i] = 0.f;  // sdB and sdC are adjacent
    __syncthreads();

    const int dpg = p.dim / p.ngroups;
    const int tl = lane * K;
    const int nvalid = L - (t0 + tl);

    for (int d = g * dpg + w; d < (g + 1) * dpg; d += W) {
        const long bdc = ((long)b * p.dim + d) * p.n_chunks;
        for (int n = lane; n < N; n += 64) {
            sA[n] = p.A[(long)d * p.A_ds + (long)n * p.A_ns] * MMU_LOG2E;
            sH[n] = (c > 0) ? p.x[(bdc + c - 1) * 2 * N + 2 * n + 1] : 0.f;
            sG[n] = (c + 1 < p.n_chunks) ? p.gx[(bdc + c + 1) * 2 * N + 2 * n + 1] : 0.f;
        }
        const float bias = p.delta_bias ? p.delta_bias[d] : 0.f;
        const float Dv = p.D ? p.D[d] : 0.f;
        float dl[K], uv[K], dy[K], y[K], dsp[K], duv[K], ddl[K], go[K];
        load_k<io_t, K, FULL>((const io_t *)p.delta + (long)b * p.delta_bs + (long)d * p.delta_ds + t0 + tl, nvalid,
                        p.vec_io, dl);
        load_k<io_t, K, FULL>((const io_t *)p.u + (long)b * p.u_bs + (long)d * p.u_ds + t0 + tl, nvalid, p.vec_io, uv);
        load_k<io_t, K, FULL>((const io_t *)p.dout + (long)b * p.dout_bs + (long)d * p.dout_ds + t0 + tl, nvalid, p.vec_io,
                        go);
        float zv[K], zsig[K];
        if (p.z) {
            load_k<io_t, K, FULL>((const io_t *)p.z + (long)b * p.z_bs + (long)d * p.z_ds + t0 + tl, nvalid, p.vec_io, zv);
#pragma unroll
            for (int i = 0; i < K; ++i) zsig[i] = sigmoidf_(zv[i]);
        }
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const float v = dl[i] + bias;
            float sp = v, dspv = 1.f;
            if (p.softplus) {
                sp = softplus_thr(v);
                dspv = v <= 20.f ? sigmoidf_(v) : 1.f;  // d softplus / dx  (bwd_kernel.cuh:439-453)
            }
            dl[i] = (FULL || i < nvalid) ? sp : 0.f;
            dsp[i] = dspv;
            dy[i] = p.z ? go[i] * zv[i] * zsig[i] : go[i];
            y[i] = Dv * uv[i];
            duv[i] = Dv * dy[i];
            ddl[i] = 0.f;
        }
        float dDp = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) dDp = fmaf(dy[i], uv[i], dDp);

        int n = 0;
        for (; n + 1 < N; n += 2) {
            float t2[2];
            bwd_states<K, 2, false>(dl, uv, dy, y, duv, ddl, sB + n * T + tl, sC + n * T + tl, sdB + n * T + lane,
                                    sdC + n * T + lane, T, sA + n, sH + n, sG + n, t2, lane);
            if (lane == 0) {
                sdA[n] = t2[0];
                sdA[n + 1] = t2[1];
            }
        }
        if (n < N) {
            float t1[1];
            bwd_states<K, 1, false>(dl, uv, dy, y, duv, ddl, sB + n * T + tl, sC + n * T + tl, sdB + n * T + lane,
                                    sdC + n * T + lane, T, sA + n, sH + n, sG + n, t1, lane);
            if (lane == 0) sdA[n] = t1[0];
        }
        // per-channel outputs
        float dbp = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            ddl[i] *= dsp[i];
            if (FULL || i < nvalid) dbp += ddl[i];
        }
        store_k<io_t, K, FULL>((io_t *)p.du + (long)b * p.du_bs + (long)d * p.du_ds + t0 + tl, nvalid, p.vec_io, duv);
        store_k<io_t, K, FULL>((io_t *)p.ddelta + (long)b * p.ddelta_bs + (long)d * p.ddelta_ds + t0 + tl, nvalid, p.vec_io,
                         ddl);
        if (p.z) {
            float dzv[K];
#pragma unroll
            for (int i = 0; i < K; ++i) dzv[i] = go[i] * y[i] * zsig[i] * (1.f + zv[i] * (1.f - zsig[i]));
            store_k<io_t, K, FULL>((io_t *)p.dz + (long)b * p.dz_bs + (long)d * p.dz_ds + t0 + tl, nvalid, p.vec_io, dzv);
            if (p.out_z) {
#pragma unroll
                for (int i = 0; i < K; ++i) dzv[i] = y[i] * zv[i] * zsig[i];
                store_k<io_t, K, FULL>((io_t *)p.out_z + (long)b * p.out_z_bs + (long)d * p.out_z_ds + t0 + tl, nvalid,
                                 p.vec_io, dzv);
            }
        }
        dDp = wave_sum(dDp);
        dbp = wave_sum(dbp);
        float *part = p.part + (((long)b * p.n_chunks + c) * p.dim + d) * (N + 2);
        for (int n = lane; n < N; n += 64) part[n] = sdA[n];
        if (lane == 0) {
            part[N] = dDp;
            part[N + 1] = dbp;
        }
    }
    __syncthreads();
    // dB / dC of this (b, g, chunk): every channel of the group has been added -> plain stores
    float *dBg = p.dB + (long)b * p.dB_bs + (long)g * p.dB_gs;
    float *dCg = p.dC + (long)b * p.dC_bs + (long)g * p.dC_gs;
    for (int idx = threadIdx.x; idx < N * T; idx += blockDim.x) {
        const int n = idx / T, j = idx % T;  // j = local token; accumulators are stored [n][j % K][j / K]
        const int t = t0 + j;
        if (t < L) {
            const int src = n * T + (j % K) * 64 + j / K;
            dBg[(long)n * p.dB_ns + t] = sdB[src];
            dCg[(long)n * p.dC_ns + t] = sdC[src];
        }
    }
}

// ---------------------------------------------------------------------------
// K4s: backward apply for dstate == 16, state-split form.
// grid (n_chunks, batch, ngroups), block 256 = 4 waves = the 4 state-quarters of one (b, chunk, g).
// Every wave walks ALL channels of the group with its 4 states, so its dB/dC sums are 4*K*2 registers
// (a wave holding all 16 states needs 64 accumulators next to ~18 live registers per unrolled state:
// >256 VGPRs, scratch spills, 12 % VALU utilisation measured).  Per channel the four partial sums of
// y / du / d(delta) meet through a double-buffered 3*K*64-float-per-wave LDS exchange (one barrier), and
// the four output streams are split one per wave: du | ddelta (+dbias) | dz | out_z (+dD).  dB/dC leave
// the registers with plain stores at the end -- no atomics anywhere, bit-reproducible.
// LDS: B[16][T] | C[16][T] | xch[2][4][3K][64]
// ---------------------------------------------------------------------------
template <typename io_t, int K, bool FULL>
__global__ __launch_bounds__(256, 3) void chunk_apply_bwd_ns4_kernel(ScanArgs p) {
    constexpr int T = 64 * K, N = 16, NS = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = p.seqlen;
    const int c = blockIdx.x, b = blockIdx.y, g = blockIdx.z;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n0 = w * NS;
    const int t0 = c * T;
    float *sB = smem;
    float *sC = sB + N * T;
    float *xch = sC + N * T;  // [parity][wave][3K][64]

    stage_tile<io_t, K, FULL>(sB, (const io_t *)p.B + (long)b * p.B_bs + (long)g * p.B_gs, p.B_ns, N, t0, L, p.vec_bc);
    stage_tile<io_t, K, FULL>(sC, (const io_t *)p.C + (long)b * p.C_bs + (long)g * p.C_gs, p.C_ns, N, t0, L, p.vec_bc);
    __syncthreads();

    const int dpg = p.dim / p.ngroups;
    const int tl = lane * K;
    const int nvalid = L - (t0 + tl);
    float accB[NS * K], accC[NS * K];
#pragma unroll
    for (int i = 0; i < NS * K; ++i) accB[i] = accC[i] = 0.f;
    int par = 0;

    for (int d = g * dpg; d < (g + 1) * dpg; ++d) {
        const long bdc = ((long)b * p.dim + d) * p.n_chunks;
        // this wave's 4 states: A*log2e, forward carry-in, adjoint carry-in (lanes 0..3, then broadcast)
        float vA = 0.f, vH = 0.f, vG = 0.f;
        if (lane < NS) {
            vA = p.A[(long)d * p.A_ds + (long)(n0 + lane) * p.A_ns] * MMU_LOG2E;
            if (c > 0) vH = p.x[(bdc + c - 1) * 2 * N + 2 * (n0 + lane) + 1];
            if (c + 1 < p.n_chunks) vG = p.gx[(bdc + c + 1) * 2 * N + 2 * (n0 + lane) + 1];
        }
        const float bias = p.delta_bias ? p.delta_bias[d] : 0.f;
        const float Dv = p.D ? p.D[d] : 0.f;
        float dl[K], uv[K], dy[K], y[K], dsp[K], duv[K], ddl[K], go[K], zv[K], zsig[K];
        load_k<io_t, K, FULL>((const io_t *)p.delta + (long)b * p.delta_bs + (long)d * p.delta_ds + t0 + tl, nvalid,
                              p.vec_io, dl);
        load_k<io_t, K, FULL>((const io_t *)p.u + (long)b * p.u_bs + (long)d * p.u_ds + t0 + tl, nvalid, p.vec_io, uv);
        load_k<io_t, K, FULL>((const io_t *)p.dout + (long)b * p.dout_bs + (long)d * p.dout_ds + t0 + tl, nvalid,
                              p.vec_io, go);
        if (p.z) {
            load_k<io_t, K, FULL>((const io_t *)p.z + (long)b * p.z_bs + (long)d * p.z_ds + t0 + tl, nvalid, p.vec_io,
                                  zv);
#pragma unroll
            for (int i = 0; i < K; ++i) zsig[i] = sigmoidf_(zv[i]);
        }
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const float v = dl[i] + bias;
            float sp = v, dspv = 1.f;
            if (p.softplus) {
                sp = softplus_thr(v);
                dspv = v <= 20.f ? sigmoidf_(v) : 1.f;  // d softplus / dx  (bwd_kernel.cuh:439-453)
            }
            dl[i] = (FULL || i < nvalid) ? sp : 0.f;
            dsp[i] = dspv;
            dy[i] = p.z ? go[i] * zv[i] * zsig[i] : go[i];
            y[i] = 0.f;    // partial sums over this wave's states
            duv[i] = 0.f;
            ddl[i] = 0.f;
        }
        float dAq[NS];
#pragma unroll
        for (int s2 = 0; s2 < NS; s2 += 2) {
            float a2v[2], h0v[2], g0v[2], t2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                a2v[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vA), s2 + j));
                h0v[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vH), s2 + j));
                g0v[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vG), s2 + j));
            }
            bwd_states<K, 2, true>(dl, uv, dy, y, duv, ddl, sB + (n0 + s2) * T + tl, sC + (n0 + s2) * T + tl,
                                   accB + s2 * K, accC + s2 * K, T, a2v, h0v, g0v, t2, lane);
            dAq[s2] = t2[0];
            dAq[s2 + 1] = t2[1];
        }
        // meet the other three state-quarters
        float *mine = xch + ((par * 4 + w) * 3 * K) * 64 + lane;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            mine[i * 64] = y[i];
            mine[(K + i) * 64] = duv[i];
            mine[(2 * K + i) * 64] = ddl[i];
        }
        __syncthreads();
        float ty[K], tdu[K], tdd[K];
#pragma unroll
        for (int i = 0; i < K; ++i) ty[i] = tdu[i] = tdd[i] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float *o = xch + ((par * 4 + q) * 3 * K) * 64 + lane;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                ty[i] += o[i * 64];
                tdu[i] += o[(K + i) * 64];
                tdd[i] += o[(2 * K + i) * 64];
            }
        }
        par ^= 1;
        float *part = p.part + (((long)b * p.n_chunks + c) * p.dim + d) * (N + 2);
        {   // dA of this wave's 4 states
            float v = dAq[0];
            v = lane == 1 ? dAq[1] : v;
            v = lane == 2 ? dAq[2] : v;
            v = lane == 3 ? dAq[3] : v;
            if (lane < NS) part[n0 + lane] = v;
        }
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < K; ++i) tdu[i] = fmaf(Dv, dy[i], tdu[i]);
            store_k<io_t, K, FULL>((io_t *)p.du + (long)b * p.du_bs + (long)d * p.du_ds + t0 + tl, nvalid, p.vec_io,
                                   tdu);
        } else if (w == 1) {
            float dbp = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                tdd[i] *= dsp[i];
                if (FULL || i < nvalid) dbp += tdd[i];
            }
            store_k<io_t, K, FULL>((io_t *)p.ddelta + (long)b * p.ddelta_bs + (long)d * p.ddelta_ds + t0 + tl, nvalid,
                                   p.vec_io, tdd);
            dbp = wave_sum(dbp);
            if (lane == 0) part[N + 1] = dbp;
        } else if (w == 2) {
            if (p.z) {
                float dzv[K];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const float yt = fmaf(Dv, uv[i], ty[i]);
                    dzv[i] = go[i] * yt * zsig[i] * (1.f + zv[i] * (1.f - zsig[i]));
                }
                store_k<io_t, K, FULL>((io_t *)p.dz + (long)b * p.dz_bs + (long)d * p.dz_ds + t0 + tl, nvalid,
                                       p.vec_io, dzv);
            }
        } else {
            float dDp = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) dDp = fmaf(dy[i], uv[i], dDp);
            dDp = wave_sum(dDp);
            if (lane == 0) part[N] = dDp;
            if (p.z && p.out_z) {
                float ozv[K];
#pragma unroll
                for (int i = 0; i < K; ++i) ozv[i] = fmaf(Dv, uv[i], ty[i]) * zv[i] * zsig[i];
                store_k<io_t, K, FULL>((io_t *)p.out_z + (long)b * p.out_z_bs + (long)d * p.out_z_ds + t0 + tl, nvalid,
                                       p.vec_io, ozv);
            }
        }
    }
    // this wave's 4 rows of dB / dC
    float *dBg = p.dB + (long)b * p.dB_bs + (long)g * p.dB_gs;
    float *dCg = p.dC + (long)b * p.dC_bs + (long)g * p.dC_gs;
#pragma unroll
    for (int s1 = 0; s1 < NS; ++s1) {
        float vb[K], vc[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            vb[i] = accB[s1 * K + i];
            vc[i] = accC[s1 * K + i];
        }
        store_k<float, K, false>(dBg + (long)(n0 + s1) * p.dB_ns + t0 + tl, nvalid, p.vec_dbc, vb);
        store_k<float, K, false>(dCg + (long)(n0 + s1) * p.dC_ns + t0 + tl, nvalid, p.vec_dbc, vc);
    }
}

// ---------------------------------------------------------------------------
// K4p: backward apply for dstate == 16 on full, aligned 256-token tiles -- the state-split design of K4s
// (4 waves = the 4 state-quarters, register dB/dC sums, LDS exchange of the per-token partial sums, one
// output stream per wave, no atomics) re-cut with what the forward kernels taught:
//   * 4 tokens per lane (both cross-lane scans amortised over twice the tokens), state PAIRS on packed math;
//   * buffer-resource addressing and a branch-free channel pipeline (everything a channel needs is one
//     group of loads issued a channel ahead; no per-stream pointers, no vmcnt(0) in the loop);
//   * the exchange barrier orders LDS only (MMU_LDS_BARRIER): __syncthreads() would drain the prefetch;
//   * each wave stages only the B / C rows of ITS four states (private LDS region, no barrier).
// grid (L / 256, batch, ngroups), block 256.  A tile is two 128-token chunks: carries come from x[2t-1]
// and gx[2t+2]; the dA/dD/dbias partial row of the second chunk is written as zeros.
// LDS: BC[4 waves][B|C][2 pairs][2][64] float4 | slots[4][2][16] | xch[2][4][3][64] float4   (56.5 KiB;
// two workgroups per CU either way: 247 VGPRs)
// ---------------------------------------------------------------------------
template <typename io_t, bool HAS_Z>
__global__ __launch_bounds__(256, 2) void chunk_apply_bwd_p4_kernel(ScanArgs p) {
    constexpr int N = 16, TT = 256;
    constexpr unsigned ES = sizeof(io_t);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile = blockIdx.x, b = blockIdx.y;
    const int g = blockIdx.z / p.d_splits, sp = blockIdx.z - g * p.d_splits;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n0 = w * 4;
    const int t0 = tile * TT;
    const int c0 = tile * 2;
    float4 *sBC = reinterpret_cast<float4 *>(smem) + w * 512;
    float *slots = smem + 4 * 512 * 4 + w * 32;
    float4 *xch = reinterpret_cast<float4 *>(smem + 4 * 512 * 4 + 4 * 32);

    {   // this wave's four B rows and four C rows, pair-interleaved: float4 q of lane l =
        // (row 2p [4l+2q], row 2p+1 [4l+2q], row 2p [4l+2q+1], row 2p+1 [4l+2q+1])
        const io_t *Bg = (const io_t *)p.B + (long)b * p.B_bs + (long)g * p.B_gs + t0 + lane * 4;
        const io_t *Cg = (const io_t *)p.C + (long)b * p.C_bs + (long)g * p.C_gs + t0 + lane * 4;
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const int n = n0 + 2 * pi;
            float r0[4], r1[4];
            load_k<io_t, 4, true>(Bg + (long)n * p.B_ns, 4, true, r0);
            load_k<io_t, 4, true>(Bg + (long)(n + 1) * p.B_ns, 4, true, r1);
            sBC[(pi * 2 + 0) * 64 + lane] = make_float4(r0[0], r1[0], r0[1], r1[1]);
            sBC[(pi * 2 + 1) * 64 + lane] = make_float4(r0[2], r1[2], r0[3], r1[3]);
            load_k<io_t, 4, true>(Cg + (long)n * p.C_ns, 4, true, r0);
            load_k<io_t, 4, true>(Cg + (long)(n + 1) * p.C_ns, 4, true, r1);
            sBC[256 + (pi * 2 + 0) * 64 + lane] = make_float4(r0[0], r1[0], r0[1], r1[1]);
            sBC[256 + (pi * 2 + 1) * 64 + lane] = make_float4(r0[2], r1[2], r0[3], r1[3]);
        }
    }

    const int dpg = p.dim / p.ngroups, cps = (dpg + p.d_splits - 1) / p.d_splits;   // host: no range is empty
    const int dbeg = g * dpg + sp * cps, dend = min(dbeg + cps, (g + 1) * dpg);
    const rsrc_t r_delta = make_rsrc((const io_t *)p.delta + (long)b * p.delta_bs + t0);
    const rsrc_t r_u = make_rsrc((const io_t *)p.u + (long)b * p.u_bs + t0);
    const rsrc_t r_go = make_rsrc((const io_t *)p.dout + (long)b * p.dout_bs + t0);
    const rsrc_t r_z = make_rsrc(HAS_Z ? (const io_t *)p.z + (long)b * p.z_bs + t0 : (const io_t *)p.u);
    const rsrc_t r_part = make_rsrc(p.part + ((long)b * p.n_chunks + c0) * p.dim * (N + 2));
    // the one output stream this wave owns: du | ddelta | dz | out_z
    const rsrc_t r_o = make_rsrc(w == 0   ? (io_t *)p.du + (long)b * p.du_bs + t0
                                 : w == 1 ? (io_t *)p.ddelta + (long)b * p.ddelta_bs + t0
                                 : w == 2 ? (HAS_Z ? (io_t *)p.dz + (long)b * p.dz_bs + t0 : (io_t *)p.du)
                                          : (p.out_z ? (io_t *)p.out_z + (long)b * p.out_z_bs + t0 : (io_t *)p.du));
    const unsigned o_ds = w == 0 ? p.du_ds : w == 1 ? p.ddelta_ds : w == 2 ? p.dz_ds : p.out_z_ds;
    const bool hasH = c0 > 0, hasG = c0 + 2 < p.n_chunks;
    const unsigned voff = lane * 4 * ES;
    // The 14 scalars a channel needs (A, forward and adjoint carry-in of this wave's 4 states, bias, D) come
    // from ONE gathered load: lane j < 14 owns one of them as a 64-bit base + a per-channel stride (five more
    // buffer descriptors would not fit the SGPR file next to the six streams; spilled SGPRs are v_readlanes
    // in the loop).  Lanes 14..63 re-read lane 0's word.
    const int st = n0 + (lane & 3);
    const float *gbase = p.A + (long)st * p.A_ns;
    unsigned gstride = (unsigned)p.A_ds;
    float gscale = MMU_LOG2E;
    if (lane >= 4 && lane < 8) {
        gbase = p.x + ((long)b * p.dim * p.n_chunks + (hasH ? c0 - 1 : 0)) * 2 * N + 2 * st + 1;
        gstride = (unsigned)p.n_chunks * 2 * N;
        gscale = hasH ? 1.f : 0.f;
    } else if (lane >= 8 && lane < 12) {
        gbase = p.gx + ((long)b * p.dim * p.n_chunks + (hasG ? c0 + 2 : 0)) * 2 * N + 2 * st + 1;
        gstride = (unsigned)p.n_chunks * 2 * N;
        gscale = hasG ? 1.f : 0.f;
    } else if (lane == 12) {
        gbase = p.delta_bias ? p.delta_bias : p.A;
        gstride = p.delta_bias ? 1u : 0u;
        gscale = p.delta_bias ? 1.f : 0.f;
    } else if (lane == 13) {
        gbase = p.D ? p.D : p.A;
        gstride = p.D ? 1u : 0u;
        gscale = p.D ? 1.f : 0.f;
    }

    float dl_n[4], u_n[4], go_n[4], z_n[4], gv;
    auto fetch = [&](int dc) {
        dc = __builtin_amdgcn_readfirstlane(dc);
        buf_load4<io_t>(r_delta, voff, dc * (unsigned)p.delta_ds * ES, dl_n);
        buf_load4<io_t>(r_u, voff, dc * (unsigned)p.u_ds * ES, u_n);
        buf_load4<io_t>(r_go, voff, dc * (unsigned)p.dout_ds * ES, go_n);
        if constexpr (HAS_Z) buf_load4<io_t>(r_z, voff, dc * (unsigned)p.z_ds * ES, z_n);
        gv = gbase[(unsigned long)dc * gstride];
    };
    // slots[par][16]: 0..3 A*log2e | 4..7 h0 | 8..11 g0 | 12 bias | 13 D
    auto put_slots = [&](int par) {
        if (lane < 14) slots[par * 16 + lane] = gv * gscale;
    };
    fetch(dbeg);
    asm volatile("" : "+v"(gv));  // consume the whole group before the loop (see chunk_apply_fwd8)
    put_slots(0);

    v2f accB[2][4], accC[2][4];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
        for (int i = 0; i < 4; ++i) accB[pi][i] = accC[pi][i] = v2f{0.f, 0.f};

    int par = 0;
    for (int d = dbeg; d < dend; ++d, par ^= 1) {
        const float *sl = slots + par * 16;
        const float bias = sl[12], Dv = sl[13];
        float vraw[4], dl[4], uv[4], go[4], zv[4], dy[4], zsig[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vraw[i] = dl_n[i] + bias;
            uv[i] = u_n[i];
            go[i] = go_n[i];
            zv[i] = HAS_Z ? z_n[i] : 0.f;
        }
        fetch(d + 1 < dend ? d + 1 : d);  // next channel (the last iteration re-reads its own rows: L2 hits)
        const unsigned dcur = __builtin_amdgcn_readfirstlane(d);
#pragma unroll
        for (int i = 0; i < 4; ++i) dl[i] = vraw[i];
        if (p.softplus) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dl[i] = softplus_thr(vraw[i]);
        }
        float dlsum = 0.f;
        v2f dl2[2], u2[2], dlu2[2], dy2[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (HAS_Z) {
                zsig[i] = sigmoidf_(zv[i]);
                dy[i] = go[i] * zv[i] * zsig[i];
            } else {
                zsig[i] = 0.f;
                dy[i] = go[i];
            }
            dlsum += dl[i];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            dl2[k] = v2f{dl[2 * k], dl[2 * k + 1]};
            u2[k] = v2f{uv[2 * k], uv[2 * k + 1]};
            dlu2[k] = dl2[k] * u2[k];
            dy2[k] = v2f{dy[2 * k], dy[2 * k + 1]};
        }
        v2f y2[4], du2[4], dd2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) y2[i] = du2[i] = dd2[i] = v2f{0.f, 0.f};
        float dAq[4];
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const v2f a2 = *reinterpret_cast<const v2f *>(sl + 2 * pi);
            const v2f h0 = *reinterpret_cast<const v2f *>(sl + 4 + 2 * pi);
            const v2f g0 = *reinterpret_cast<const v2f *>(sl + 8 + 2 * pi);
            v2f a[4], bb[4], hh[4];  // bb becomes a_t * h_{t-1} in the forward pass
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                a[2 * k] = exp2_2(mul_bcast<0>(dl2[k], a2));
                a[2 * k + 1] = exp2_2(mul_bcast<1>(dl2[k], a2));
            }
            const v2f P = exp2_2(a2 * dlsum);
            v2f S, R;
            {   // local composition of this lane's 4 tokens, forward (S) and adjoint (R, right-to-left:
                // gamma_out = a_i (c_i + gamma_in)); the B / C values are re-read in the gamma loop below
                // instead of being kept (16 registers through both scans)
                const float4 f0 = sBC[(pi * 2 + 0) * 64 + lane], f1 = sBC[(pi * 2 + 1) * 64 + lane];
                const float4 c0_ = sBC[256 + (pi * 2 + 0) * 64 + lane], c1_ = sBC[256 + (pi * 2 + 1) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                bb[0] = mul_bcast<0>(dlu2[0], v2f{f0.x, f0.y});
                bb[1] = mul_bcast<1>(dlu2[0], v2f{f0.z, f0.w});
                bb[2] = mul_bcast<0>(dlu2[1], v2f{f1.x, f1.y});
                bb[3] = mul_bcast<1>(dlu2[1], v2f{f1.z, f1.w});
                S = bb[0];
#pragma unroll
                for (int i = 1; i < 4; ++i) S = fma2(a[i], S, bb[i]);
                R = a[3] * mul_bcast<1>(dy2[1], v2f{c1_.z, c1_.w});
                R = a[2] * fma_bcast<0>(dy2[1], v2f{c1_.x, c1_.y}, R);
                R = a[1] * fma_bcast<1>(dy2[0], v2f{c0_.z, c0_.w}, R);
                R = a[0] * fma_bcast<0>(dy2[0], v2f{c0_.x, c0_.y}, R);
            }
            const v2f S_in = fma2(P, h0, S), R_in = fma2(P, g0, R);
            S = lane == 0 ? S_in : S;
            R = lane == 63 ? R_in : R;
            float P0 = P.x, S0 = S.x, P1 = P.y, S1 = S.y;
            float Q0 = wave_reverse(P.x), R0 = wave_reverse(R.x), Q1 = wave_reverse(P.y), R1 = wave_reverse(R.y);
            wave_scan_affine_x2(P0, S0, P1, S1);
            wave_scan_affine_x2(Q0, R0, Q1, R1);
            v2f h = v2f{wave_shift_up1(S0, h0.x), wave_shift_up1(S1, h0.y)};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v2f ah = a[i] * h;  // a_t * h_{t-1}
                h = ah + bb[i];
                bb[i] = ah;
                hh[i] = h;
            }
            // gamma entering this lane from the right = reversed-order inclusive R of lane + 1
            v2f gam;
            gam.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((62 - lane) << 2, __builtin_bit_cast(int, R0)));
            gam.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((62 - lane) << 2, __builtin_bit_cast(int, R1)));
            gam = lane == 63 ? g0 : gam;
            const v2f An = a2 * MMU_LN2;
            v2f dAp = v2f{0.f, 0.f};
#pragma unroll
            for (int k = 1; k >= 0; --k) {
                const float4 fb = sBC[(pi * 2 + k) * 64 + lane], fc = sBC[256 + (pi * 2 + k) * 64 + lane];
#pragma unroll
                for (int hsel = 1; hsel >= 0; --hsel) {
                    const int i = 2 * k + hsel;
                    const v2f Bt = hsel ? v2f{fb.z, fb.w} : v2f{fb.x, fb.y};
                    const v2f Ct = hsel ? v2f{fc.z, fc.w} : v2f{fc.x, fc.y};
                    const v2f gt = hsel ? fma_bcast<1>(dy2[k], Ct, gam) : fma_bcast<0>(dy2[k], Ct, gam);
                    gam = a[i] * gt;
                    // With q_t = sum_n g B (summed over the states, then over the four waves):
                    //   du = delta q + D dy ;  ddelta' = u q + sum_n A (g a h_prev) ;  dA = sum_t delta (g a h_prev) ;
                    //   dB = (delta u) g  -- two packed instructions per token and pair fewer than forming
                    //   g delta and u B first; the q -> du / ddelta step runs once per token in the epilogue.
                    const v2f gah = gt * bb[i];  // g_t * a_t h_{t-1}
                    du2[i] = fma2(gt, Bt, du2[i]);          // q
                    dd2[i] = fma2(An, gah, dd2[i]);
                    dAp = hsel ? fma_bcast<1>(dl2[k], gah, dAp) : fma_bcast<0>(dl2[k], gah, dAp);
                    y2[i] = fma2(Ct, hh[i], y2[i]);
                    accB[pi][i] = hsel ? fma_bcast<1>(dlu2[k], gt, accB[pi][i]) : fma_bcast<0>(dlu2[k], gt, accB[pi][i]);
                    accC[pi][i] = hsel ? fma_bcast<1>(dy2[k], hh[i], accC[pi][i]) : fma_bcast<0>(dy2[k], hh[i], accC[pi][i]);
                }
            }
            dAq[2 * pi] = dAp.x;      // per-lane partials; reduced over the wave below, all four at once
            dAq[2 * pi + 1] = dAp.y;
            __builtin_amdgcn_sched_barrier(0);  // one pair at a time: interleaving two doubles the live registers
        }
        put_slots(par ^ 1);  // (the prefetch group has landed long ago)

        // meet the other three state-quarters (exchange buffer double-buffered by channel parity: one barrier)
        float4 *xc = xch + par * (4 * 3 * 64);
        xc[(w * 3 + 0) * 64 + lane] = make_float4(y2[0].x + y2[0].y, y2[1].x + y2[1].y, y2[2].x + y2[2].y, y2[3].x + y2[3].y);
        xc[(w * 3 + 1) * 64 + lane] = make_float4(du2[0].x + du2[0].y, du2[1].x + du2[1].y, du2[2].x + du2[2].y, du2[3].x + du2[3].y);
        xc[(w * 3 + 2) * 64 + lane] = make_float4(dd2[0].x + dd2[0].y, dd2[1].x + dd2[1].y, dd2[2].x + dd2[2].y, dd2[3].x + dd2[3].y);
        MMU_LDS_BARRIER();
        const int arr = w == 0 ? 1 : (w == 1 ? 2 : 0);   // rows of the exchange: 0 = y, 1 = q, 2 = sum_n A g a h_prev
        float tot[4] = {0.f, 0.f, 0.f, 0.f}, totq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 o = xc[(q * 3 + arr) * 64 + lane];
            tot[0] += o.x; tot[1] += o.y; tot[2] += o.z; tot[3] += o.w;
        }
        if (w == 1) {  // d(delta) needs q as well
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 o = xc[(q * 3 + 1) * 64 + lane];
                totq[0] += o.x; totq[1] += o.y; totq[2] += o.z; totq[3] += o.w;
            }
        }
        // partial sums of dA (this wave's 4 states), dD, dbias: row of chunk c0; zeros for chunk c0 + 1
        const unsigned prow = dcur * (unsigned)(N + 2) * 4u, prow1 = prow + (unsigned)p.dim * (N + 2) * 4u;
        {
            const float v = wave_sum4(dAq[0], dAq[1], dAq[2], dAq[3]);   // lanes 12..15: totals of states n0..n0+3
            if (lane >= 12 && lane < 16) {
                buf_store1(r_part, (n0 + lane - 12) * 4u, prow, v);
                buf_store1(r_part, (n0 + lane - 12) * 4u, prow1, 0.f);
            }
        }
        float ov[4];
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ov[i] = fmaf(Dv, dy[i], dl[i] * tot[i]);   // du = delta q + D dy
            buf_store4<io_t>(r_o, voff, dcur * o_ds * ES, ov);
        } else if (w == 1) {
            float dbp = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dsp = (p.softplus && vraw[i] <= 20.f) ? sigmoidf_(vraw[i]) : 1.f;  // bwd_kernel.cuh:439-453
                ov[i] = fmaf(uv[i], totq[i], tot[i]) * dsp;                          // ddelta' = u q + sum_n A g a h_prev
                dbp += ov[i];
            }
            buf_store4<io_t>(r_o, voff, dcur * o_ds * ES, ov);
            dbp = wave_sum(dbp);
            if (lane == 0) {
                buf_store1(r_part, (N + 1) * 4u, prow, dbp);
                buf_store1(r_part, (N + 1) * 4u, prow1, 0.f);
            }
        } else if (w == 2) {
            if constexpr (HAS_Z) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float yt = fmaf(Dv, uv[i], tot[i]);
                    ov[i] = go[i] * yt * zsig[i] * (1.f + zv[i] * (1.f - zsig[i]));
                }
                buf_store4<io_t>(r_o, voff, dcur * o_ds * ES, ov);
            }
        } else {
            float dDp = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) dDp = fmaf(dy[i], uv[i], dDp);
            dDp = wave_sum(dDp);
            if (lane == 0) {
                buf_store1(r_part, N * 4u, prow, dDp);
                buf_store1(r_part, N * 4u, prow1, 0.f);
            }
            if (HAS_Z && p.out_z) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = fmaf(Dv, uv[i], tot[i]) * zv[i] * zsig[i];
                buf_store4<io_t>(r_o, voff, dcur * o_ds * ES, ov);
            }
        }
    }
    // this wave's 4 rows of dB / dC
    float *dBg = p.dB + sp * p.dBC_ss + (long)b * p.dB_bs + (long)g * p.dB_gs + t0 + lane * 4;
    float *dCg = p.dC + sp * p.dBC_ss + (long)b * p.dC_bs + (long)g * p.dC_gs + t0 + lane * 4;
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
        const int n = n0 + 2 * pi;
        *reinterpret_cast<float4 *>(dBg + (long)n * p.dB_ns) = make_float4(accB[pi][0].x, accB[pi][1].x, accB[pi][2].x, accB[pi][3].x);
        *reinterpret_cast<float4 *>(dBg + (long)(n + 1) * p.dB_ns) = make_float4(accB[pi][0].y, accB[pi][1].y, accB[pi][2].y, accB[pi][3].y);
        *reinterpret_cast<float4 *>(dCg + (long)n * p.dC_ns) = make_float4(accC[pi][0].x, accC[pi][1].x, accC[pi][2].x, accC[pi][3].x);
        *reinterpret_cast<float4 *>(dCg + (long)(n + 1) * p.dC_ns) = make_float4(accC[pi][0].y, accC[pi][1].y, accC[pi][2].y, accC[pi][3].y);
    }
}

// ---------------------------------------------------------------------------
// K5: dA[d][n], dD[d], dbias[d] += sum over a slice of (b, chunk) of part[b][c][d][N+2].
// grid (dim, n_slices = ceil(BC / 512)), block 256.  One slice: results go straight to dA/dD/dbias.
// Several slices: slice sums go to part2[slice][d][N+2] and reduce_slices_kernel adds them in fixed
// order -- no float atomics, so dA/dD/dbias stay bit-reproducible.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ part, int BC, int dim, int N,
                                                              float *dA, float *dD, float *dbias,
                                                              float *__restrict__ part2, const float *__restrict__ Asc) {
    __shared__ float red[256];
    const int d = blockIdx.x;
    const int bc0 = blockIdx.y * 512;
    const int bc1 = bc0 + 512 < BC ? bc0 + 512 : BC;
    const int M = N + 2;
    for (int j0 = 0; j0 < M; j0 += 32) {
        const int j = j0 + (threadIdx.x & 31);
        const int r = threadIdx.x >> 5;
        float s = 0.f;
        if (j < M)
            for (int bc = bc0 + r; bc < bc1; bc += 8) s += part[((long)bc * dim + d) * M + j];
        red[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < 32) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * 32 + threadIdx.x];
            if (part2) {
                if (j < M) part2[((long)blockIdx.y * dim + d) * M + j] = t;
            } else if (j < N) {
                dA[(long)d * N + j] = Asc ? t * Asc[(long)d * N + j] : t;
            } else if (j == N) {
                if (dD) dD[d] = t;
            } else if (j == N + 1) {
                if (dbias) dbias[d] = t;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void reduce_slices_kernel(const float *__restrict__ part2, int n_slices, int dim, int N,
                                                           float *dA, float *dD, float *dbias,
                                                           const float *__restrict__ Asc) {
    const int d = blockIdx.x, M = N + 2;
    for (int j = threadIdx.x; j < M; j += 64) {
        float t = 0.f;
        for (int sl = 0; sl < n_slices; ++sl) t += part2[((long)sl * dim + d) * M + j];
        if (j < N)
            dA[(long)d * N + j] = Asc ? t * Asc[(long)d * N + j] : t;
        else if (j == N) {
            if (dD) dD[d] = t;
        } else if (dbias)
            dbias[d] = t;
    }
}

// ---------------------------------------------------------------------------
// debug: wave scan primitives
// ---------------------------------------------------------------------------
__global__ void debug_wave_scan_kernel(const float *P, const float *S, float *oP, float *oS, int reverse,
                                       int variant) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    float p = P[i], s = S[i];
    if (reverse) {
        p = wave_reverse(p);
        s = wave_reverse(s);
    }
    if (variant == 0) {
        wave_scan_affine_dpp(p, s);      // compiler-lowered DPP intrinsics
    } else if (variant == 1) {
        wave_scan_affine_shfl(p, s);     // shuffle reference
    } else if (variant == 2) {
        wave_scan_affine(p, s);          // hand-written fused DPP, one scan
    } else {
        float p2 = p * 0.5f, s2 = -s;    // hand-written fused DPP, two interleaved scans
        wave_scan_affine_x2(p2, s2, p, s);
        if (variant == 4) {              // return the first of the pair instead
            p = p2;
            s = s2;
        }
    }
    if (reverse) {
        p = wave_reverse(p);
        s = wave_reverse(s);
    }
    oP[i] = p;
    oS[i] = s;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// tokens per lane of the chunk kernels (chunk = 64 * this).  dstate <= 16 uses 2 so that the backward's
// per-lane dB/dC register accumulators (2 * dstate * K) fit next to its working set at 2+ waves/SIMD.
inline int items_per_lane(int dstate) { return dstate <= 16 ? 2 : (dstate <= 32 ? 4 : (dstate <= 64 ? 2 : 1)); }

inline bool aligned_to(const void *p, size_t a) { return p == nullptr || ((uintptr_t)p % a) == 0; }
inline bool mult(long v, int k) { return (v % k) == 0; }


// K1 in its fast (8 tokens / lane, 4 chunks per wave) or generic form
template <typename io_t, int K, bool BWD>
int launch_reduce(const ScanArgs &a, int W, hipStream_t st) {
    const int N = a.dstate, T = 64 * K;
    const long span = (long)a.dim * std::max({a.u_ds, a.delta_ds, BWD ? a.dout_ds : 0L, (BWD && a.z) ? a.z_ds : 0L}) + a.seqlen;
    if (K == 2 && N <= 16 && a.vec_io && a.vec_bc && a.seqlen % 512 == 0 && span * 4 < (1L << 31) &&
        (long)a.dim * a.n_chunks * 2 * N * 4 < (1L << 31)) {
        // fast form: full aligned 512-token tiles, buffer addressing (32-bit offsets inside a batch item)
        const size_t NE = (N + 1) & ~1;  // pair tiles pad an odd dstate
        const int W8 = W > 8 ? 8 : W;    // several workgroups per CU: one stages its tile while the others compute
        size_t lds = sizeof(float) * (NE * 512 + (size_t)W8 * NE * 10);
        dim3 grid(a.seqlen / 512, a.batch, a.ngroups);
        MMU_BOOL(BWD && a.z != nullptr, HAS_Z, {
            if (int r = set_lds(chunk_reduce8_kernel<io_t, BWD, HAS_Z>, lds)) return r;
            chunk_reduce8_kernel<io_t, BWD, HAS_Z><<<grid, W8 * 64, lds, st>>>(a);
        });
    } else {
        const bool full = a.vec_io && a.vec_bc && a.seqlen % T == 0;
        size_t lds = sizeof(float) * ((size_t)N * T + (size_t)W * N * 3);
        dim3 grid(a.n_chunks, a.batch, a.ngroups);
        MMU_BOOL(full, FULL, {
            if (int r = set_lds(chunk_reduce_kernel<io_t, K, BWD, FULL>, lds)) return r;
            chunk_reduce_kernel<io_t, K, BWD, FULL><<<grid, W * 64, lds, st>>>(a);
        });
    }
    MMU_HIP_LAUNCH_CHECK(BWD ? "chunk_reduce<bwd>" : "chunk_reduce<fwd>");
    return 0;
}

template <typename io_t, int K>
int launch_fwd(const ScanArgs &a, hipStream_t st) {
    const int N = a.dstate, T = 64 * K;
    const int dpg = a.dim / a.ngroups;
    const int W = dpg < 16 ? dpg : 16;
    dim3 grid(a.n_chunks, a.batch, a.ngroups);
    const bool full = a.vec_io && a.vec_bc && a.seqlen % T == 0;  // no ragged tail, everything aligned
    // enough (batch, channel) rows to fill the chip: stream each row front to back (selective_scan_stream.hip)
    if (const int r = mmu_scan_fwd_stream(a, sizeof(io_t) == 4 ? MMU_DTYPE_F32 : MMU_DTYPE_BF16, st)) return r < 0;
    if (int r = launch_reduce<io_t, K, false>(a, W, st)) return r;
    if (int r = launch_carry(a.x, a.batch, a.dim, N, a.n_chunks, 0, st)) return r;
    {
        // 512-token tiles on packed state pairs (buffer addressing: 32-bit offsets inside a batch item)
        const long span = (long)a.dim * std::max({a.u_ds, a.delta_ds, a.z ? a.z_ds : 0L, a.out ? a.out_ds : 0L,
                                                  a.out_z ? a.out_z_ds : 0L}) + a.seqlen;
        if (K == 2 && N <= 16 && full && a.seqlen % 512 == 0 && span * 4 < (1L << 31) && W >= 1) {
            dim3 gridw(a.seqlen / 512, a.batch, a.ngroups);
            const int Ww = W > 8 ? 8 : W;
            const size_t NE = (N + 1) & ~1;
            const size_t lds = sizeof(float) * (2 * NE * 512 + (size_t)Ww * NE * 4);
            MMU_BOOL(a.z != nullptr, HAS_Z, {
                if (int r = set_lds(chunk_apply_fwd8_kernel<io_t, HAS_Z>, lds)) return r;
                chunk_apply_fwd8_kernel<io_t, HAS_Z><<<gridw, Ww * 64, lds, st>>>(a);
            });
        } else {
            size_t lds = sizeof(float) * ((size_t)2 * N * T + (size_t)W * N * 2);
            MMU_BOOL(full, FULL, {
                if (int r = set_lds(chunk_apply_fwd_kernel<io_t, K, FULL>, lds)) return r;
                chunk_apply_fwd_kernel<io_t, K, FULL><<<grid, W * 64, lds, st>>>(a);
            });
        }
        MMU_HIP_LAUNCH_CHECK("chunk_apply_fwd");
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Channel-range splits of the fast backward apply kernels.  A launch with few tiles (L = 4,096, batch 8: 128
// workgroups of the 256-token kernel on 256 CUs) leaves every workgroup a serial walk over all channels of its
// group; cutting the channels into S ranges (grid.z = S) gives S times the workgroups.  Each range sums dB / dC
// over its own channels into its own [B][N][L] slab of the workspace; sum_splits_kernel adds the slabs in a fixed
// order.  Only for ngroups == 1 (MM-UNet's case: the workspace query does not know the group count).
// ---------------------------------------------------------------------------
inline int bwd_d_splits(int batch, int dim, int seqlen, int dstate, int ngroups) {
    if (dstate != 16 || ngroups != 1 || seqlen % 256 != 0 || dim < 16) return 1;
    const long wgs = (long)(seqlen / 256) * batch;   // workgroups of the 256-token kernel
    const long want = 2L * mmu_cu_count();
    if (wgs >= want) return 1;
    long S = (want + wgs - 1) / wgs;
    if (S > 8) S = 8;
    if (S > dim / 8) S = dim / 8;                    // at least 8 channels per range
    if (S < 2) return 1;
    const int cps = (int)((dim + S - 1) / S);
    return (dim + cps - 1) / cps;                    // no empty range
}

// out[b][n][t] (strides bs, ns) = sum over s of part[s][b][n][t];  L % 4 == 0.  grid ceil(B*N*L/4 / 256), block 256
__global__ __launch_bounds__(256) void sum_splits_kernel(const float *__restrict__ part, int S, long ss, int B, int N, int L,
                                                         float *__restrict__ out, long bs, long ns) {
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
    const long total4 = (long)B * N * L / 4;
    if (i4 >= total4) return;
    const long i = i4 * 4;
    const int t = (int)(i % L);
    const long r = i / L;
    const int n = (int)(r % N), b = (int)(r / N);
    float4 acc = *reinterpret_cast<const float4 *>(part + i);
    for (int s1 = 1; s1 < S; ++s1) {
        const float4 v = *reinterpret_cast<const float4 *>(part + s1 * ss + i);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4 *>(out + b * bs + n * ns + t) = acc;
}

template <typename io_t, int K>
int launch_bwd(ScanArgs a, bool have_x, float *ws, hipStream_t st, bool *w8_layout) {
    const int N = a.dstate, T = 64 * K;
    const int dpg = a.dim / a.ngroups;
    dim3 grid(a.n_chunks, a.batch, a.ngroups);
    const bool full = a.vec_io && a.vec_bc && a.seqlen % T == 0;
    const size_t xs = (size_t)a.batch * a.dim * a.n_chunks * 2 * N;
    a.gx = ws;
    a.part = ws + xs;
    const int W16 = dpg < 16 ? dpg : 16;
    if (!have_x) {
        a.x = ws + xs + (size_t)a.batch * a.n_chunks * a.dim * (N + 2);
        if (int r = launch_reduce<io_t, K, false>(a, W16, st)) return r;
        if (int r = launch_carry(a.x, a.batch, a.dim, N, a.n_chunks, 0, st)) return r;
    }
    if (int r = launch_reduce<io_t, K, true>(a, W16, st)) return r;
    if (int r = launch_carry(a.gx, a.batch, a.dim, N, a.n_chunks, 1, st)) return r;
    const auto al16 = [](const void *q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    const long span = (long)a.dim * std::max({a.u_ds, a.delta_ds, a.dout_ds, a.du_ds, a.ddelta_ds, a.z ? a.z_ds : 0L,
                                              a.z ? a.dz_ds : 0L, a.out_z ? a.out_z_ds : 0L}) + a.seqlen;
    const bool rows4 = mult(a.u_bs, 4) && mult(a.u_ds, 4) && mult(a.delta_bs, 4) && mult(a.delta_ds, 4) &&
                       mult(a.dout_bs, 4) && mult(a.dout_ds, 4) && mult(a.du_bs, 4) && mult(a.du_ds, 4) &&
                       mult(a.ddelta_bs, 4) && mult(a.ddelta_ds, 4) &&
                       (!a.z || (mult(a.z_bs, 4) && mult(a.z_ds, 4) && mult(a.dz_bs, 4) && mult(a.dz_ds, 4))) &&
                       (!a.out_z || (mult(a.out_z_bs, 4) && mult(a.out_z_ds, 4))) && mult(a.B_bs, 4) &&
                       mult(a.B_gs, 4) && mult(a.B_ns, 4) && mult(a.C_bs, 4) && mult(a.C_gs, 4) && mult(a.C_ns, 4) &&
                       mult(a.dB_bs, 4) && mult(a.dB_gs, 4) && mult(a.dB_ns, 4) && mult(a.dC_bs, 4) &&
                       mult(a.dC_gs, 4) && mult(a.dC_ns, 4);
    const bool ptrs16 = al16(a.u) && al16(a.delta) && al16(a.dout) && al16(a.z) && al16(a.du) && al16(a.ddelta) &&
                        al16(a.dz) && al16(a.out_z) && al16(a.B) && al16(a.C) && al16(a.dB) && al16(a.dC);
    const bool fast = K == 2 && N == 16 && a.seqlen % 256 == 0 && rows4 && ptrs16 && span * 4 < (1L << 31) &&
                      (long)a.dim * a.n_chunks * 2 * N * 4 < (1L << 31) && (long)a.n_chunks * a.dim * (N + 2) * 4 < (1L << 31);
    // few tiles: the channels are cut into ranges, each with its own dB / dC slab behind the other workspace regions
    float *const dB_out = a.dB, *const dC_out = a.dC;
    const long dB_bs = a.dB_bs, dB_ns = a.dB_ns, dC_bs = a.dC_bs, dC_ns = a.dC_ns;
    const int S = fast ? bwd_d_splits(a.batch, a.dim, a.seqlen, N, a.ngroups) : 1;
    if (S > 1) {
        const size_t parts = (size_t)a.batch * a.n_chunks * a.dim * (N + 2);
        const size_t slices = ((size_t)a.batch * a.n_chunks + 511) / 512 * a.dim * (N + 2);
        float *slab = ws + xs + parts + (have_x ? 0 : xs) + slices;
        a.d_splits = S;
        a.dBC_ss = (long)a.batch * N * a.seqlen;
        a.dB = slab;
        a.dC = slab + (size_t)S * a.dBC_ss;
        a.dB_bs = a.dC_bs = (long)N * a.seqlen;
        a.dB_gs = a.dC_gs = 0;
        a.dB_ns = a.dC_ns = a.seqlen;
    }
    const auto join_splits = [&]() -> int {
        if (S == 1) return 0;
        const long total4 = (long)a.batch * N * a.seqlen / 4;
        const unsigned blocks = (unsigned)((total4 + 255) / 256);
        sum_splits_kernel<<<blocks, 256, 0, st>>>(a.dB, S, a.dBC_ss, a.batch, N, a.seqlen, dB_out, dB_bs, dB_ns);
        sum_splits_kernel<<<blocks, 256, 0, st>>>(a.dC, S, a.dBC_ss, a.batch, N, a.seqlen, dC_out, dC_bs, dC_ns);
        MMU_HIP_LAUNCH_CHECK("sum_splits");
        return 0;
    };
    if (fast) {   // 512-token tiles, one state pair per wave (selective_scan_bwd_w8.hip), when the launch fills the chip
        const int r = mmu_scan_bwd_apply_w8(a, sizeof(io_t) == 4 ? MMU_DTYPE_F32 : MMU_DTYPE_BF16, st);
        if (r < 0) return 1;   // the message is already in mmu_last_error()
        if (r == 1) {
            *w8_layout = true;
            return join_splits();
        }
    }
    if (fast) {
        // fast form: full aligned 256-token tiles, packed state pairs, buffer addressing
        const size_t lds = sizeof(float) * (4 * 512 * 4 + 4 * 32 + 2 * 4 * 3 * 64 * 4);
        dim3 gridp(a.seqlen / 256, a.batch, a.ngroups * a.d_splits);
        MMU_BOOL(a.z != nullptr, HAS_Z, {
            if (int r = set_lds(chunk_apply_bwd_p4_kernel<io_t, HAS_Z>, lds)) return r;
            chunk_apply_bwd_p4_kernel<io_t, HAS_Z><<<gridp, 256, lds, st>>>(a);
        });
        MMU_HIP_LAUNCH_CHECK("chunk_apply_bwd_p4");
        if (int r = join_splits()) return r;
    } else if (K == 2 && N == 16) {
        size_t lds = sizeof(float) * ((size_t)2 * N * T + (size_t)2 * 4 * 3 * 2 * 64);
        MMU_BOOL(full, FULL, {
            if (int r = set_lds(chunk_apply_bwd_ns4_kernel<io_t, 2, FULL>, lds)) return r;
            chunk_apply_bwd_ns4_kernel<io_t, 2, FULL><<<grid, 256, lds, st>>>(a);
        });
        MMU_HIP_LAUNCH_CHECK("chunk_apply_bwd_ns4");
    } else {
        const int W = dpg < 8 ? dpg : 8;
        size_t lds = sizeof(float) * ((size_t)4 * N * T + (size_t)W * N * 4);
        MMU_BOOL(full, FULL, {
            if (int r = set_lds(chunk_apply_bwd_kernel<io_t, K, FULL>, lds)) return r;
            chunk_apply_bwd_kernel<io_t, K, FULL><<<grid, W * 64, lds, st>>>(a);
        });
        MMU_HIP_LAUNCH_CHECK("chunk_apply_bwd");
    }
    return 0;  // K5 (reduce_partials) is launched by the extern "C" wrapper, which owns dA/dD/dbias
}

}  // namespace

extern "C" int mmu_scan_chunk_len(int dstate, int dtype) {
    (void)dtype;
    if (dstate < 1 || dstate > 128) return 0;
    return 64 * items_per_lane(dstate);
}

extern "C" size_t mmu_scan_bwd_workspace_bytes(int batch, int dim, int seqlen, int dstate, int dtype, int have_x) {
    const int T = mmu_scan_chunk_len(dstate, dtype);
    if (T == 0) return 0;
    const size_t nc = ((size_t)seqlen + T - 1) / T;
    const size_t xs = (size_t)batch * dim * nc * 2 * dstate;
    const size_t parts = (size_t)batch * nc * dim * (dstate + 2);
    const size_t slices = ((size_t)batch * nc + 511) / 512 * dim * (dstate + 2);
    const int S = bwd_d_splits(batch, dim, seqlen, dstate, 1);   // dB / dC slabs of the channel ranges (ngroups == 1)
    const size_t slabs = S > 1 ? (size_t)2 * S * batch * dstate * seqlen : 0;
    return sizeof(float) * (xs + parts + (have_x ? 0 : xs) + slices + slabs);
}

#define SCAN_COMMON_CHECKS(p)                                                                                       \
    MMU_CHECK((p) != nullptr, "selective_scan: null params");                                                       \
    MMU_CHECK((p)->dtype == MMU_DTYPE_F32 || (p)->dtype == MMU_DTYPE_BF16, "selective_scan: unsupported dtype %d",  \
              (p)->dtype);                                                                                          \
    MMU_CHECK((p)->batch > 0 && (p)->dim > 0 && (p)->seqlen > 0, "selective_scan: empty tensor");                   \
    MMU_CHECK((p)->dstate >= 1 && (p)->dstate <= 128,                                                               \
              "selective_scan only supports state dimension <= 128 (got %d)", (p)->dstate);                         \
    MMU_CHECK((p)->ngroups >= 1 && (p)->dim % (p)->ngroups == 0, "selective_scan: dim must be divisible by ngroups"); \
    MMU_CHECK((p)->u && (p)->delta && (p)->A && (p)->B && (p)->C, "selective_scan: u, delta, A, B, C are required"); \
    {                                                                                                               \
        const int T__ = mmu_scan_chunk_len((p)->dstate, (p)->dtype);                                                \
        MMU_CHECK((p)->n_chunks == ((p)->seqlen + T__ - 1) / T__, "selective_scan: n_chunks must be %d (chunk %d)", \
                  ((p)->seqlen + T__ - 1) / T__, T__);                                                              \
    }

extern "C" int mmu_selective_scan_fwd(const mmu_scan_fwd_params *p, void *stream) {
    SCAN_COMMON_CHECKS(p);
    MMU_CHECK(p->x != nullptr, "selective_scan_fwd: chunk-state tensor x is required");
    MMU_CHECK((p->z == nullptr) == (p->out_z == nullptr), "selective_scan_fwd: out_z must be given iff z is");
    MMU_CHECK(p->out != nullptr || p->out_z != nullptr, "selective_scan_fwd: no output tensor");
    const int K = items_per_lane(p->dstate);
    const size_t es = p->dtype == MMU_DTYPE_F32 ? 4 : 2;
    ScanArgs a = {};
    a.d_splits = 1;
    a.batch = p->batch; a.dim = p->dim; a.seqlen = p->seqlen; a.dstate = p->dstate; a.ngroups = p->ngroups;
    a.n_chunks = p->n_chunks; a.softplus = p->delta_softplus;
    a.u = p->u; a.delta = p->delta; a.z = p->z; a.B = p->B; a.C = p->C; a.A = p->A; a.D = p->D;
    a.delta_bias = p->delta_bias; a.out = p->out; a.out_z = p->out_z; a.x = p->x;
    a.u_bs = p->u_bs; a.u_ds = p->u_ds; a.delta_bs = p->delta_bs; a.delta_ds = p->delta_ds;
    a.z_bs = p->z_bs; a.z_ds = p->z_ds; a.out_bs = p->out_bs; a.out_ds = p->out_ds;
    a.out_z_bs = p->out_z_bs; a.out_z_ds = p->out_z_ds;
    a.A_ds = p->A_ds; a.A_ns = p->A_ns; a.B_bs = p->B_bs; a.B_gs = p->B_gs; a.B_ns = p->B_ns;
    a.C_bs = p->C_bs; a.C_gs = p->C_gs; a.C_ns = p->C_ns;
    const size_t al = es * K;
    a.vec_io = aligned_to(p->u, al) && aligned_to(p->delta, al) && aligned_to(p->z, al) && aligned_to(p->out, al) &&
               aligned_to(p->out_z, al) && mult(p->u_bs, K) && mult(p->u_ds, K) && mult(p->delta_bs, K) &&
               mult(p->delta_ds, K) && (!p->z || (mult(p->z_bs, K) && mult(p->z_ds, K))) &&
               (!p->out || (mult(p->out_bs, K) && mult(p->out_ds, K))) &&
               (!p->out_z || (mult(p->out_z_bs, K) && mult(p->out_z_ds, K)));
    a.vec_bc = aligned_to(p->B, al) && aligned_to(p->C, al) && mult(p->B_bs, K) && mult(p->B_gs, K) &&
               mult(p->B_ns, K) && mult(p->C_bs, K) && mult(p->C_gs, K) && mult(p->C_ns, K);
    hipStream_t st = (hipStream_t)stream;
    if (p->dtype == MMU_DTYPE_F32) {
        if (K == 4) return launch_fwd<float, 4>(a, st);
        if (K == 2) return launch_fwd<float, 2>(a, st);
        return launch_fwd<float, 1>(a, st);
    } else {
        if (K == 4) return launch_fwd<bf16_t, 4>(a, st);
        if (K == 2) return launch_fwd<bf16_t, 2>(a, st);
        return launch_fwd<bf16_t, 1>(a, st);
    }
}

extern "C" int mmu_selective_scan_bwd(const mmu_scan_bwd_params *p, void *stream) {
    SCAN_COMMON_CHECKS(p);
    MMU_CHECK(p->dout && p->du && p->ddelta && p->dA && p->dB && p->dC,
              "selective_scan_bwd: dout, du, ddelta, dA, dB, dC are required");
    MMU_CHECK((p->z == nullptr) == (p->dz == nullptr), "selective_scan_bwd: dz must be given iff z is");
    MMU_CHECK(p->out_z == nullptr || p->z != nullptr, "selective_scan_bwd: out_z recompute needs z");
    MMU_CHECK(p->workspace != nullptr, "selective_scan_bwd: workspace is required");
    const int K = items_per_lane(p->dstate);
    const size_t es = p->dtype == MMU_DTYPE_F32 ? 4 : 2;
    ScanArgs a = {};
    a.d_splits = 1;
    a.batch = p->batch; a.dim = p->dim; a.seqlen = p->seqlen; a.dstate = p->dstate; a.ngroups = p->ngroups;
    a.n_chunks = p->n_chunks; a.softplus = p->delta_softplus;
    a.u = p->u; a.delta = p->delta; a.z = p->z; a.B = p->B; a.C = p->C; a.A = p->A; a.D = p->D;
    a.delta_bias = p->delta_bias; a.dout = p->dout; a.x = const_cast<float *>(p->x);
    a.du = p->du; a.ddelta = p->ddelta; a.dz = p->dz; a.out_z = p->out_z; a.dB = p->dB; a.dC = p->dC;
    // the forward's y (before gating), if it was kept: read by the w8 apply kernel instead of recomputing it (only with z:
    // without z nothing in the backward needs y)
    a.out = p->z ? const_cast<void *>(p->out) : nullptr; a.out_bs = p->out_bs; a.out_ds = p->out_ds;
    a.u_bs = p->u_bs; a.u_ds = p->u_ds; a.delta_bs = p->delta_bs; a.delta_ds = p->delta_ds;
    a.z_bs = p->z_bs; a.z_ds = p->z_ds; a.dout_bs = p->dout_bs; a.dout_ds = p->dout_ds;
    a.du_bs = p->du_bs; a.du_ds = p->du_ds; a.ddelta_bs = p->ddelta_bs; a.ddelta_ds = p->ddelta_ds;
    a.dz_bs = p->dz_bs; a.dz_ds = p->dz_ds; a.out_z_bs = p->out_z_bs; a.out_z_ds = p->out_z_ds;
    a.A_ds = p->A_ds; a.A_ns = p->A_ns; a.B_bs = p->B_bs; a.B_gs = p->B_gs; a.B_ns = p->B_ns;
    a.C_bs = p->C_bs; a.C_gs = p->C_gs; a.C_ns = p->C_ns;
    a.dB_bs = p->dB_bs; a.dB_gs = p->dB_gs; a.dB_ns = p->dB_ns;
    a.dC_bs = p->dC_bs; a.dC_gs = p->dC_gs; a.dC_ns = p->dC_ns;
    a.vec_dbc = aligned_to(p->dB, 4 * K) && aligned_to(p->dC, 4 * K) && mult(p->dB_bs, K) && mult(p->dB_gs, K) &&
                mult(p->dB_ns, K) && mult(p->dC_bs, K) && mult(p->dC_gs, K) && mult(p->dC_ns, K);
    const size_t al = es * K;
    a.vec_io = aligned_to(p->u, al) && aligned_to(p->delta, al) && aligned_to(p->z, al) && aligned_to(p->dout, al) &&
               aligned_to(p->du, al) && aligned_to(p->ddelta, al) && aligned_to(p->dz, al) &&
               aligned_to(p->out_z, al) && mult(p->u_bs, K) && mult(p->u_ds, K) && mult(p->delta_bs, K) &&
               mult(p->delta_ds, K) && mult(p->dout_bs, K) && mult(p->dout_ds, K) && mult(p->du_bs, K) &&
               mult(p->du_ds, K) && mult(p->ddelta_bs, K) && mult(p->ddelta_ds, K) &&
               (!p->z || (mult(p->z_bs, K) && mult(p->z_ds, K) && mult(p->dz_bs, K) && mult(p->dz_ds, K))) &&
               (!p->out_z || (mult(p->out_z_bs, K) && mult(p->out_z_ds, K)));
    a.vec_bc = aligned_to(p->B, al) && aligned_to(p->C, al) && mult(p->B_bs, K) && mult(p->B_gs, K) &&
               mult(p->B_ns, K) && mult(p->C_bs, K) && mult(p->C_gs, K) && mult(p->C_ns, K);
    hipStream_t st = (hipStream_t)stream;
    const bool have_x = p->x != nullptr;
    bool w8_layout = false;
    int r;
    if (p->dtype == MMU_DTYPE_F32) {
        r = K == 4 ? launch_bwd<float, 4>(a, have_x, p->workspace, st, &w8_layout)
                   : (K == 2 ? launch_bwd<float, 2>(a, have_x, p->workspace, st, &w8_layout)
                             : launch_bwd<float, 1>(a, have_x, p->workspace, st, &w8_layout));
    } else {
        r = K == 4 ? launch_bwd<bf16_t, 4>(a, have_x, p->workspace, st, &w8_layout)
                   : (K == 2 ? launch_bwd<bf16_t, 2>(a, have_x, p->workspace, st, &w8_layout)
                             : launch_bwd<bf16_t, 1>(a, have_x, p->workspace, st, &w8_layout));
    }
    if (r) return r;
    const size_t xs = (size_t)p->batch * p->dim * p->n_chunks * 2 * p->dstate;
    const float *Asc = nullptr;   // dA * A wanted: A read as a dense [dim][dstate] matrix
    if (p->dA_times_A) {
        MMU_CHECK(p->A_ns == 1 && p->A_ds == p->dstate, "selective_scan_bwd: dA_times_A needs a contiguous A");
        Asc = p->A;
    }
    if (w8_layout) {
        const size_t parts = (size_t)p->batch * p->n_chunks * p->dim * (p->dstate + 2);
        return mmu_scan_bwd_reduce_w8(p->workspace + xs, p->workspace + xs + parts + (have_x ? 0 : xs), p->batch, p->dim,
                                      p->seqlen, p->dA, p->dD, p->ddelta_bias, Asc, st);
    }
    {
        const int BC = p->batch * p->n_chunks;
        const int n_slices = (BC + 511) / 512;
        const size_t parts = (size_t)BC * p->dim * (p->dstate + 2);
        // inside a deferred scope: with the other parameter-gradient sums of the pass (deferred_reduce.hip, kind 5)
        const long job[8] = {5, (long)(p->workspace + xs), (long)p->dA, (long)p->dD, (long)p->ddelta_bias, BC,
                             (long)p->dim | ((long)p->dstate << 32), (long)Asc};
        if (mmu_defer_job(job)) return 0;
        float *part2 = n_slices > 1 ? p->workspace + xs + parts + (have_x ? 0 : xs) : nullptr;
        dim3 g5(p->dim, n_slices);
        reduce_partials_kernel<<<g5, 256, 0, st>>>(p->workspace + xs, BC, p->dim, p->dstate, p->dA, p->dD,
                                                   p->ddelta_bias, part2, Asc);
        if (part2)
            reduce_slices_kernel<<<p->dim, 64, 0, st>>>(part2, n_slices, p->dim, p->dstate, p->dA, p->dD,
                                                        p->ddelta_bias, Asc);
    }
    MMU_HIP_LAUNCH_CHECK("reduce_partials");
    return 0;
}

extern "C" int mmu_debug_wave_scan(const float *P, const float *S, float *outP, float *outS, int n_waves, int reverse,
                                   int variant, void *stream) {
    MMU_CHECK(P && S && outP && outS && n_waves > 0, "debug_wave_scan: bad arguments");
    debug_wave_scan_kernel<<<n_waves, 64, 0, (hipStream_t)stream>>>(P, S, outP, outS, reverse, variant);
    MMU_HIP_LAUNCH_CHECK("debug_wave_scan");
    return 0;
}
