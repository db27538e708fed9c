// selective_scan_stream.hip -- streaming selective-scan forward for gfx950 (MI355X).
//
// Same math as the reference kernel (requirements/Mamba/mamba/csrc/selective_scan/selective_scan_fwd_kernel.cuh:147-298)
//   dl = softplus(delta + bias) ;  a = exp2(dl * A * log2e) ;  h_t = a h_{t-1} + dl u B ;  y = sum_n C h + D u ;
//   out_z = y * silu(z)
// and the same walk as the reference -- every (batch, channel) row is scanned front to back with the state
// carried in registers -- but cut for this chip: the chunk-parallel kernels of selective_scan.hip pay for their
// parallelism over L with a second exp per element (chunk aggregates, K1) and the chunk-record traffic (K2);
// when batch * dim alone fills the chip (>= 512 rows; the three large Mamba blocks of MM-UNet have 1,024) that
// price buys nothing.  Here:
//
//   workgroup = 8 waves = 4 channels x 2 state-halves of ONE batch item; it walks the sequence in 512-token tiles.
//   wave (c, hf) owns channel c and the state pairs 4hf .. 4hf+3 for the whole sequence: lane l holds tokens
//   8l .. 8l+7 of the tile, the 8-token recurrence of a state pair runs in registers on packed math, lanes are
//   joined by the DPP affine scan, and the state that leaves lane 63 enters lane 0 of the next tile through
//   two SGPRs per pair (v_readlane).  No chunk aggregates, no carry kernel, one exp per (d, n, t).
//   The B / C tile is staged in LDS once per tile for the 4 channels (pair-interleaved, conflict-free);
//   what is per (channel, token) -- softplus, delta * u, the D u term, the z gate, the loads and stores --
//   is split between the two waves of a channel by TOKEN (wave hf owns tokens 256hf .. 256hf+255 of the
//   tile, 4 per lane, fully coalesced 1-KiB rows) and meets the per-state work through LDS:
//       phase C(k-1): softplus of tile k -> sDL / sDU ; B / C of tile k -> sBC        | barrier 1
//       phase B(k)  : (loads of tile k+1 issued) 4 state pairs over the tile, partial y -> sY  | barrier 2
//       phase C(k)  : y = both halves + D u, gate, store ; prepare tile k+1 from the loads
//   Loads run a phase B ahead of their use and the barriers order LDS only (no vmcnt drain).
//
// Chunk states: x[b, d, c, 2n+1] = h_n at the end of 128-token chunk c (what the backward kernels and
// last_state read); x[b, d, c, 2n] = decay product from the start of the chunk's 512-token tile to the end
// of the chunk (nothing reads it).
#include <algorithm>
#include <stdlib.h>
#include "scan_common.h"
#include "../../include/mmunet_amd.h"

// Diagnostic build only (-DMMU_STREAM_STAMPS, tools/stream_stamps.sh): s_memtime at the phase boundaries of a few
// tiles, read back through mmu_debug_stream_stamps.  In the product build no stamp executes.
#ifdef MMU_STREAM_STAMPS
__device__ unsigned long long g_stream_stamps[2 * 8 * 8 * 16];  // [block sel][wave][tile 16..23][slot]
#define ST_STAMP(slot)                                                                                     \
    do {                                                                                                   \
        if ((blockIdx.x == 0 || blockIdx.x == 131) && k >= 16 && k < 24 && lane == 0)                      \
            g_stream_stamps[(((blockIdx.x != 0) * 8 + w) * 8 + (k - 16)) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define ST_STAMP(slot)
#endif

namespace {

constexpr int ST_TT = 512;                   // tokens per tile
constexpr int ST_CH = 4;                     // channels per workgroup
constexpr int ST_BC4 = 2 * 8 * 256;          // float4 per buffer: [B|C][8 pairs][4][64]
constexpr int ST_DL4 = ST_CH * 2 * 2 * 64;   // float4: [channel][DL|DU][2][64]
constexpr int ST_Y4 = ST_CH * 2 * 2 * 64;    // float4: [channel][half][2][64]
constexpr size_t ST_LDS = sizeof(float4) * (2 * ST_BC4 + ST_DL4 + ST_Y4);  // 160 KiB: the whole LDS of a CU
static_assert(ST_LDS == 160 * 1024, "the double-buffered B / C tile + exchange rows fill the LDS exactly");

// One state pair over this lane's 8 tokens (see fwd_pair8 in selective_scan.hip): exps and the lane's local
// composition, the cross-lane scan with the carry folded into lane 0, then the true recurrence and y.
// hc: state of the pair entering the tile; leaves holding the state at the end of the tile.
#ifdef MMU_STREAM_STAMPS
#define ST_T(i) do { asm volatile("" ::: "memory"); ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ST_T(i)
#endif
__device__ __forceinline__ void stream_pair(const v2f (&dl)[4], const v2f (&du)[4], float dlsum, v2f (&yp)[8],
                                            const v2f a2, v2f &hc, const float *tileB, const float *tileC, int pr,
                                            int lane, float4 &rec, unsigned long long *ts = nullptr) {
    v2f a[8], bb[8], Bv[8], Cv[8];
    ST_T(0);
    pair8_read(tileB, pr, lane, Bv);  // lands behind the 16 exps
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a[2 * q] = exp2_2(mul_bcast<0>(dl[q], a2));
        a[2 * q + 1] = exp2_2(mul_bcast<1>(dl[q], a2));
    }
    const v2f P = exp2_2(a2 * dlsum);
    __builtin_amdgcn_sched_barrier(0);
    ST_T(1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bb[2 * q] = mul_bcast<0>(du[q], Bv[2 * q]);
        bb[2 * q + 1] = mul_bcast<1>(du[q], Bv[2 * q + 1]);
    }
    v2f S = bb[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) S = fma2(a[i], S, bb[i]);
    const v2f S_in = fma2(P, hc, S);
    S = lane == 0 ? S_in : S;
    float P0 = P.x, S0 = S.x, P1 = P.y, S1 = S.y;
    __builtin_amdgcn_sched_barrier(0);
    ST_T(2);
    pair8_read(tileC, pr, lane, Cv);  // lands during the scan
    wave_scan_affine_x2(P0, S0, P1, S1);
    v2f h = v2f{wave_shift_up1(S0, hc.x), wave_shift_up1(S1, hc.y)};  // states entering this lane's tokens
#ifdef MMU_STREAM_STAMPS
    asm volatile("" : "+v"(h));
#endif
    ST_T(3);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h = fma2(a[i], h, bb[i]);
        yp[i] = fma2(Cv[i], h, yp[i]);
    }
#ifdef MMU_STREAM_STAMPS
    asm volatile("" : "+v"(yp[7]), "+v"(h));
#endif
    ST_T(4);
    rec = make_float4(P0, S0, P1, S1);
    hc = v2f{wave_bcast_last(S0), wave_bcast_last(S1)};
    asm volatile("" : "+v"(hc));  // carried in VGPRs (see a2)
}

template <typename io_t, bool HAS_Z, bool SOFTPLUS, bool HAS_OUT>
__global__ __launch_bounds__(512, 2) void scan_fwd_stream_kernel(ScanArgs p) {
    constexpr unsigned ES = sizeof(io_t);
    constexpr int N = 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *sBC = reinterpret_cast<float4 *>(smem);  // [2 buffers][B|C][8 pairs][4][64]
    float4 *sDL = sBC + 2 * ST_BC4;                   // [channel][DL|DU][2][64]
    float4 *sY = sDL + ST_DL4;                        // [channel][half][2][64]
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = w & 3, hf = w >> 2;
    // batch fastest: the workgroups of one batch item land on one XCD (round-robin dispatch) and share its L2
    // for the B / C tiles they all read.  Speed only.
    const int b = blockIdx.x % p.batch, dg = blockIdx.x / p.batch;
    const int d = dg * ST_CH + c;
    const int g = (dg * ST_CH) / (p.dim / p.ngroups);
    const int nt = p.seqlen / ST_TT;

    const rsrc_t r_delta = make_rsrc((const io_t *)p.delta + (long)b * p.delta_bs + (long)d * p.delta_ds);
    const rsrc_t r_u = make_rsrc((const io_t *)p.u + (long)b * p.u_bs + (long)d * p.u_ds);
    const rsrc_t r_z = make_rsrc(HAS_Z ? (const io_t *)p.z + (long)b * p.z_bs + (long)d * p.z_ds : (const io_t *)p.u);
    const rsrc_t r_oz = make_rsrc(HAS_Z ? (io_t *)p.out_z + (long)b * p.out_z_bs + (long)d * p.out_z_ds : (io_t *)p.out);
    // No control flow inside the tile loop (a branch splits the block, the scheduling fences stop holding and
    // the pair bodies smear into each other: 256 VGPRs + scratch): `out` is a template flag, and the lanes that
    // own no chunk record carry an offset beyond the record resource, whose range check drops their store.
    const rsrc_t r_out = make_rsrc(HAS_OUT ? (io_t *)p.out + (long)b * p.out_bs + (long)d * p.out_ds : (io_t *)p.u);
    // this wave stages state pair w of the B tile and of the C tile (rows 2w, 2w+1)
    const rsrc_t r_B = make_rsrc((const io_t *)p.B + (long)b * p.B_bs + (long)g * p.B_gs + (long)(2 * w) * p.B_ns);
    const rsrc_t r_C = make_rsrc((const io_t *)p.C + (long)b * p.C_bs + (long)g * p.C_gs + (long)(2 * w) * p.C_ns);
    const rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + ((long)b * p.dim + d) * p.n_chunks * 2 * N), 0,
                                                         p.n_chunks * 2 * N * 4, 0x00020000);
    const unsigned voff_io = (256 * hf + 4 * lane) * ES;  // this wave's 4 tokens per lane of a tile
    const unsigned voff_bc = 8 * lane * ES;
    const unsigned row1_B = (unsigned)p.B_ns * ES, row1_C = (unsigned)p.C_ns * ES;

    // loop constants of this wave: A * log2e of its 8 states, bias, D
    v2f a2[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        const float *Ap = p.A + (long)d * p.A_ds + (long)(8 * hf + 2 * pr) * p.A_ns;
        a2[pr] = v2f{Ap[0] * MMU_LOG2E, Ap[p.A_ns] * MMU_LOG2E};
        asm volatile("" : "+v"(a2[pr]));  // loop constants live in VGPRs: the SGPR file is full of buffer resources
    }
    const float bias = p.delta_bias ? p.delta_bias[d] : 0.f;
    const float Dv = p.D ? p.D[d] : 0.f;

    // where this lane's 4 I/O tokens live in the [2 quarters][64 lanes] float4 rows of sDL / sDU / sY:
    // tokens 256hf + 4lane .. +3 of the tile = state-lane 32hf + lane/2, quarter lane&1
    const int io_q = (lane & 1) * 64 + 32 * hf + (lane >> 1);
    float4 *sDLc = sDL + c * 256, *sDUc = sDLc + 128;
    float4 *sYc = sY + c * 256;

    // staging registers of the B / C rows: the raw 16-byte tuples as loaded (NQ per row: 2 for float, 1 for bf16).
    // They are carried over the loop's back-edge while the loads are in flight, so nothing may touch them in
    // between: kept as whole tuples and pinned (empty asm) at the point of use, otherwise the compiler places
    // the copies of the pair interleave right behind the loads -- one iteration early, a vmcnt(0) wait.
    constexpr int NQ = ES == 4 ? 2 : 1;
    float dl_n[4], u_n[4], z_n[4], u_cur[4];
    v4u rB[2 * NQ], rC[2 * NQ];
    auto tile_off = [&](int kt) { return (unsigned)__builtin_amdgcn_readfirstlane(kt) * (unsigned)(ST_TT * ES); };
    auto fetch_rows = [&](rsrc_t r, unsigned row1, int kt, v4u (&q)[2 * NQ]) {
        const unsigned so = tile_off(kt);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            q[j] = __builtin_amdgcn_raw_buffer_load_b128(r, voff_bc + 16 * j, so, 0);
            q[NQ + j] = __builtin_amdgcn_raw_buffer_load_b128(r, voff_bc + 16 * j, so + row1, 0);
        }
    };
    auto fetch_B = [&](int kt) { fetch_rows(r_B, row1_B, kt, rB); };
    auto fetch_C = [&](int kt) { fetch_rows(r_C, row1_C, kt, rC); };
    auto fetch_io = [&](int kt) {
        const unsigned so = tile_off(kt);
        buf_load4<io_t>(r_delta, voff_io, so, dl_n);
        buf_load4<io_t>(r_u, voff_io, so, u_n);
    };
    auto fetch_z = [&](int kt) {
        if constexpr (HAS_Z) buf_load4<io_t>(r_z, voff_io, tile_off(kt), z_n);
    };
    // pair-interleaved image of this wave's two rows (see stage_pair8 in selective_scan.hip)
    auto put_rows = [&](v4u (&q)[2 * NQ], float4 *dst) {
#pragma unroll
        for (int j = 0; j < 2 * NQ; ++j) asm volatile("" : "+v"(q[j]));
        float r0[8], r1[8];
        if constexpr (ES == 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                r0[4 * j] = __uint_as_float(q[j].x); r0[4 * j + 1] = __uint_as_float(q[j].y);
                r0[4 * j + 2] = __uint_as_float(q[j].z); r0[4 * j + 3] = __uint_as_float(q[j].w);
                r1[4 * j] = __uint_as_float(q[2 + j].x); r1[4 * j + 1] = __uint_as_float(q[2 + j].y);
                r1[4 * j + 2] = __uint_as_float(q[2 + j].z); r1[4 * j + 3] = __uint_as_float(q[2 + j].w);
            }
        } else {
            const unsigned w0[4] = {q[0].x, q[0].y, q[0].z, q[0].w}, w1[4] = {q[1].x, q[1].y, q[1].z, q[1].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r0[2 * i] = __uint_as_float(w0[i] << 16); r0[2 * i + 1] = __uint_as_float(w0[i] & 0xffff0000u);
                r1[2 * i] = __uint_as_float(w1[i] << 16); r1[2 * i + 1] = __uint_as_float(w1[i] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i * 64] = make_float4(r0[2 * i], r1[2 * i], r0[2 * i + 1], r1[2 * i + 1]);
    };
    auto put_B = [&](int buf) { put_rows(rB, sBC + buf * ST_BC4 + w * 256 + lane); };
    auto put_C = [&](int buf) { put_rows(rC, sBC + buf * ST_BC4 + (8 + w) * 256 + lane); };
    auto prepare = [&]() {  // softplus and delta * u of this wave's tokens of the next tile
        float dl[4], du[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = dl_n[i] + bias;
            if constexpr (SOFTPLUS) v = softplus_thr(v);
            dl[i] = v;
            du[i] = v * u_n[i];
            u_cur[i] = u_n[i];
        }
        sDLc[io_q] = make_float4(dl[0], dl[1], dl[2], dl[3]);
        sDUc[io_q] = make_float4(du[0], du[1], du[2], du[3]);
    };

    // tile 0 -> LDS; the B / C rows of tile 1 -> staging registers
    fetch_io(0);
    fetch_B(0);
    fetch_C(0);
    prepare();
    put_B(0);
    put_C(0);
    fetch_B(nt > 1 ? 1 : 0);
    fetch_C(nt > 1 ? 1 : 0);

    v2f hc[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) hc[pr] = v2f{0.f, 0.f};
    // chunk records: lanes 15, 31, 47, 63 hold the states at the ends of the tile's four 128-token chunks
    unsigned voff_x = (lane & 15) == 15 ? ((lane >> 4) * 2 * N + 16 * hf) * 4u : 0x80000000u;

    for (int k = 0; k < nt; ++k) {
        const int cur = k & 1;
        const float *tileB = reinterpret_cast<const float *>(sBC + cur * ST_BC4);
        const float *tileC = reinterpret_cast<const float *>(sBC + cur * ST_BC4 + 8 * 256);
        // The per-token streams are loaded and consumed inside ONE iteration (phase B lies between): a load
        // result carried over the back-edge AND copied (u_cur) gets its copy at the loop top, i.e. a vmcnt wait
        // for loads that were just issued.
        ST_STAMP(0);
        fetch_io(k + 1 < nt ? k + 1 : k);  // (the last iteration re-reads its own tile: L2 hits)
        fetch_z(k);
        ST_STAMP(1);
        MMU_LDS_BARRIER();  // tile k's B / C (written during phase B of tile k-1), dl, dl*u are in LDS
        ST_STAMP(2);
        // ---- phase B: this wave's 4 state pairs over the tile; between them, in the issue slots the VALU
        // leaves free, the staging of the NEXT tiles: registers (tile k+1) -> the other LDS buffer, then the
        // loads of tile k+2 into the same registers.  Done as a block before / after the pairs, all eight waves
        // queue on the LDS store path (~13 cycles per ds_write_b128) and the texture path (16 per 1-KiB load)
        // at the same time with the VALU idle: 3,400 of 10,900 cycles per tile (tools/stream_stamps.py).
        v2f dl2[4], du2[4], yp[8];
        float dlsum;
        {
            const float4 d0 = sDLc[lane], d1 = sDLc[64 + lane];
            const float4 e0 = sDUc[lane], e1 = sDUc[64 + lane];
            dl2[0] = v2f{d0.x, d0.y}; dl2[1] = v2f{d0.z, d0.w}; dl2[2] = v2f{d1.x, d1.y}; dl2[3] = v2f{d1.z, d1.w};
            du2[0] = v2f{e0.x, e0.y}; du2[1] = v2f{e0.z, e0.w}; du2[2] = v2f{e1.x, e1.y}; du2[3] = v2f{e1.z, e1.w};
            dlsum = ((d0.x + d0.y) + (d0.z + d0.w)) + ((d1.x + d1.y) + (d1.z + d1.w));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) yp[i] = v2f{0.f, 0.f};
#ifdef MMU_STREAM_STAMPS
        asm volatile("" : "+v"(dlsum));
#endif
        ST_STAMP(3);
        const int kn = k + 2 < nt ? k + 2 : nt - 1;
        // The two waves of a SIMD (c, 0) and (c, 1) run the same program; VALU issue goes to the older one first,
        // which then runs its four pairs at the single-wave rate while the younger gets the leftover slots and
        // ends up finishing two pairs alone (stamps: 750 vs 1,500 cycles per pair, 5,000 instead of ~4,000 for
        // the phase).  The younger half leads for the first two pairs instead.
        if (hf) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            float4 rec;
#ifdef MMU_STREAM_STAMPS
            unsigned long long ts[5];
            stream_pair(dl2, du2, dlsum, yp, a2[pr], hc[pr], tileB, tileC, 4 * hf + pr, lane, rec, ts);
            if (pr == 2 && (blockIdx.x == 0 || blockIdx.x == 131) && k >= 16 && k < 24 && lane == 0)
                for (int i = 0; i < 4; ++i)
                    g_stream_stamps[(((blockIdx.x != 0) * 8 + w) * 8 + (k - 16)) * 16 + 12 + i] = ts[i + 1] - ts[i];
#else
            stream_pair(dl2, du2, dlsum, yp, a2[pr], hc[pr], tileB, tileC, 4 * hf + pr, lane, rec);
#endif
            __builtin_amdgcn_raw_buffer_store_b128(
                v4u{__float_as_uint(rec.x), __float_as_uint(rec.y), __float_as_uint(rec.z), __float_as_uint(rec.w)},
                r_x, voff_x + 16u * pr, 0, 0);
            if (pr == 0) {
                put_B(cur ^ 1);
                fetch_B(kn);  // a whole tile ahead of its use: every CU asks for the same tiles at the same time
            }
            if (pr == 1) {
                put_C(cur ^ 1);
                fetch_C(kn);
                if (hf) __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);  // one pair at a time: interleaving two doubles the live registers
#ifdef MMU_STREAM_STAMPS
            asm volatile("" : "+v"(yp[0]), "+v"(yp[7]));
#endif
            ST_STAMP(4 + pr);
        }
        voff_x += 4 * 2 * N * 4;
        sYc[hf * 128 + lane] = make_float4(yp[0].x + yp[0].y, yp[1].x + yp[1].y, yp[2].x + yp[2].y, yp[3].x + yp[3].y);
        sYc[hf * 128 + 64 + lane] = make_float4(yp[4].x + yp[4].y, yp[5].x + yp[5].y, yp[6].x + yp[6].y, yp[7].x + yp[7].y);
        ST_STAMP(8);
        MMU_LDS_BARRIER();  // both halves' partial y are in LDS; everyone is done with tile k's dl, dl*u
        ST_STAMP(9);
        // ---- phase C: finish tile k on this wave's tokens, prepare tile k+1 ----
        {
            const float4 ya = sYc[io_q], yb = sYc[128 + io_q];
            float y[4] = {ya.x + yb.x, ya.y + yb.y, ya.z + yb.z, ya.w + yb.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = fmaf(Dv, u_cur[i], y[i]);
            const unsigned so = tile_off(k);
            if constexpr (HAS_OUT) buf_store4<io_t>(r_out, voff_io, so, y);
            if constexpr (HAS_Z) {
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] *= z_n[i] * sigmoidf_(z_n[i]);
                buf_store4<io_t>(r_oz, voff_io, so, y);
            }
        }
        ST_STAMP(10);
        prepare();  // tile k+1: consumes the loads issued at the top of this iteration
        ST_STAMP(11);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Version 2 (round 4): ONE barrier per tile.
//
// Stamps of the kernel above (profiles/r02_scan_fwd_stream_phase_stamps.txt): of a tile's 8,300 cycles the four
// state pairs take 5,000-5,400, and their instruction stream accounts for ~4,100 of those; the other 3,000 are the
// two barriers and the exchange phases around them (dl / dl*u through LDS, partial y through LDS, finalize,
// prepare) in which all eight waves wait on the same LDS / memory latencies with the VALU idle.  Here nothing but
// the B / C tile and HALF of the partial y goes through LDS, and every latency sits under pair work:
//   * each wave loads delta and u of ALL 8 tokens of its lanes itself (two 16-byte loads per stream) and computes
//     softplus and delta * u redundantly with its partner wave (same channel, other state half): 8 softplus per
//     lane and tile instead of 4, no exchange rows, no barrier between the loads and their use;
//   * wave (c, hf) finishes tokens 8l + 4hf .. + 3 of its lanes: it keeps its own partial y of those in registers and
//     gets the partner's through LDS (one ds_write_b128 / ds_read_b128 per lane and tile, double-buffered by tile
//     parity); z is loaded and out_z stored for those 4 tokens only (16 bytes per lane at a 32-byte stride: the
//     partner fills the other half of every line);
//   * the finishing of tile k - 1 (partner's y, D u, gate, store) runs between pairs 2 and 3 of tile k, the
//     preparation of tile k + 1 after pair 3, the B / C staging between pairs 0 / 1 / 2 as before;
//   * one LDS-only barrier at the top of a tile orders: B / C of tile k (staged during tile k - 1), the partner's y of
//     tile k - 1 (written at the end of tile k - 1).
// LDS: 2 x 64 KiB (B / C) + 2 x 8 KiB (y halves) = 144 KiB.
// ---------------------------------------------------------------------------------------------------------------
constexpr int S2_Y4 = 8 * 64;  // float4 per buffer: [wave][lane]
constexpr size_t S2_LDS = sizeof(float4) * (2 * ST_BC4 + 2 * S2_Y4);

template <typename io_t, bool HAS_Z, bool SOFTPLUS, bool HAS_OUT>
__global__ __launch_bounds__(512, 2) void scan_fwd_stream2_kernel(ScanArgs p) {
    constexpr unsigned ES = sizeof(io_t);
    constexpr int N = 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *sBC = reinterpret_cast<float4 *>(smem);  // [2 buffers][B|C][8 pairs][4][64]
    float4 *sY = sBC + 2 * ST_BC4;                    // [2 buffers][wave][lane]
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = w & 3, hf = w >> 2;
    const int b = blockIdx.x % p.batch, dg = blockIdx.x / p.batch;
    const int d = dg * ST_CH + c;
    const int g = (dg * ST_CH) / (p.dim / p.ngroups);
    const int nt = p.seqlen / ST_TT;
    const unsigned row_bytes = (unsigned)p.seqlen * ES;

    const rsrc_t r_delta = make_rsrc((const io_t *)p.delta + (long)b * p.delta_bs + (long)d * p.delta_ds);
    const rsrc_t r_u = make_rsrc((const io_t *)p.u + (long)b * p.u_bs + (long)d * p.u_ds);
    const rsrc_t r_z = make_rsrc(HAS_Z ? (const io_t *)p.z + (long)b * p.z_bs + (long)d * p.z_ds : (const io_t *)p.u);
    // the output rows carry their true length: the lane offset 0x80000000 of "nothing to finish yet" (tile -1) fails
    // the range check and the store is dropped
    const rsrc_t r_oz = __builtin_amdgcn_make_buffer_rsrc(
        HAS_Z ? (void *)((io_t *)p.out_z + (long)b * p.out_z_bs + (long)d * p.out_z_ds) : (void *)p.u, 0, row_bytes, 0x00020000);
    const rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
        HAS_OUT ? (void *)((io_t *)p.out + (long)b * p.out_bs + (long)d * p.out_ds) : (void *)p.u, 0, row_bytes, 0x00020000);
    const rsrc_t r_B = make_rsrc((const io_t *)p.B + (long)b * p.B_bs + (long)g * p.B_gs + (long)(2 * w) * p.B_ns);
    const rsrc_t r_C = make_rsrc((const io_t *)p.C + (long)b * p.C_bs + (long)g * p.C_gs + (long)(2 * w) * p.C_ns);
    const rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + ((long)b * p.dim + d) * p.n_chunks * 2 * N), 0,
                                                         p.n_chunks * 2 * N * 4, 0x00020000);
    // hf again, as a per-lane value: the selects on it below must be v_cndmasks -- on the scalar the compiler threads
    // them into the branch around s_setprio and splits the tile loop's block
    const bool hfv = (threadIdx.x & 256) != 0;
    const unsigned voff8 = 8 * lane * ES;             // the lane's 8 tokens of a tile
    const unsigned voff4 = (8 * lane + 4 * hf) * ES;  // the 4 of them this wave finishes
    const unsigned voff_bc = 8 * lane * ES;
    const unsigned row1_B = (unsigned)p.B_ns * ES, row1_C = (unsigned)p.C_ns * ES;

    v2f a2[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        const float *Ap = p.A + (long)d * p.A_ds + (long)(8 * hf + 2 * pr) * p.A_ns;
        a2[pr] = v2f{Ap[0] * MMU_LOG2E, Ap[p.A_ns] * MMU_LOG2E};
        asm volatile("" : "+v"(a2[pr]));
    }
    const float bias = p.delta_bias ? p.delta_bias[d] : 0.f;
    const float Dv = p.D ? p.D[d] : 0.f;

    constexpr int NQ = ES == 4 ? 2 : 1;
    float dl_n[8], u_n[8], z_p[4];
    v4u rB[2 * NQ], rC[2 * NQ];
    auto tile_off = [&](int kt) { return (unsigned)__builtin_amdgcn_readfirstlane(kt) * (unsigned)(ST_TT * ES); };
    auto fetch_rows = [&](rsrc_t r, unsigned row1, int kt, v4u (&q)[2 * NQ]) {
        const unsigned so = tile_off(kt);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            q[j] = __builtin_amdgcn_raw_buffer_load_b128(r, voff_bc + 16 * j, so, 0);
            q[NQ + j] = __builtin_amdgcn_raw_buffer_load_b128(r, voff_bc + 16 * j, so + row1, 0);
        }
    };
    auto fetch_B = [&](int kt) { fetch_rows(r_B, row1_B, kt, rB); };
    auto fetch_C = [&](int kt) { fetch_rows(r_C, row1_C, kt, rC); };
    auto fetch_io = [&](int kt) {
        const unsigned so = tile_off(kt);
        buf_load8<io_t>(r_delta, voff8, so, dl_n);
        buf_load8<io_t>(r_u, voff8, so, u_n);
    };
    auto fetch_z = [&](int kt) {
        if constexpr (HAS_Z) buf_load4<io_t>(r_z, voff4, tile_off(kt), z_p);
    };
    auto put_rows = [&](v4u (&q)[2 * NQ], float4 *dst) {
#pragma unroll
        for (int j = 0; j < 2 * NQ; ++j) asm volatile("" : "+v"(q[j]));
        float r0[8], r1[8];
        if constexpr (ES == 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                r0[4 * j] = __uint_as_float(q[j].x); r0[4 * j + 1] = __uint_as_float(q[j].y);
                r0[4 * j + 2] = __uint_as_float(q[j].z); r0[4 * j + 3] = __uint_as_float(q[j].w);
                r1[4 * j] = __uint_as_float(q[2 + j].x); r1[4 * j + 1] = __uint_as_float(q[2 + j].y);
                r1[4 * j + 2] = __uint_as_float(q[2 + j].z); r1[4 * j + 3] = __uint_as_float(q[2 + j].w);
            }
        } else {
            const unsigned w0[4] = {q[0].x, q[0].y, q[0].z, q[0].w}, w1[4] = {q[1].x, q[1].y, q[1].z, q[1].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r0[2 * i] = __uint_as_float(w0[i] << 16); r0[2 * i + 1] = __uint_as_float(w0[i] & 0xffff0000u);
                r1[2 * i] = __uint_as_float(w1[i] << 16); r1[2 * i + 1] = __uint_as_float(w1[i] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i * 64] = make_float4(r0[2 * i], r1[2 * i], r0[2 * i + 1], r1[2 * i + 1]);
    };
    auto put_B = [&](int buf) { put_rows(rB, sBC + buf * ST_BC4 + w * 256 + lane); };
    auto put_C = [&](int buf) { put_rows(rC, sBC + buf * ST_BC4 + (8 + w) * 256 + lane); };

    // the tile in work: softplus(delta + bias), its product with u (all 8 tokens of the lane), D * u of the 4 tokens
    // this wave finishes
    v2f dl2[4], du2[4];
    float dlsum, Du_own[4];
    auto prepare = [&]() {
        float dl[8], du[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = dl_n[i] + bias;
            if constexpr (SOFTPLUS) v = softplus_thr(v);
            dl[i] = v;
            du[i] = v * u_n[i];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            dl2[q] = v2f{dl[2 * q], dl[2 * q + 1]};
            du2[q] = v2f{du[2 * q], du[2 * q + 1]};
        }
        dlsum = ((dl[0] + dl[1]) + (dl[2] + dl[3])) + ((dl[4] + dl[5]) + (dl[6] + dl[7]));
#pragma unroll
        for (int i = 0; i < 4; ++i) Du_own[i] = Dv * (hfv ? u_n[4 + i] : u_n[i]);
    };
    // y of the wave's 4 tokens of tile kt: own partial (+ D u) in y_own, the partner's in yb; z of those tokens in z_p
    float y_own[4] = {0.f, 0.f, 0.f, 0.f};
    auto finish = [&](int kt, const float4 yb, unsigned voff) {
        float y[4] = {y_own[0] + yb.x, y_own[1] + yb.y, y_own[2] + yb.z, y_own[3] + yb.w};
        const unsigned so = tile_off(kt);
        if constexpr (HAS_OUT) buf_store4<io_t>(r_out, voff, so, y);
        if constexpr (HAS_Z) {
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] *= z_p[i] * sigmoidf_(z_p[i]);
            buf_store4<io_t>(r_oz, voff, so, y);
        }
    };

    fetch_io(0);
    fetch_B(0);
    fetch_C(0);
    prepare();
    put_B(0);
    put_C(0);
    fetch_B(nt > 1 ? 1 : 0);
    fetch_C(nt > 1 ? 1 : 0);

    v2f hc[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) hc[pr] = v2f{0.f, 0.f};
    unsigned voff_x = (lane & 15) == 15 ? ((lane >> 4) * 2 * N + 16 * hf) * 4u : 0x80000000u;
    const float4 *sYin = sY + (w ^ 4) * 64 + lane;   // the partner's row
    float4 *sYout = sY + w * 64 + lane;

    for (int k = 0; k < nt; ++k) {
        const int cur = k & 1;
        const float *tileB = reinterpret_cast<const float *>(sBC + cur * ST_BC4);
        const float *tileC = reinterpret_cast<const float *>(sBC + cur * ST_BC4 + 8 * 256);
        fetch_io(k + 1 < nt ? k + 1 : k);   // consumed by prepare() at the end of this iteration
        fetch_z(k > 0 ? k - 1 : 0);         // consumed by finish() between pairs 2 and 3
        MMU_LDS_BARRIER();
        v2f yp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) yp[i] = v2f{0.f, 0.f};
        const int kn = k + 2 < nt ? k + 2 : nt - 1;
        const unsigned voff_fin = k > 0 ? voff4 : 0x80000000u;
        float4 yb;
        if (hf) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            float4 rec;
            stream_pair(dl2, du2, dlsum, yp, a2[pr], hc[pr], tileB, tileC, 4 * hf + pr, lane, rec);
            __builtin_amdgcn_raw_buffer_store_b128(
                v4u{__float_as_uint(rec.x), __float_as_uint(rec.y), __float_as_uint(rec.z), __float_as_uint(rec.w)},
                r_x, voff_x + 16u * pr, 0, 0);
            if (pr == 0) {
                put_B(cur ^ 1);
                fetch_B(kn);
            }
            if (pr == 1) {
                put_C(cur ^ 1);
                fetch_C(kn);
                if (hf) __builtin_amdgcn_s_setprio(0);
                yb = sYin[(cur ^ 1) * S2_Y4];   // lands during pair 2
            }
            if (pr == 2) finish(k - 1, yb, voff_fin);
            __builtin_amdgcn_sched_barrier(0);
        }
        voff_x += 4 * 2 * N * 4;
        {
            float ys[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ys[i] = yp[i].x + yp[i].y;
            float oth[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                y_own[i] = (hfv ? ys[4 + i] : ys[i]) + Du_own[i];
                oth[i] = hfv ? ys[i] : ys[4 + i];
            }
            sYout[cur * S2_Y4] = make_float4(oth[0], oth[1], oth[2], oth[3]);
        }
        prepare();   // tile k + 1 (the last iteration prepares its own tile again: nothing reads it)
    }
    fetch_z(nt - 1);
    MMU_LDS_BARRIER();
    finish(nt - 1, sYin[((nt - 1) & 1) * S2_Y4], voff4);
}


// ---------------------------------------------------------------------------------------------------------------
// Version 3 (round 4): SIXTEEN tokens per lane, two state pairs side by side in the half-waves.
//
// The cross-lane scan costs the same ~26 DPP instructions (+ selects, carries, record copies) per state pair and
// tile however many tokens a lane owns, and at 8 tokens per lane that is a quarter of a pair's instructions.  A
// 1,024-token tile does not fit (B / C alone would be 128 KiB single-buffered), so the tile stays 512 tokens and the
// WAVE is cut instead: lanes 0-31 run one state pair, lanes 32-63 the next, each lane over 16 consecutive tokens.
// One pass = two pairs: the scan is five DPP steps (row_shr 1/2/4/8, row_bcast:15 into rows 1 and 3 -- no step
// crosses lane 31 | 32), paid once per 2 x 16 tokens; a wave runs two passes per tile instead of four pairs.
// The partial y of the two half-waves (same tokens, different states) meet through v_permlane32_swap: swapping
// register m with register m + 8 and adding leaves tokens 0-7 of the lane's 16 in the lower half-wave and
// tokens 8-15 in the upper one, which is also the split the exchange with the partner wave wants.
// Everything else is version 1: who prepares / finishes which token (256 hf + 4 lane .. + 3: coalesced 1-KiB
// rows), two LDS-only barriers per tile, loads a phase ahead, staging of the next B / C tile between the passes.
// LDS images:  B / C tile  [buffer][B|C][pair][r = 0..7][l5 = 0..31] float4 = (row 2p, row 2p+1)[16 l5 + 2r],
//                                                                             (row 2p, row 2p+1)[16 l5 + 2r + 1]
//              dl, dl*u, y [channel][..][(g & 3) * 32 + (g >> 2)] float4 for the 4-token group g of the tile: the
//                          16-token lanes read / write four contiguous rows.
// ---------------------------------------------------------------------------------------------------------------
#define MMU_SCANH_STEP(ctrl, mask)                                                         \
    "v_fmac_f32_dpp %1, %1, %0 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"               \
    "v_fmac_f32_dpp %3, %3, %2 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"               \
    "v_mul_f32_dpp %0, %0, %0 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                \
    "v_mul_f32_dpp %2, %2, %2 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"
// inclusive affine scan inside each 32-lane half of the wave, two states at once (see wave_scan_affine_x2)
__device__ __forceinline__ void half_scan_affine_x2(float &P0, float &S0, float &P1, float &S1) {
    asm volatile("s_nop 1\n\t"
                 MMU_SCANH_STEP("row_shr:1", "0xf")
                 MMU_SCANH_STEP("row_shr:2", "0xf")
                 MMU_SCANH_STEP("row_shr:4", "0xf")
                 MMU_SCANH_STEP("row_shr:8", "0xf")
                 MMU_SCANH_STEP("row_bcast:15", "0xa")
                 "s_nop 1"
                 : "+v"(P0), "+v"(S0), "+v"(P1), "+v"(S1));
}

// One pass: this lane's state pair over its 16 tokens.  tB / tC: the pair's B / C rows of the tile, at this lane
// (stride 32 float4 between the 8 two-token groups).  hc: the pair's state entering the tile (per lane: the two
// half-waves carry different pairs); leaves holding the state at the end of the tile.
template <bool FIRST>
__device__ __forceinline__ void stream_pass16(const v2f (&dl)[8], const v2f (&du)[8], float dlsum, v2f (&yq)[16],
                                              const v2f a2, v2f &hc, const float4 *tB, const float4 *tC, bool l5zero,
                                              bool upper, float4 &rec) {
    v2f a[16], bb[16];
    {
        v2f Bv[16];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float4 f = tB[r * 32];
            Bv[2 * r] = v2f{f.x, f.y};
            Bv[2 * r + 1] = v2f{f.z, f.w};
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            a[2 * q] = exp2_2(mul_bcast<0>(dl[q], a2));
            a[2 * q + 1] = exp2_2(mul_bcast<1>(dl[q], a2));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            bb[2 * q] = mul_bcast<0>(du[q], Bv[2 * q]);
            bb[2 * q + 1] = mul_bcast<1>(du[q], Bv[2 * q + 1]);
        }
    }
    const v2f P = exp2_2(a2 * dlsum);
    v2f S = bb[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) S = fma2(a[i], S, bb[i]);
    const v2f S_in = fma2(P, hc, S);
    S = l5zero ? S_in : S;
    float P0 = P.x, S0 = S.x, P1 = P.y, S1 = S.y;
    __builtin_amdgcn_sched_barrier(0);
    v2f Cv[16];
#pragma unroll
    for (int r = 0; r < 8; ++r) {   // lands during the scan
        const float4 f = tC[r * 32];
        Cv[2 * r] = v2f{f.x, f.y};
        Cv[2 * r + 1] = v2f{f.z, f.w};
    }
    half_scan_affine_x2(P0, S0, P1, S1);
    // state entering this lane's tokens: the previous lane's inclusive S; the first lane of a half-wave takes the carry
    const float sh0 = dpp_mov<MMU_DPP_WAVE_SHR1, 0xf>(0.f, S0), sh1 = dpp_mov<MMU_DPP_WAVE_SHR1, 0xf>(0.f, S1);
    v2f h = v2f{l5zero ? hc.x : sh0, l5zero ? hc.y : sh1};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        h = fma2(a[i], h, bb[i]);
        if constexpr (FIRST)
            yq[i] = Cv[i] * h;
        else
            yq[i] = fma2(Cv[i], h, yq[i]);
    }
    rec = make_float4(P0, S0, P1, S1);
    const float e0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, S0), 31));
    const float e1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, S1), 31));
    const float f0 = wave_bcast_last(S0), f1 = wave_bcast_last(S1);
    hc = v2f{upper ? f0 : e0, upper ? f1 : e1};
}

template <typename io_t, bool HAS_Z, bool SOFTPLUS, bool HAS_OUT>
__global__ __launch_bounds__(512, 2) void scan_fwd_stream16_kernel(ScanArgs p) {
    constexpr unsigned ES = sizeof(io_t);
    constexpr int N = 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *sBC = reinterpret_cast<float4 *>(smem);  // [2 buffers][B|C][8 pairs][8][32]
    float4 *sDL = sBC + 2 * ST_BC4;                   // [channel][DL|DU][128]
    float4 *sY = sDL + ST_DL4;                        // [channel][half][128]
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = w & 3, hf = w >> 2;
    const int l5 = lane & 31, hw = lane >> 5;
    const bool l5zero = l5 == 0, upper = hw != 0;
    const int b = blockIdx.x % p.batch, dg = blockIdx.x / p.batch;
    const int d = dg * ST_CH + c;
    const int g = (dg * ST_CH) / (p.dim / p.ngroups);
    const int nt = p.seqlen / ST_TT;

    const rsrc_t r_delta = make_rsrc((const io_t *)p.delta + (long)b * p.delta_bs + (long)d * p.delta_ds);
    const rsrc_t r_u = make_rsrc((const io_t *)p.u + (long)b * p.u_bs + (long)d * p.u_ds);
    const rsrc_t r_z = make_rsrc(HAS_Z ? (const io_t *)p.z + (long)b * p.z_bs + (long)d * p.z_ds : (const io_t *)p.u);
    const rsrc_t r_oz = make_rsrc(HAS_Z ? (io_t *)p.out_z + (long)b * p.out_z_bs + (long)d * p.out_z_ds : (io_t *)p.out);
    const rsrc_t r_out = make_rsrc(HAS_OUT ? (io_t *)p.out + (long)b * p.out_bs + (long)d * p.out_ds : (io_t *)p.u);
    const rsrc_t r_B = make_rsrc((const io_t *)p.B + (long)b * p.B_bs + (long)g * p.B_gs + (long)(2 * w) * p.B_ns);
    const rsrc_t r_C = make_rsrc((const io_t *)p.C + (long)b * p.C_bs + (long)g * p.C_gs + (long)(2 * w) * p.C_ns);
    const rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + ((long)b * p.dim + d) * p.n_chunks * 2 * N), 0,
                                                         p.n_chunks * 2 * N * 4, 0x00020000);
    // Which lane loads / prepares / finishes what is chosen for the LDS banks: consecutive lanes of a 128-bit store
    // must fall into consecutive float4 slots.  I/O: lane L owns the 4-token group 64 hf + 4 (L & 15) + (L >> 4) (a
    // wave still covers one contiguous 1-KiB row per access); staging: lane L loads tokens 16 (L & 31) + 8 (L >> 5) .. + 7
    // of its two rows.
    const int io_g = 64 * hf + 4 * (lane & 15) + (lane >> 4);
    const unsigned voff_io = 4 * io_g * ES;
    const unsigned voff_bc = (16 * (lane & 31) + 8 * (lane >> 5)) * ES;
    const unsigned row1_B = (unsigned)p.B_ns * ES, row1_C = (unsigned)p.C_ns * ES;

    // A * log2e of this LANE's pair in each of the two passes
    v2f a2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float *Ap = p.A + (long)d * p.A_ds + (long)(8 * hf + 4 * j + 2 * hw) * p.A_ns;
        a2[j] = v2f{Ap[0] * MMU_LOG2E, Ap[p.A_ns] * MMU_LOG2E};
    }
    const float bias = p.delta_bias ? p.delta_bias[d] : 0.f;
    const float Dv = p.D ? p.D[d] : 0.f;

    // the lane's I/O group g sits at (g & 3) * 32 + (g >> 2)
    const int io_q = (io_g & 3) * 32 + (io_g >> 2);
    float4 *sDLc = sDL + c * 256, *sDUc = sDLc + 128;
    float4 *sYc = sY + c * 256;
    // groups 4 l5 + 2 hw + {0, 1} of this lane's partial y: rows 2 hw, 2 hw + 1
    float4 *sYw = sYc + hf * 128 + (2 * hw) * 32 + l5;

    constexpr int NQ = ES == 4 ? 2 : 1;
    float dl_n[4], u_n[4], z_n[4], u_cur[4];
    v4u rB[2 * NQ], rC[2 * NQ];
    auto tile_off = [&](int kt) { return (unsigned)__builtin_amdgcn_readfirstlane(kt) * (unsigned)(ST_TT * ES); };
    auto fetch_rows = [&](rsrc_t r, unsigned row1, int kt, v4u (&q)[2 * NQ]) {
        const unsigned so = tile_off(kt);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            q[j] = __builtin_amdgcn_raw_buffer_load_b128(r, voff_bc + 16 * j, so, 0);
            q[NQ + j] = __builtin_amdgcn_raw_buffer_load_b128(r, voff_bc + 16 * j, so + row1, 0);
        }
    };
    auto fetch_B = [&](int kt) { fetch_rows(r_B, row1_B, kt, rB); };
    auto fetch_C = [&](int kt) { fetch_rows(r_C, row1_C, kt, rC); };
    auto fetch_io = [&](int kt) {
        const unsigned so = tile_off(kt);
        buf_load4<io_t>(r_delta, voff_io, so, dl_n);
        buf_load4<io_t>(r_u, voff_io, so, u_n);
    };
    auto fetch_z = [&](int kt) {
        if constexpr (HAS_Z) buf_load4<io_t>(r_z, voff_io, tile_off(kt), z_n);
    };
    // this wave's two rows, tokens 16 (L & 31) + 8 (L >> 5) .. + 7 -> two-token groups r = 4 (L >> 5) + i of 16-token lane L & 31
    auto put_rows = [&](v4u (&q)[2 * NQ], float4 *dst) {
#pragma unroll
        for (int j = 0; j < 2 * NQ; ++j) asm volatile("" : "+v"(q[j]));
        float r0[8], r1[8];
        if constexpr (ES == 4) {
            // (v_pk_mov_b32 pairs instead of the four v_mov per float4 were measured: 480 vs 472 us, round 4)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                r0[4 * j] = __uint_as_float(q[j].x); r0[4 * j + 1] = __uint_as_float(q[j].y);
                r0[4 * j + 2] = __uint_as_float(q[j].z); r0[4 * j + 3] = __uint_as_float(q[j].w);
                r1[4 * j] = __uint_as_float(q[2 + j].x); r1[4 * j + 1] = __uint_as_float(q[2 + j].y);
                r1[4 * j + 2] = __uint_as_float(q[2 + j].z); r1[4 * j + 3] = __uint_as_float(q[2 + j].w);
            }
        } else {
            const unsigned w0[4] = {q[0].x, q[0].y, q[0].z, q[0].w}, w1[4] = {q[1].x, q[1].y, q[1].z, q[1].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r0[2 * i] = __uint_as_float(w0[i] << 16); r0[2 * i + 1] = __uint_as_float(w0[i] & 0xffff0000u);
                r1[2 * i] = __uint_as_float(w1[i] << 16); r1[2 * i + 1] = __uint_as_float(w1[i] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i * 32] = make_float4(r0[2 * i], r1[2 * i], r0[2 * i + 1], r1[2 * i + 1]);
    };
    const int st_lane = (lane >> 5) * 128 + (lane & 31);
    auto put_B = [&](int buf) { put_rows(rB, sBC + buf * ST_BC4 + w * 256 + st_lane); };
    auto put_C = [&](int buf) { put_rows(rC, sBC + buf * ST_BC4 + (8 + w) * 256 + st_lane); };
    auto prepare = [&]() {
        float dl[4], du[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = dl_n[i] + bias;
            if constexpr (SOFTPLUS) v = softplus_thr(v);
            dl[i] = v;
            du[i] = v * u_n[i];
            u_cur[i] = u_n[i];
        }
        sDLc[io_q] = make_float4(dl[0], dl[1], dl[2], dl[3]);
        sDUc[io_q] = make_float4(du[0], du[1], du[2], du[3]);
    };

    fetch_io(0);
    fetch_B(0);
    fetch_C(0);
    prepare();
    put_B(0);
    put_C(0);
    fetch_B(nt > 1 ? 1 : 0);
    fetch_C(nt > 1 ? 1 : 0);

    v2f hc[2];
    hc[0] = hc[1] = v2f{0.f, 0.f};
    // chunk records: lanes 7, 15, 23, 31 of a half-wave hold the states at the ends of the tile's four 128-token chunks
    unsigned voff_x = (l5 & 7) == 7 ? ((l5 >> 3) * 2 * N * 4u + (4 * hf + hw) * 16u) : 0x80000000u;
    const int bc_lane = (4 * hf + hw) * 256 + l5;   // this lane's pair of pass 0 in a B or C tile

    for (int k = 0; k < nt; ++k) {
        const int cur = k & 1;
        const float4 *tileB = sBC + cur * ST_BC4 + bc_lane;
        const float4 *tileC = tileB + 8 * 256;
        fetch_io(k + 1 < nt ? k + 1 : k);
        fetch_z(k);
        MMU_LDS_BARRIER();
        v2f dl2[8], du2[8], yq[16];
        float dlsum;
        {
            float s4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 dd = sDLc[i * 32 + l5], ee = sDUc[i * 32 + l5];
                dl2[2 * i] = v2f{dd.x, dd.y}; dl2[2 * i + 1] = v2f{dd.z, dd.w};
                du2[2 * i] = v2f{ee.x, ee.y}; du2[2 * i + 1] = v2f{ee.z, ee.w};
                s4[i] = (dd.x + dd.y) + (dd.z + dd.w);
            }
            dlsum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
        const int kn = k + 2 < nt ? k + 2 : nt - 1;
#ifndef MMU_S16_PRIO
#define MMU_S16_PRIO 1
#endif
        if (MMU_S16_PRIO == 1 && hf) __builtin_amdgcn_s_setprio(1);
        if (MMU_S16_PRIO == 2 && !hf) __builtin_amdgcn_s_setprio(1);
        {
            float4 rec;
            stream_pass16<true>(dl2, du2, dlsum, yq, a2[0], hc[0], tileB, tileC, l5zero, upper, rec);
            __builtin_amdgcn_raw_buffer_store_b128(
                v4u{__float_as_uint(rec.x), __float_as_uint(rec.y), __float_as_uint(rec.z), __float_as_uint(rec.w)},
                r_x, voff_x, 0, 0);
            put_B(cur ^ 1);
            fetch_B(kn);
            if (MMU_S16_PRIO == 1 && hf) __builtin_amdgcn_s_setprio(0);
            if (MMU_S16_PRIO == 2 && !hf) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            stream_pass16<false>(dl2, du2, dlsum, yq, a2[1], hc[1], tileB + 2 * 256, tileC + 2 * 256, l5zero, upper, rec);
            __builtin_amdgcn_raw_buffer_store_b128(
                v4u{__float_as_uint(rec.x), __float_as_uint(rec.y), __float_as_uint(rec.z), __float_as_uint(rec.w)},
                r_x, voff_x + 32u, 0, 0);
            put_C(cur ^ 1);
            fetch_C(kn);
            __builtin_amdgcn_sched_barrier(0);
        }
        voff_x += 4 * 2 * N * 4;
        {   // sum over the state pair, then over the two half-waves: register m <-> m + 8
            float ys[16], yo[8];
#pragma unroll
            for (int i = 0; i < 16; ++i) ys[i] = yq[i].x + yq[i].y;
            asm volatile("s_nop 1\n\t"
                         "v_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\t"
                         "v_permlane32_swap_b32 %2, %10\n\tv_permlane32_swap_b32 %3, %11\n\t"
                         "v_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\t"
                         "v_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15\n\ts_nop 1"
                         : "+v"(ys[0]), "+v"(ys[1]), "+v"(ys[2]), "+v"(ys[3]), "+v"(ys[4]), "+v"(ys[5]), "+v"(ys[6]), "+v"(ys[7]),
                           "+v"(ys[8]), "+v"(ys[9]), "+v"(ys[10]), "+v"(ys[11]), "+v"(ys[12]), "+v"(ys[13]), "+v"(ys[14]), "+v"(ys[15]));
#pragma unroll
            for (int m = 0; m < 8; ++m) yo[m] = ys[m] + ys[m + 8];   // lower half-wave: token m of the lane's 16, upper: token m + 8
            sYw[0] = make_float4(yo[0], yo[1], yo[2], yo[3]);
            sYw[32] = make_float4(yo[4], yo[5], yo[6], yo[7]);
        }
        MMU_LDS_BARRIER();
        {
            const float4 ya = sYc[io_q], yb = sYc[128 + io_q];
            float y[4] = {ya.x + yb.x, ya.y + yb.y, ya.z + yb.z, ya.w + yb.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = fmaf(Dv, u_cur[i], y[i]);
            const unsigned so = tile_off(k);
            if constexpr (HAS_OUT) buf_store4<io_t>(r_out, voff_io, so, y);
            if constexpr (HAS_Z) {
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] *= z_n[i] * sigmoidf_(z_n[i]);
                buf_store4<io_t>(r_oz, voff_io, so, y);
            }
        }
        prepare();
    }
}

inline bool al16(const void *q) { return q == nullptr || ((uintptr_t)q & 15) == 0; }
inline bool m4(long v) { return (v & 3) == 0; }

}  // namespace

#ifdef MMU_STREAM_STAMPS
extern "C" int mmu_debug_stream_stamps(unsigned long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stream_stamps), sizeof(g_stream_stamps)) == hipSuccess ? 0 : 1;
}
#endif

int mmu_scan_fwd_stream(const ScanArgs &a, int dtype, hipStream_t st) {
    // MMU_SCAN_STREAM=0 keeps every call on the chunk-parallel kernels (A/B runs, tests/test_hip_kernels.py)
    if (const char *e = getenv("MMU_SCAN_STREAM"); e && e[0] == '0') return 0;
    const int es = dtype == MMU_DTYPE_F32 ? 4 : 2;
    const long bcm = dtype == MMU_DTYPE_F32 ? 3 : 7;  // B / C rows are read 8 tokens (16 or 32 B) per lane
    if (a.dstate != 16 || a.seqlen % ST_TT != 0 || a.dim % a.ngroups != 0 || (a.dim / a.ngroups) % ST_CH != 0) return 0;
    if ((long)a.batch * a.dim < 512) return 0;  // too few rows to fill the chip: the chunk-parallel kernels win
    if ((a.z == nullptr) != (a.out_z == nullptr)) return 0;
    const bool ok = al16(a.u) && al16(a.delta) && al16(a.z) && al16(a.out) && al16(a.out_z) && al16(a.B) && al16(a.C) &&
                    m4(a.u_bs) && m4(a.u_ds) && m4(a.delta_bs) && m4(a.delta_ds) &&
                    (!a.z || (m4(a.z_bs) && m4(a.z_ds) && m4(a.out_z_bs) && m4(a.out_z_ds))) &&
                    (!a.out || (m4(a.out_bs) && m4(a.out_ds))) && !(a.B_bs & bcm) && !(a.B_gs & bcm) &&
                    !(a.B_ns & bcm) && !(a.C_bs & bcm) && !(a.C_gs & bcm) && !(a.C_ns & bcm);
    if (!ok) return 0;
    // buffer addressing: 32-bit byte offsets inside one (batch, channel) row / one B or C row pair
    if ((long)a.seqlen * es >= (1L << 31) || (std::max(a.B_ns, a.C_ns) + a.seqlen) * es >= (1L << 31) ||
        (long)a.n_chunks * 2 * 16 * 4 >= (1L << 31))
        return 0;
    const unsigned grid = (unsigned)a.batch * (a.dim / ST_CH);
    int r = 0;
    // MMU_SCAN_STREAM=1: the two-barrier kernel of rounds 2-3 (A/B runs); default: the one-barrier kernel
    static const int ver = [] { const char *e = getenv("MMU_SCAN_STREAM"); return e && e[0] >= '1' && e[0] <= '3' ? e[0] - '0' : 3; }();
#define ST_LAUNCH(T)                                                                   \
    MMU_BOOL(a.z != nullptr, HAS_Z, MMU_BOOL(a.softplus != 0, SOFTPLUS, MMU_BOOL(a.out != nullptr, HAS_OUT, { \
        if (ver == 1) {                                                                                      \
            r = set_lds(scan_fwd_stream_kernel<T, HAS_Z, SOFTPLUS, HAS_OUT>, ST_LDS);                        \
            if (!r) scan_fwd_stream_kernel<T, HAS_Z, SOFTPLUS, HAS_OUT><<<grid, 512, ST_LDS, st>>>(a);       \
        } else if (ver == 3) {                                                                               \
            r = set_lds(scan_fwd_stream16_kernel<T, HAS_Z, SOFTPLUS, HAS_OUT>, ST_LDS);                      \
            if (!r) scan_fwd_stream16_kernel<T, HAS_Z, SOFTPLUS, HAS_OUT><<<grid, 512, ST_LDS, st>>>(a);     \
        } else {                                                                                             \
            r = set_lds(scan_fwd_stream2_kernel<T, HAS_Z, SOFTPLUS, HAS_OUT>, S2_LDS);                       \
            if (!r) scan_fwd_stream2_kernel<T, HAS_Z, SOFTPLUS, HAS_OUT><<<grid, 512, S2_LDS, st>>>(a);      \
        }                                                                                                    \
    });););
    if (dtype == MMU_DTYPE_F32) {
        ST_LAUNCH(float);
    } else {
        ST_LAUNCH(bf16_t);
    }
#undef ST_LAUNCH
    if (r) return -1;
    if (hipGetLastError() != hipSuccess) {
        mmu_fail("scan_fwd_stream: launch failed");
        return -1;
    }
    return 1;
}
