// selective_scan_bwd_w8.hip -- backward apply (K4) of the chunk-parallel selective scan for dstate == 16 on full
// 512-token tiles, and the reduction (K5) of its dA / dD / dbias partials.
//
// Replaces, for the shapes it takes, chunk_apply_bwd_p4_kernel (selective_scan.hip) in the pipeline
//   K1 chunk_reduce<BWD> -> K2 chunk_carry(reverse) -> K4 -> K5
// (reference: selective_scan_bwd_kernel.cuh:75-531, one block per (batch, channel) walking L backwards).
//
// What limits K4 is VALU issue, and half of the p4 kernel's instructions were not gradient math: the two
// cross-lane affine scans (forward state, adjoint) cost ~55 DPP instructions per state pair however many tokens
// a lane owns, and every one of its four waves recomputed softplus / sigmoid / dy for all tokens of the tile.
// This kernel re-cuts the tile so that both are paid once per EIGHT tokens:
//   * tile = 512 tokens, lane l owns tokens 8l..8l+7: one scan pair per 8 tokens instead of per 4;
//   * 8 waves, wave w owns the ONE state pair (2w, 2w+1): its B / C values and its dB / dC sums live in registers
//     for the whole channel loop (no B / C tile in LDS);
//   * the per-token preparation (softplus, sigmoid(z), dy, ...) is done ONCE per token: wave w prepares tokens
//     64w..64w+63 (one per lane, coalesced dword loads) of the NEXT channel and publishes dl / dl*u / dy through
//     LDS; the same lane later finishes that token (sums the eight waves' partial y / q / dd, writes du, ddelta,
//     dz, out_z), so nothing per-token is recomputed or re-read;
//   * a_t h_{t-1} is not kept: g_t a_t h_{t-1} = (a_t g_t) h_{t-1} and a_t g_t is the adjoint carry anyway, so the
//     forward recompute is one packed fma per token and 16 registers shorter.
// One LDS-only barrier per channel; exchange buffers double-buffered by channel parity.
//
// grid (L / 512, batch, ngroups), block 512, one workgroup per CU (118 KiB LDS).
// LDS (floats): xch[2][8][1632] | prep[2][3][544] | slots[8][2][24]
//   prep rows: token T sits at ((T >> 2) & 1) * 288 + (T >> 3) * 4 + (T & 3): a lane's two float4 groups are
//   contiguous across lanes (b128 reads), the per-token dword writes fall in 64 distinct banks.
//   xch row of a wave: (q, dd) pairs of token T at ((T >> 1) & 3) * 272 + (T >> 3) * 4 + (T & 1) * 2 (b128 writes of
//   two tokens, b64 reads of one), then y in a prep-style row at +1088.
// Partials: part8[b][tile][d][wave][4] = (dA[2w], dA[2w+1], dD share, dbias share), summed by
// reduce_partials_w8_kernel in a fixed order.
#include <type_traits>
#include "mmu_common.h"
#include "scan_common.h"
#include "../../include/mmunet_amd.h"

// Diagnostic build only (-DMMU_W8_STAMPS, tools/dbg/w8_stamps.sh): s_memtime at the phase boundaries of channels
// 16..23 of two workgroups, read back through mmu_debug_w8_stamps.  In the product build no stamp executes.
#ifdef MMU_W8_STAMPS
__device__ unsigned long long g_w8_stamps[2 * 8 * 8 * 8];  // [block sel][wave][channel 16..23][slot]
#define W8_STAMP(slot)                                                                                              \
    do {                                                                                                            \
        if ((blockIdx.x == 0 || blockIdx.x == 77) && blockIdx.y == 0 && ch >= 16 && ch < 24 && lane == 0)           \
            g_w8_stamps[(((blockIdx.x != 0) * 8 + w) * 8 + (ch - 16)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define W8_STAMP(slot)
#endif

namespace {

constexpr int W8_TT = 512, W8_AS = 544, W8_SUB = 272, W8_XS = 1632, W8_SUB2 = 272;

// The per-lane stream pointers live in the GLOBAL address space by type: they are made opaque to the compiler below
// (asm "+v"), and a generic pointer would lose the address-space inference there -- flat_load / flat_store, which
// count in lgkmcnt as well and return out of order.
#define MMU_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ float ld1(const MMU_GLOBAL float *p) { return *p; }
__device__ __forceinline__ float ld1(const MMU_GLOBAL bf16_t *p) {
    bf16_t v;
    v.bits = *reinterpret_cast<const MMU_GLOBAL uint16_t *>(p);
    return to_f32(v);
}
__device__ __forceinline__ void st1(MMU_GLOBAL float *p, float v) { *p = v; }
__device__ __forceinline__ void st1(MMU_GLOBAL bf16_t *p, float v) {
    *reinterpret_cast<MMU_GLOBAL uint16_t *>(p) = from_f32<bf16_t>(v).bits;
}

// Y_IN: the forward's y (before gating) is read from p.out instead of being recomputed: no C h products, no y row in
// the exchange (a third of its LDS traffic), no eight-wave sum per token.
template <typename io_t, bool HAS_Z, bool HAS_OZ, bool Y_IN>
__global__ __launch_bounds__(512, 1) void chunk_apply_bwd_w8_kernel(ScanArgs p) {
    static_assert(HAS_Z || !Y_IN, "y is only needed with z");
    constexpr int N = 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile = blockIdx.x, b = blockIdx.y;
    const int g = blockIdx.z / p.d_splits, sp = blockIdx.z - g * p.d_splits;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n0 = 2 * w;
    const int t0 = tile * W8_TT, c0 = tile * 4, n_tiles = gridDim.x;
    float *xch = smem;                                   // [2][8][XS]
    float *prep = smem + 2 * 8 * W8_XS;                  // [2][3][AS]
    float *slots = prep + 2 * 3 * W8_AS + w * 48;        // [8][2][24]: A2 pair | bias | D | h0 pair x 4 rows | g0 pair x 4 rows
    const int dpg = p.dim / p.ngroups, cps = (dpg + p.d_splits - 1) / p.d_splits;   // host: no range is empty
    const int dbeg = g * dpg + sp * cps, dend = min(dbeg + cps, (g + 1) * dpg);
    const unsigned T = w * 64 + lane;                    // the token this lane prepares and finishes
    const int posT = ((T >> 2) & 1) * W8_SUB + (T >> 3) * 4 + (T & 3);
    const int posL = lane * 4;                           // this lane's 8 tokens: float4 at posL and posL + SUB
    const int posT2 = ((T >> 1) & 3) * W8_SUB2 + (T >> 3) * 4 + (T & 1) * 2;   // (q, dd) of token T in an xch row

    // this wave's state pair of B and C for the lane's 8 tokens, in registers for the whole channel loop
    v2f Bv[8], Cv[8];
    {
        const io_t *Bg = (const io_t *)p.B + (long)b * p.B_bs + (long)g * p.B_gs + (long)n0 * p.B_ns + t0 + lane * 8;
        const io_t *Cg = (const io_t *)p.C + (long)b * p.C_bs + (long)g * p.C_gs + (long)n0 * p.C_ns + t0 + lane * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float r0[4], r1[4];
            load_k<io_t, 4, true>(Bg + 4 * j, 4, true, r0);
            load_k<io_t, 4, true>(Bg + p.B_ns + 4 * j, 4, true, r1);
#pragma unroll
            for (int i = 0; i < 4; ++i) Bv[4 * j + i] = v2f{r0[i], r1[i]};
            load_k<io_t, 4, true>(Cg + 4 * j, 4, true, r0);
            load_k<io_t, 4, true>(Cg + p.C_ns + 4 * j, 4, true, r1);
#pragma unroll
            for (int i = 0; i < 4; ++i) Cv[4 * j + i] = v2f{r0[i], r1[i]};
        }
    }

    // running per-lane pointers (token T of the tile; one 64-bit add per stream and channel -- the scalar file has
    // no room for nine uniform bases and their strides): f_* = the channel of the next fetch (two ahead of the
    // loop), o_* = the loop's channel
    const MMU_GLOBAL io_t *f_dl = (const MMU_GLOBAL io_t *)((const io_t *)p.delta + (long)b * p.delta_bs + (long)dbeg * p.delta_ds + t0 + T);
    const MMU_GLOBAL io_t *f_u = (const MMU_GLOBAL io_t *)((const io_t *)p.u + (long)b * p.u_bs + (long)dbeg * p.u_ds + t0 + T);
    const MMU_GLOBAL io_t *f_go = (const MMU_GLOBAL io_t *)((const io_t *)p.dout + (long)b * p.dout_bs + (long)dbeg * p.dout_ds + t0 + T);
    const MMU_GLOBAL io_t *f_z = (const MMU_GLOBAL io_t *)(HAS_Z ? (const io_t *)p.z + (long)b * p.z_bs + (long)dbeg * p.z_ds + t0 + T : nullptr);
    const MMU_GLOBAL io_t *f_y = (const MMU_GLOBAL io_t *)(Y_IN ? (const io_t *)p.out + (long)b * p.out_bs + (long)dbeg * p.out_ds + t0 + T : nullptr);
    MMU_GLOBAL io_t *o_du = (MMU_GLOBAL io_t *)((io_t *)p.du + (long)b * p.du_bs + (long)dbeg * p.du_ds + t0 + T);
    MMU_GLOBAL io_t *o_dd = (MMU_GLOBAL io_t *)((io_t *)p.ddelta + (long)b * p.ddelta_bs + (long)dbeg * p.ddelta_ds + t0 + T);
    MMU_GLOBAL io_t *o_dz = (MMU_GLOBAL io_t *)(HAS_Z ? (io_t *)p.dz + (long)b * p.dz_bs + (long)dbeg * p.dz_ds + t0 + T : nullptr);
    MMU_GLOBAL io_t *o_oz = (MMU_GLOBAL io_t *)(HAS_OZ ? (io_t *)p.out_z + (long)b * p.out_z_bs + (long)dbeg * p.out_z_ds + t0 + T : nullptr);
    // one opaque 64-bit register pair per stream: left visible, the bf16 build splits each into a uniform base (kept in
    // VGPRs: the scalar file is full) plus a lane offset -- 12 more VGPRs, 20 spilled, 18 v_mov_b64 per two channels
    asm volatile("" : "+v"(f_dl), "+v"(f_u), "+v"(f_go), "+v"(f_z), "+v"(f_y));
    asm volatile("" : "+v"(o_du), "+v"(o_dd), "+v"(o_dz), "+v"(o_oz));
    float *part8 = p.part + (((long)b * n_tiles + tile) * p.dim + dbeg) * 32 + w * 4 + ((lane - 12) & 3);
    // the 20 scalars of a channel: A pair, bias, D, and the carries of the wave's state pair at the four 128-token
    // chunks of the tile -- h entering chunk c0 + r from the left (the forward's chunk states x), the adjoint entering
    // it from the right (the carried reverse aggregates gx): every 16-lane DPP row is one chunk, so both cross-lane
    // scans stay inside a row (4 steps, no lane reversal).  One gathered load: lane j < 20 owns one scalar as
    // base + channel * stride; the other lanes re-read lane 0's word.
    const int jj = lane < 20 ? lane : 0;
    const float *gbase = p.A + (long)(n0 + (jj & 1)) * p.A_ns;
    unsigned gstride = (unsigned)p.A_ds;
    float gscale = MMU_LOG2E;
    if (jj == 2) {
        gbase = p.delta_bias ? p.delta_bias : p.A;
        gstride = p.delta_bias ? 1u : 0u;
        gscale = p.delta_bias ? 1.f : 0.f;
    } else if (jj == 3) {
        gbase = p.D ? p.D : p.A;
        gstride = p.D ? 1u : 0u;
        gscale = p.D ? 1.f : 0.f;
    } else if (jj >= 4 && jj < 12) {
        const int cc = c0 + ((jj - 4) >> 1) - 1;
        gbase = p.x + ((long)b * p.dim * p.n_chunks + (cc >= 0 ? cc : 0)) * 2 * N + 2 * (n0 + (jj & 1)) + 1;
        gstride = (unsigned)p.n_chunks * 2 * N;
        gscale = cc >= 0 ? 1.f : 0.f;
    } else if (jj >= 12) {
        const int cc = c0 + ((jj - 12) >> 1) + 1;
        gbase = p.gx + ((long)b * p.dim * p.n_chunks + (cc < p.n_chunks ? cc : 0)) * 2 * N + 2 * (n0 + (jj & 1)) + 1;
        gstride = (unsigned)p.n_chunks * 2 * N;
        gscale = cc < p.n_chunks ? 1.f : 0.f;
    }
    const int row = lane >> 4;
    const bool rl0 = (lane & 15) == 0, rl15 = (lane & 15) == 15;

    // ---- per-token pipeline: raw loads (two channels ahead) -> prepared values (one channel ahead) ----------
    const float *gp = gbase + (unsigned long)dbeg * gstride;
    struct Raw { float dl, u, go, z, y, gv; };   // one channel's loads in flight; two sets, alternating by channel parity
    auto fetch = [&](Raw &r) {
        r.dl = ld1(f_dl);
        r.u = ld1(f_u);
        r.go = ld1(f_go);
        r.z = 0.f;
        if constexpr (HAS_Z) r.z = ld1(f_z);
        r.y = 0.f;
        if constexpr (Y_IN) r.y = ld1(f_y);
        r.gv = *gp;
    };
    auto advance_fetch = [&]() {   // to the next channel (uniform pointers: scalar adds)
        f_dl += p.delta_ds;
        f_u += p.u_ds;
        f_go += p.dout_ds;
        if constexpr (HAS_Z) f_z += p.z_ds;
        if constexpr (Y_IN) f_y += p.out_ds;
        gp += gstride;
    };
    // what the finishing step of a channel needs of its token
    struct Tok { float dl, u, dy, dsp, F, G, D, y; };
    auto prepare = [&](int par, const Raw &r, Tok &k) {   // a channel's loads -> prep[par], slots[par], k
        const float r_dl = r.dl, r_u = r.u, r_go = r.go, r_z = r.z;
        const float gvs = r.gv * gscale;
        if (lane < 20) slots[par * 24 + lane] = gvs;
        const float bias = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gvs), 2));
        k.D = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gvs), 3));
        const float vraw = r_dl + bias;
        k.dl = p.softplus ? softplus_thr(vraw) : vraw;
        k.dsp = (p.softplus && vraw <= 20.f) ? sigmoidf_(vraw) : 1.f;   // bwd_kernel.cuh:439-453
        k.u = r_u;
        k.y = r.y;
        if constexpr (HAS_Z) {
            const float zs = sigmoidf_(r_z);
            k.G = r_z * zs;                                  // out_z = y * G ; dy = dout * G
            k.F = r_go * zs * (1.f + r_z * (1.f - zs));      // dz = y * F
            k.dy = r_go * k.G;
        } else {
            k.G = k.F = 0.f;
            k.dy = r_go;
        }
        float *pr = prep + par * (3 * W8_AS) + posT;
        pr[0] = k.dl;
        pr[W8_AS] = k.dl * k.u;
        pr[2 * W8_AS] = k.dy;
    };

    // loads run three channels ahead of the loop (HBM latency under load is about one channel's time): the loop at
    // channel d prepares d + 1 from the set fetched two iterations ago and refills that set with d + 3
    Tok cur;
    Raw raw0, raw1;
    int fd = dbeg;   // the channel the f_* pointers stand on
    auto next_fetch = [&](Raw &r) {
        if (fd + 1 < dend) {
            advance_fetch();
            ++fd;
        }
        fetch(r);
    };
    fetch(raw0);
    prepare(0, raw0, cur);
    next_fetch(raw1);   // dbeg + 1
    next_fetch(raw0);   // dbeg + 2
    MMU_LDS_BARRIER();

    v2f accB[8], accC[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) accB[i] = accC[i] = v2f{0.f, 0.f};

#ifdef MMU_W8_STAMPS
    int ch = 0;
#endif
    // Finishing token T of a channel: meeting the eight state pairs (sums of q / dd / y over the waves), du, ddelta,
    // dz, out_z, and this wave's share of the dA / dD / dbias partials.  It runs at the top of the NEXT channel's
    // call, right after the barrier that completed the exchange rows.  (Measured and dropped: running it inside
    // that channel's main phase to hide its LDS / reduction latencies, and alternating s_setprio by phase to keep
    // the two waves of a SIMD in step -- both within noise of this form, 1.27-1.31 ms for the headline backward.)
    struct Fin { v2f qd[8]; float y[8]; };
    auto finish_issue = [&](auto PAR, Fin &f) {
        constexpr int par = decltype(PAR)::value;
        const float *xc = xch + par * (8 * W8_XS);
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) {
            f.qd[wv] = *reinterpret_cast<const v2f *>(xc + wv * W8_XS + posT2);
            if constexpr (HAS_Z && !Y_IN) f.y[wv] = xc[wv * W8_XS + 4 * W8_SUB2 + posT];
        }
    };
    auto finish_done = [&](const Fin &f, const Tok &k, v2f dA) {
        v2f QD = (f.qd[0] + f.qd[1]) + (f.qd[2] + f.qd[3]);
        QD += (f.qd[4] + f.qd[5]) + (f.qd[6] + f.qd[7]);
        const float Q = QD.x, DDs = QD.y;
        const float du = fmaf(k.D, k.dy, k.dl * Q);
        const float ddel = fmaf(k.u, Q, DDs) * k.dsp;
        st1(o_du, du);
        st1(o_dd, ddel);
        if constexpr (HAS_Z) {
            float yt;
            if constexpr (Y_IN) {
                yt = k.y;   // the forward's out already holds D u
            } else {
                const float Y = ((f.y[0] + f.y[1]) + (f.y[2] + f.y[3])) + ((f.y[4] + f.y[5]) + (f.y[6] + f.y[7]));
                yt = fmaf(k.D, k.u, Y);
            }
            st1(o_dz, yt * k.F);
            if constexpr (HAS_OZ) st1(o_oz, yt * k.G);
        }
        const float v = wave_sum4_swap(dA.x, dA.y, k.dy * k.u, ddel);   // lanes 12..15: the four wave totals
        if (lane >= 12 && lane < 16) *part8 = v;
        o_du += p.du_ds;
        o_dd += p.ddelta_ds;
        if constexpr (HAS_Z) {
            o_dz += p.dz_ds;
            if constexpr (HAS_OZ) o_oz += p.out_z_ds;
        }
        part8 += 32;
    };

    // one channel; `par` is a compile-time constant (the loop below is unrolled by two: no parity arithmetic, no
    // copies of the prepared tokens).  FIN: the previous channel (token `prv`, dA sums `dA_prv`) is finished inside.
    v2f dA_prv = v2f{0.f, 0.f};
    auto channel = [&](auto PAR, auto FIN, Tok &prv_nxt, Raw &raw) {
        constexpr int par = decltype(PAR)::value;
        constexpr bool fin = decltype(FIN)::value;
        // (Measured and dropped in round 4: issuing the finishing reads behind this channel's reads and consuming them
        // behind its 16 exps -- 1,066 vs 1,066 us, 6 more VGPRs.)
#ifndef MMU_W8_PRIO
#define MMU_W8_PRIO 0
#endif
        // VALU issue goes to the OLDER of the two waves of a SIMD (waves 0-3): it runs its channel at nearly the
        // single-wave rate (3,000 cycles) and then waits 1,400 at the barrier while the younger one, starved until then,
        // finishes alone (stamps, round 4).  MMU_W8_PRIO=1 lets the younger half lead through the first half of the
        // channel and hand the lead back for the walk: the waves then arrive together (barrier waits 1,400 -> 300
        // cycles, SQ_WAIT_ANY -9 %) -- and the kernel takes 987 instead of 970 us: what was barrier wait becomes issue
        // stall (SQ_WAIT_INST_ANY +26 %), the SIMD's vector pipe is the limit either way.  Off.
        if (MMU_W8_PRIO && w >= 4) __builtin_amdgcn_s_setprio(1);
        if constexpr (fin) {
            Fin fbuf;
            finish_issue(std::integral_constant<int, par ^ 1>{}, fbuf);
            finish_done(fbuf, prv_nxt, dA_prv);
        }
        W8_STAMP(0);
        // ---- this channel's scalars and prepared per-token values ------------------------------------
        const v2f a2 = *reinterpret_cast<const v2f *>(slots + par * 24);
        const v2f h0 = *reinterpret_cast<const v2f *>(slots + par * 24 + 4 + 2 * row);    // this row's chunk: state entering it
        const v2f g0 = *reinterpret_cast<const v2f *>(slots + par * 24 + 12 + 2 * row);   // adjoint entering it from the right
        v2f dl2[4], dlu2[4], dy2[4];
        {
            const float *pr = prep + par * (3 * W8_AS) + posL;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 f0 = *reinterpret_cast<const float4 *>(pr + j * W8_SUB);
                const float4 f1 = *reinterpret_cast<const float4 *>(pr + W8_AS + j * W8_SUB);
                const float4 f2 = *reinterpret_cast<const float4 *>(pr + 2 * W8_AS + j * W8_SUB);
                dl2[2 * j] = v2f{f0.x, f0.y}; dl2[2 * j + 1] = v2f{f0.z, f0.w};
                dlu2[2 * j] = v2f{f1.x, f1.y}; dlu2[2 * j + 1] = v2f{f1.z, f1.w};
                dy2[2 * j] = v2f{f2.x, f2.y}; dy2[2 * j + 1] = v2f{f2.z, f2.w};
            }
        }
#ifdef MMU_W8_STAMPS
        asm volatile("" : "+v"(dl2[0]), "+v"(dy2[3]), "+v"(dlu2[3]));
#endif
        W8_STAMP(7);
        // ---- forward recompute and the two cross-lane scans -------------------------------------------
        v2f a[8], hh[8];
        float dlsum;
        {
            const v2f s01 = dl2[0] + dl2[1], s23 = dl2[2] + dl2[3];
            const v2f s = s01 + s23;
            dlsum = s.x + s.y;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[2 * k] = exp2_2(mul_bcast<0>(dl2[k], a2));
            a[2 * k + 1] = exp2_2(mul_bcast<1>(dl2[k], a2));
        }
        const v2f P = exp2_2(a2 * dlsum);
        v2f S, R, h_in, gam;
        {
            v2f bb[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bb[2 * k] = mul_bcast<0>(dlu2[k], Bv[2 * k]);
                bb[2 * k + 1] = mul_bcast<1>(dlu2[k], Bv[2 * k + 1]);
            }
            S = bb[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) S = fma2(a[i], S, bb[i]);
            R = a[7] * mul_bcast<1>(dy2[3], Cv[7]);
#pragma unroll
            for (int i = 6; i >= 0; --i)
                R = a[i] * ((i & 1) ? fma_bcast<1>(dy2[i >> 1], Cv[i], R) : fma_bcast<0>(dy2[i >> 1], Cv[i], R));
            const v2f S_in = fma2(P, h0, S), R_in = fma2(P, g0, R);
            S = rl0 ? S_in : S;
            R = rl15 ? R_in : R;
            float P0 = P.x, S0 = S.x, P1 = P.y, S1 = S.y;
            float Q0 = P.x, R0 = R.x, Q1 = P.y, R1 = R.y;
            row_scan_affine_x2(P0, S0, P1, S1);     // inclusive prefix inside the chunk's 16 lanes
            row_rscan_affine_x2(Q0, R0, Q1, R1);    // inclusive suffix
            // state entering this lane's tokens = the previous lane's prefix (the chunk carry on the row's first lane);
            // adjoint entering from the right = the next lane's suffix (the chunk carry on the row's last lane)
            h_in = v2f{dpp_mov<MMU_DPP_ROW_SHR(1), 0xf>(h0.x, S0), dpp_mov<MMU_DPP_ROW_SHR(1), 0xf>(h0.y, S1)};
            v2f h = h_in;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                h = fma2(a[i], h, bb[i]);
                hh[i] = h;
            }
            gam = v2f{dpp_mov<MMU_DPP_ROW_SHL(1), 0xf>(g0.x, R0), dpp_mov<MMU_DPP_ROW_SHL(1), 0xf>(g0.y, R1)};
        }
#ifdef MMU_W8_STAMPS
        asm volatile("" : "+v"(gam), "+v"(hh[7]));
#endif
        W8_STAMP(6);
        if (MMU_W8_PRIO && w >= 4) __builtin_amdgcn_s_setprio(0);
        // ---- adjoint walk, right to left: gradients of this state pair -------------------------------------
        //   g_t = dy_t C_t + gamma_{t+1};  gamma_t = a_t g_t;  with q_t = sum_n g B:
        //   du = delta q + D dy;  ddelta' = u q + sum_n A (gamma_t h_{t-1});  dA = sum_t delta (gamma_t h_{t-1});
        //   dB = (delta u) g;  dC = dy h
        const v2f An = a2 * MMU_LN2;
        v2f dAp = v2f{0.f, 0.f};
        float qv[8], ddv[8], yv[8];
        float *xc = xch + par * (8 * W8_XS) + w * W8_XS + posL;
#pragma unroll
        for (int i = 7; i >= 0; --i) {
            const int k = i >> 1;
            const v2f gt = (i & 1) ? fma_bcast<1>(dy2[k], Cv[i], gam) : fma_bcast<0>(dy2[k], Cv[i], gam);
            gam = a[i] * gt;
            const v2f gah = gam * (i ? hh[i - 1] : h_in);
            {   // sums over the state pair: one packed product + one add each
                const v2f tq = gt * Bv[i], td = An * gah;
                qv[i] = tq.x + tq.y;
                ddv[i] = td.x + td.y;
                if constexpr (HAS_Z && !Y_IN) {
                    const v2f ty = Cv[i] * hh[i];
                    yv[i] = ty.x + ty.y;
                }
            }
            dAp = (i & 1) ? fma_bcast<1>(dl2[k], gah, dAp) : fma_bcast<0>(dl2[k], gah, dAp);
            accB[i] = (i & 1) ? fma_bcast<1>(dlu2[k], gt, accB[i]) : fma_bcast<0>(dlu2[k], gt, accB[i]);
            accC[i] = (i & 1) ? fma_bcast<1>(dy2[k], hh[i], accC[i]) : fma_bcast<0>(dy2[k], hh[i], accC[i]);
            // the exchange rows leave as soon as their tokens are done: six 128-bit LDS stores per lane take the eight
            // waves ~600 cycles of the store path, which the rest of the walk now covers (they used to follow the loop,
            // with every wave waiting for its last one in front of the barrier)
            if (!(i & 1))   // tokens i, i + 1: (q, dd, q, dd)
                *reinterpret_cast<float4 *>(xc + (i >> 1) * W8_SUB2) = make_float4(qv[i], ddv[i], qv[i + 1], ddv[i + 1]);
            if constexpr (HAS_Z && !Y_IN) {
                if (!(i & 3))
                    *reinterpret_cast<float4 *>(xc + 4 * W8_SUB2 + (i >> 2) * W8_SUB) = make_float4(yv[i], yv[i + 1], yv[i + 2], yv[i + 3]);
            }
            if (!(i & 1)) __builtin_amdgcn_sched_barrier(0);
        }
        W8_STAMP(1);
        dA_prv = dAp;
        // ---- the next channel's tokens (loads issued two channels ago), then the loads of the one after it ----
        W8_STAMP(2);
        prepare(par ^ 1, raw, prv_nxt);
        W8_STAMP(3);
        next_fetch(raw);
        W8_STAMP(4);
        MMU_LDS_BARRIER();
        W8_STAMP(5);
#ifdef MMU_W8_STAMPS
        ++ch;
#endif
    };
    // tokE / tokO: prepared tokens of the even / odd channels (counted from dbeg).  A channel call finishes the
    // previous channel from the token of the OTHER parity and then overwrites it with the next channel's.
    Tok tokE = cur, tokO = cur;
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    constexpr std::true_type FIN{};
    channel(P0, std::false_type{}, tokO, raw1);             // dbeg: nothing to finish yet
    int d = dbeg + 1;
    for (; d + 1 < dend; d += 2) {
        channel(P1, FIN, tokE, raw0);                       // d odd: finishes d - 1 (tokE), prepares d + 1 into tokE
        channel(P0, FIN, tokO, raw1);
    }
    if (d < dend) {                                         // an odd channel is left
        channel(P1, FIN, tokE, raw0);
        Fin fb;
        finish_issue(P1, fb);
        finish_done(fb, tokO, dA_prv);
    } else {
        Fin fb;
        finish_issue(P0, fb);
        finish_done(fb, tokE, dA_prv);
    }
    // this wave's two rows of dB / dC
    float *dBg = p.dB + sp * p.dBC_ss + (long)b * p.dB_bs + (long)g * p.dB_gs + (long)n0 * p.dB_ns + t0 + lane * 8;
    float *dCg = p.dC + sp * p.dBC_ss + (long)b * p.dC_bs + (long)g * p.dC_gs + (long)n0 * p.dC_ns + t0 + lane * 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        *reinterpret_cast<float4 *>(dBg + 4 * j) = make_float4(accB[4 * j].x, accB[4 * j + 1].x, accB[4 * j + 2].x, accB[4 * j + 3].x);
        *reinterpret_cast<float4 *>(dBg + p.dB_ns + 4 * j) = make_float4(accB[4 * j].y, accB[4 * j + 1].y, accB[4 * j + 2].y, accB[4 * j + 3].y);
        *reinterpret_cast<float4 *>(dCg + 4 * j) = make_float4(accC[4 * j].x, accC[4 * j + 1].x, accC[4 * j + 2].x, accC[4 * j + 3].x);
        *reinterpret_cast<float4 *>(dCg + p.dC_ns + 4 * j) = make_float4(accC[4 * j].y, accC[4 * j + 1].y, accC[4 * j + 2].y, accC[4 * j + 3].y);
    }
}

// ---------------------------------------------------------------------------
// K5 for the w8 partial layout: part8[bt][d][wave][4] -> dA[d][16], dD[d], dbias[d], summed over a slice of 512 rows
// bt = (batch, tile) in a fixed order.  grid (dim, n_slices), block 256.  One slice: straight to the outputs;
// several: slice sums to part2[slice][d][32], added by reduce_slices_w8_kernel.  No atomics.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void w8_emit(const float *t32 /* LDS, 32 column sums */, int d, float *dA, float *dD, float *dbias,
                                        const float *Asc /* optional: dA *= A */) {
    const int j = threadIdx.x;
    if (j < 16) {
        const float v = t32[(j >> 1) * 4 + (j & 1)];
        dA[(long)d * 16 + j] = Asc ? v * Asc[(long)d * 16 + j] : v;
    } else if (j < 18) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) s += t32[wv * 4 + 2 + (j - 16)];
        float *o = j == 16 ? dD : dbias;
        if (o) o[d] = s;
    }
}

__global__ __launch_bounds__(256) void reduce_partials_w8_kernel(const float *__restrict__ part8, int BT, int dim, float *dA,
                                                                 float *dD, float *dbias, float *__restrict__ part2,
                                                                 const float *__restrict__ Asc) {
    __shared__ float red[256];
    __shared__ float t32[32];
    const int d = blockIdx.x;
    const int r0 = blockIdx.y * 512;
    const int r1 = r0 + 512 < BT ? r0 + 512 : BT;
    const int j = threadIdx.x & 31, r = threadIdx.x >> 5;
    float s = 0.f;
    for (int bt = r0 + r; bt < r1; bt += 8) s += part8[((long)bt * dim + d) * 32 + j];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 32 + threadIdx.x];
        if (part2)
            part2[((long)blockIdx.y * dim + d) * 32 + threadIdx.x] = t;
        else
            t32[threadIdx.x] = t;
    }
    __syncthreads();
    if (!part2) w8_emit(t32, d, dA, dD, dbias, Asc);
}

__global__ __launch_bounds__(64) void reduce_slices_w8_kernel(const float *__restrict__ part2, int n_slices, int dim, float *dA,
                                                              float *dD, float *dbias, const float *__restrict__ Asc) {
    __shared__ float t32[32];
    const int d = blockIdx.x;
    if (threadIdx.x < 32) {
        float t = 0.f;
        for (int sl = 0; sl < n_slices; ++sl) t += part2[((long)sl * dim + d) * 32 + threadIdx.x];
        t32[threadIdx.x] = t;
    }
    __syncthreads();
    w8_emit(t32, d, dA, dD, dbias, Asc);
}

unsigned long long g_lds_done[2][6];

}  // namespace

#ifdef MMU_W8_STAMPS
extern "C" int mmu_debug_w8_stamps(unsigned long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w8_stamps), sizeof(g_w8_stamps)) == hipSuccess ? 0 : 1;
}
#endif

// Takes the call (returns 1) when the shape is the kernel's: the caller has already checked dstate == 16, the
// 16-byte alignment of every row and the 2^31-byte spans (launch_bwd, selective_scan.hip).  0: not taken.
int mmu_scan_bwd_apply_w8(const ScanArgs &a, int dtype, hipStream_t st) {
    // MMU_SCAN_BWD_W8=0 keeps every call on chunk_apply_bwd_p4 (A/B runs, tests), =1 forces this kernel
    const char *e = getenv("MMU_SCAN_BWD_W8");
    if (e && e[0] == '0') return 0;
    if (a.dstate != 16 || a.seqlen % W8_TT != 0) return 0;
    const long wgs = (long)(a.seqlen / W8_TT) * a.batch * a.ngroups * a.d_splits;
    if (!(e && e[0] == '1') && wgs < mmu_cu_count()) return 0;   // too few tiles to give every CU one: p4's are half the size
    const size_t lds = sizeof(float) * (2 * 8 * W8_XS + 2 * 3 * W8_AS + 8 * 48);
    dim3 grid(a.seqlen / W8_TT, a.batch, a.ngroups * a.d_splits);
    const bool f32 = dtype == MMU_DTYPE_F32;
    hipError_t err = hipSuccess;
    MMU_BOOL(a.z != nullptr, HAS_Z, {
        MMU_BOOL(a.z != nullptr && a.out_z != nullptr, HAS_OZ, {
            MMU_BOOL(a.z != nullptr && a.out != nullptr, Y_IN, {
                if constexpr ((HAS_Z || !HAS_OZ) && (HAS_Z || !Y_IN)) {
                    constexpr int vi = HAS_Z + HAS_OZ + 3 * Y_IN;
                    if (f32) {
                        err = mmu_set_lds_once(chunk_apply_bwd_w8_kernel<float, HAS_Z, HAS_OZ, Y_IN>, (int)lds, g_lds_done[0][vi]);
                        if (err == hipSuccess) chunk_apply_bwd_w8_kernel<float, HAS_Z, HAS_OZ, Y_IN><<<grid, 512, lds, st>>>(a);
                    } else {
                        err = mmu_set_lds_once(chunk_apply_bwd_w8_kernel<bf16_t, HAS_Z, HAS_OZ, Y_IN>, (int)lds, g_lds_done[1][vi]);
                        if (err == hipSuccess) chunk_apply_bwd_w8_kernel<bf16_t, HAS_Z, HAS_OZ, Y_IN><<<grid, 512, lds, st>>>(a);
                    }
                }
            });
        });
    });
    // contract: 1 = taken, 0 = not taken, < 0 = error (mmu_fail() itself returns 1 = "an error", which would read as
    // "taken" here)
    if (err != hipSuccess) {
        mmu_fail("chunk_apply_bwd_w8: hipFuncSetAttribute: %s", hipGetErrorString(err));
        return -1;
    }
    err = hipGetLastError();
    if (err != hipSuccess) {
        mmu_fail("chunk_apply_bwd_w8: %s", hipGetErrorString(err));
        return -1;
    }
    return 1;
}

// K5 of the w8 layout.  `part8` = the partial region of the workspace, `part2` = its slice region.
int mmu_scan_bwd_reduce_w8(const float *part8, float *part2, int batch, int dim, int seqlen, float *dA, float *dD,
                           float *dbias, const float *Asc, hipStream_t st) {
    const int BT = batch * (seqlen / W8_TT);
    const int n_slices = (BT + 511) / 512;
    {   // inside a deferred scope: with the other parameter-gradient sums of the pass (deferred_reduce.hip, kind 4)
        const long job[8] = {4, (long)part8, (long)dA, (long)dD, (long)dbias, BT, dim, (long)Asc};
        if (mmu_defer_job(job)) return 0;
    }
    dim3 g5(dim, n_slices);
    reduce_partials_w8_kernel<<<g5, 256, 0, st>>>(part8, BT, dim, dA, dD, dbias, n_slices > 1 ? part2 : nullptr, Asc);
    if (n_slices > 1) reduce_slices_w8_kernel<<<dim, 64, 0, st>>>(part2, n_slices, dim, dA, dD, dbias, Asc);
    MMU_HIP_LAUNCH_CHECK("reduce_partials_w8");
    return 0;
}
