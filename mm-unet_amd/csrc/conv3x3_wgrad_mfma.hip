// conv3x3_wgrad_mfma.hip -- weight gradient of the dense 3x3 / stride 1 / padding 1 convolution on the bf16 matrix
// cores with float32 accuracy (hi/lo split, three MFMAs per product: conv3x3_mfma.hip).
//
//   dW[co][ci][kh][kw] = sum_{b,y,x} dout[b][co][y][x] * in[b][ci][y+kh-1][x+kw-1]
//
// Per 3x3 shift this is a GEMM with M = co, N = ci and the contraction over PIXELS: A[m][k] = dout[co][pixel] is
// pixel-contiguous (a lane's 8 k values are one ds_read_b128 of a [co][pixel] image), but B[k][n] = in[pixel + shift][ci]
// wants 8 consecutive pixels of one channel at an arbitrary pixel offset -- 16-byte LDS reads misalign.  gfx950's
// ds_read_b64_tr_b16 reads a 4-row x 16-column block of 16-bit elements and delivers it column-major: with the input
// patch staged as [pixel][32 ci] (64-byte rows) a B fragment is two such reads whose rows are 8 consecutive pixels,
// and a 3x3 shift is again just an address offset.  (tools/ubench/tr_probe.hip pins the lane mapping with exact data.)
//
// Workgroup = 9 waves = the 9 shifts; tile = 4 rows x 64 columns of pixels (16 k-steps of 16 pixels), 64 output
// channels x 32 input channels: wave s keeps dW[64][32] of ITS shift in 32 accumulator registers and reads, per
// k-step, 4 x ds_read_b128 (dout hi/lo, 2 row tiles) + 4 x ds_read_b64_tr_b16 (patch hi/lo) for 6 MFMAs.
// Persistent workgroups (one per CU, 116 KB of LDS) walk the tiles of one (co tile, ci chunk); the next tile's
// global loads are issued before the MFMAs of the current one.  Partials per workgroup go to a workspace, a second
// kernel adds them in fixed order (deterministic, no atomics).
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) short s8;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TH = 4, TW = 64, PH = TH + 2, PW = TW + 2, NPX = PH * PW;   // 396 patch pixels
constexpr int CI = 32, CO = 64, NT = 576;
constexpr int PATCH_IMG = NPX * CI * 2;            // [pixel][32 ci] bf16: 25,344 B
constexpr int DROW = TH * TW * 2 + 16;             // dout image row: 256 pixels bf16 + pad (bank spread): 528 B
constexpr int DOUT_IMG = CO * DROW;                // 33,792 B
constexpr int LDS_BYTES = 2 * PATCH_IMG + 2 * DOUT_IMG;   // 118,272 B

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo) {
    hi = pack_bf16(a, b);
    const float ah = __builtin_bit_cast(float, hi << 16), bh = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pack_bf16(a - ah, b - bh);
}

struct WgArgs {
    const void *x, *g;        // float32, or (XB) bfloat16: both operands are then exact bf16 values -- one MFMA per product
    float *ws;
    int B, Cin, Cout, H, W, tiles_x, tiles_y, n_cic, n_cot, wg_per_cc;
};

template <bool XB>
__global__ __launch_bounds__(NT, 1) void conv3x3_wgrad_mfma_kernel(WgArgs p) {
    using px_t = typename std::conditional<XB, unsigned short, float>::type;   // raw activations as loaded
    using dv_t = typename std::conditional<XB, v2u, float4>::type;            // four dout pixels as loaded
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *patch_hi = lds, *patch_lo = lds + PATCH_IMG;
    unsigned char *dout_hi = lds + 2 * PATCH_IMG, *dout_lo = dout_hi + DOUT_IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // = the 3x3 shift of this wave
    const int cc = blockIdx.x / p.wg_per_cc, wl = blockIdx.x - cc * p.wg_per_cc;
    const int cot = cc / p.n_cic, cic = cc - cot * p.n_cic;
    const long HW = (long)p.H * p.W;
    const int tiles_img = p.tiles_x * p.tiles_y, ntiles = tiles_img * p.B;

    // ---- staging items.  patch: (pixel, group of 8 ci) -> 8 dword loads, one 16-byte write: 396 x 4 = 1,584 items,
    // 3 rounds.  dout: (co, group of 4 pixels) -> one float4 load, one 8-byte write per image: 64 x 64 = 4,096 items,
    // 8 rounds (the last one partial).
    int p_pr[3], p_pc[3], p_cg[3], p_off[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int q = tid + NT * k;
        const bool live = q < NPX * 4;
        q = live ? q : NPX * 4 - 1;
        p_cg[k] = q / NPX;
        const int pxi = q - p_cg[k] * NPX;
        p_pr[k] = pxi / PW;
        p_pc[k] = pxi - p_pr[k] * PW;
        p_off[k] = live ? pxi * (CI * 2) + p_cg[k] * 16 : -1;
    }
    // (NT = 9 x 64: item tid + NT k is channel row wv + 9 k, pixel group tid & 63 -- one pixel offset for all eight
    // rounds, nothing per round kept in registers)
    const int d_px = (tid & 63) * 4;                  // pixel index inside the 4 x 64 tile (row-major)
    const int d_co0 = tid >> 6;
    auto d_live = [&](int k) { return d_co0 + 9 * k < CO; };
    auto d_co = [&](int k) { return d_live(k) ? d_co0 + 9 * k : CO - 1; };

    px_t px[3][8];
    float pm[3];
    dv_t dv[8];
    auto prefetch = [&](int t) {
        t = t < ntiles ? t : ntiles - 1;              // past the end: re-read the last tile (nobody stages it)
        const int b = t / tiles_img, r = t - b * tiles_img;
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int gy = y0 - 1 + p_pr[k], gx = x0 - 1 + p_pc[k];
            const bool inb = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            pm[k] = inb ? 1.f : 0.f;  // (selects below: 0 * a non-finite pixel (0, 0) would poison every padded tap)
            const px_t *s = (const px_t *)p.x + ((long)b * p.Cin + cic * CI + 8 * p_cg[k]) * HW + (inb ? (long)gy * p.W + gx : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) px[k][j] = s[j * HW];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int gy = y0 + (d_px >> 6), gx = x0 + (d_px & 63);
            const bool inb = gy < p.H && gx < p.W;     // W % 4 == 0: a group of 4 is in or out as a whole
            const long go = ((long)b * p.Cout + cot * CO + d_co(k)) * HW + (inb ? (long)gy * p.W + gx : 0);
            if constexpr (XB) {
                const v2u v = *reinterpret_cast<const v2u *>((const unsigned short *)p.g + go);
                dv[k] = inb ? v : v2u{0u, 0u};
            } else {
                const float4 v = *reinterpret_cast<const float4 *>((const float *)p.g + go);
                const float m = inb ? 1.f : 0.f;
                dv[k] = make_float4(v.x * m, v.y * m, v.z * m, v.w * m);
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            unsigned hw[4], lw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (XB)
                    hw[j] = pm[k] != 0.f ? (unsigned)px[k][2 * j] | ((unsigned)px[k][2 * j + 1] << 16) : 0u;
                else
                    split2(pm[k] != 0.f ? px[k][2 * j] : 0.f, pm[k] != 0.f ? px[k][2 * j + 1] : 0.f, hw[j], lw[j]);
            }
            if (p_off[k] >= 0) {
                *reinterpret_cast<v4u *>(patch_hi + p_off[k]) = v4u{hw[0], hw[1], hw[2], hw[3]};
                if constexpr (!XB) *reinterpret_cast<v4u *>(patch_lo + p_off[k]) = v4u{lw[0], lw[1], lw[2], lw[3]};
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if constexpr (XB) {
                if (d_live(k)) *reinterpret_cast<v2u *>(dout_hi + d_co(k) * DROW + d_px * 2) = dv[k];
            } else {
                unsigned h0, l0, h1, l1;
                split2(dv[k].x, dv[k].y, h0, l0);
                split2(dv[k].z, dv[k].w, h1, l1);
                if (d_live(k)) {
                    *reinterpret_cast<v2u *>(dout_hi + d_co(k) * DROW + d_px * 2) = v2u{h0, h1};
                    *reinterpret_cast<v2u *>(dout_lo + d_co(k) * DROW + d_px * 2) = v2u{l0, l1};
                }
            }
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    const int kh = wv / 3, kw = wv - 3 * kh;
    // A fragment (dout): lane (row r = lane & 31, half h) reads pixels 16 ks + 8 h .. + 7 of channel row m*32 + r
    const int a_lane = (lane & 31) * DROW + (lane >> 5) * 16;
    // B fragment (patch, transposed read): lane 4q + pp of a 16-lane group supplies the address of pixel row q,
    // channels 4 pp .. 4 pp + 3 of the group's 16-channel block; two reads cover the lane half's 8 pixels
    const int li = lane & 15, bq = li >> 2, bp = li & 3;
    const int b_lane = bq * (CI * 2) + (((lane >> 4) & 1) * 16 + 4 * bp) * 2 + (lane >> 5) * 8 * (CI * 2);

    // XCD-aware tile walk (workgroup ids go round-robin over the 8 XCDs; blockIdx = cc * wg_per_cc + wl): XCD k takes the
    // k-th contiguous eighth of the tile list, so the halo rows of x that neighbouring tiles share (6 patch rows per 4
    // output rows) and the dout tile the two ci chunks read are fetched into one L2.  MMU_C3WG_XCD=0: plain strided walk.
#ifndef MMU_C3WG_XCD
#define MMU_C3WG_XCD 1
#endif
    const bool xcd_walk = MMU_C3WG_XCD && (p.wg_per_cc & 7) == 0;
    const int per_xcd = (ntiles + 7) >> 3;
    int t = xcd_walk ? (wl & 7) * per_xcd + (wl >> 3) : wl;
    const int t_step = xcd_walk ? p.wg_per_cc >> 3 : p.wg_per_cc;
    const int t_end = xcd_walk ? min(((wl & 7) + 1) * per_xcd, ntiles) : ntiles;
    prefetch(t);
    for (; t < t_end; t += t_step) {
        __syncthreads();                 // the previous tile's fragments have been read
        stage();
        __syncthreads();
        prefetch(t + t_step);            // in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const int row = ks >> 2, xk = (ks & 3) * 16;            // 16 pixels of tile row `row` starting at column xk
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int off = m * 32 * DROW + (row * TW + xk) * 2 + a_lane;
                ah[m] = *reinterpret_cast<const bf16x8 *>(dout_hi + off);
                if constexpr (!XB) al[m] = *reinterpret_cast<const bf16x8 *>(dout_lo + off);
            }
            const int poff = ((row + kh) * PW + xk + kw) * (CI * 2) + b_lane;
            auto tr = [&](const unsigned char *img, int o) {
                return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s4 *)(uintptr_t)(unsigned)(uintptr_t)(img + o));
            };
            const s4 h0 = tr(patch_hi, poff), h1 = tr(patch_hi, poff + 4 * (CI * 2));
            const bf16x8 bh = __builtin_bit_cast(bf16x8, s8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]});
            if constexpr (XB) {
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh, acc[m], 0, 0, 0);
            } else {
                const s4 l0 = tr(patch_lo, poff), l1 = tr(patch_lo, poff + 4 * (CI * 2));
                const bf16x8 bl = __builtin_bit_cast(bf16x8, s8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]});
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh, acc[m], 0, 0, 0);
                }
            }
        }
    }
    // partial of this workgroup: ws[wg][co 64][ci 32][9]
    float *wp = p.ws + (long)blockIdx.x * (CO * CI * 9);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), ci = lane & 31;
            wp[(co * CI + ci) * 9 + wv] = acc[m][e];
        }
}

// dW[co][ci][s] = sum over the workgroups of (co tile, ci chunk): 16 lanes per output stride over the partials
// (fixed order), row-local DPP sum at the end
__global__ __launch_bounds__(256) void conv3x3_wgrad_sum_kernel(const float *__restrict__ ws, float *__restrict__ dW, int Cin,
                                                                int Cout, int n_cic, int wg_per_cc) {
    const int sub = threadIdx.x & 15;
    const long n = (long)Cout * Cin * 9;
    long i = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = i < n;
    i = live ? i : n - 1;
    const int s = (int)(i % 9);
    const long r = i / 9;
    const int ci = (int)(r % Cin), co = (int)(r / Cin);
    const int cc = (co / CO) * n_cic + ci / CI;
    const float *src = ws + ((long)cc * wg_per_cc) * (CO * CI * 9) + ((co % CO) * CI + ci % CI) * 9 + s;
    float a = 0.f;
    for (int k = sub; k < wg_per_cc; k += 16) a += src[(long)k * (CO * CI * 9)];
    a += dpp_mov<MMU_DPP_ROW_SHR(1), 0xf>(0.f, a);
    a += dpp_mov<MMU_DPP_ROW_SHR(2), 0xf>(0.f, a);
    a += dpp_mov<MMU_DPP_ROW_SHR(4), 0xf>(0.f, a);
    a += dpp_mov<MMU_DPP_ROW_SHR(8), 0xf>(0.f, a);
    if (live && sub == 15) dW[i] = a;
}

int n_cu_cached() { return mmu_cu_count(); }

int wg_per_cc_for(int batch, int cin, int cout, int h, int w) {
    const int ncc = (cin / CI) * (cout / CO);
    const long ntiles = (long)batch * ((h + TH - 1) / TH) * ((w + TW - 1) / TW);
    long per = n_cu_cached() / ncc;
    per = per < 1 ? 1 : per;
    return (int)(per > ntiles ? ntiles : per);
}

}  // namespace

extern "C" size_t mmu_conv3x3_wgrad_mfma_workspace_floats(int batch, int in_channels, int out_channels, int height,
                                                          int width) {
    if (batch <= 0 || in_channels <= 0 || out_channels <= 0 || height <= 0 || width <= 0 || in_channels % CI ||
        out_channels % CO)
        return 0;
    return (size_t)wg_per_cc_for(batch, in_channels, out_channels, height, width) * (in_channels / CI) *
           (out_channels / CO) * (CO * CI * 9);
}

extern "C" int mmu_conv3x3_wgrad_mfma(const mmu_conv3x3_mfma_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "conv3x3_wgrad_mfma: null params");
    MMU_CHECK(p->batch > 0 && p->height > 0 && p->width > 0, "conv3x3_wgrad_mfma: empty tensor");
    MMU_CHECK(p->in_channels > 0 && p->in_channels % CI == 0 && p->out_channels > 0 && p->out_channels % CO == 0,
              "conv3x3_wgrad_mfma: in_channels must be a multiple of 32 and out_channels of 64 (got %d, %d)",
              p->in_channels, p->out_channels);
    MMU_CHECK(p->width % 4 == 0, "conv3x3_wgrad_mfma: width must be a multiple of 4 (got %d)", p->width);
    // input = x [B, Cin, H, W]; weight field = dout [B, Cout, H, W]; out = dW [Cout, Cin, 3, 3]
    MMU_CHECK(p->input && p->weight && p->out && p->workspace, "conv3x3_wgrad_mfma: input, dout, dweight, workspace required");
    MMU_CHECK(p->io_dtype == MMU_DTYPE_F32 || p->io_dtype == MMU_DTYPE_BF16, "conv3x3_wgrad_mfma: io_dtype must be float32 or bfloat16");
    const bool xb = p->io_dtype == MMU_DTYPE_BF16;
    MMU_CHECK(((uintptr_t)p->weight & (xb ? 7 : 15)) == 0, "conv3x3_wgrad_mfma: dout must be %d-byte aligned", xb ? 8 : 16);
    hipStream_t st = (hipStream_t)stream;
    static unsigned long long attr_mask = 0, attr_mask_xb = 0;  // per device
    if (hipError_t e = xb ? mmu_set_lds_once(conv3x3_wgrad_mfma_kernel<true>, LDS_BYTES, attr_mask_xb)
                          : mmu_set_lds_once(conv3x3_wgrad_mfma_kernel<false>, LDS_BYTES, attr_mask);
        e != hipSuccess)
        return mmu_fail("conv3x3_wgrad_mfma: LDS attribute: %s", hipGetErrorString(e));
    WgArgs a;
    a.x = p->input; a.g = p->weight; a.ws = (float *)p->workspace;
    a.B = p->batch; a.Cin = p->in_channels; a.Cout = p->out_channels; a.H = p->height; a.W = p->width;
    a.tiles_x = (p->width + TW - 1) / TW; a.tiles_y = (p->height + TH - 1) / TH;
    a.n_cic = p->in_channels / CI; a.n_cot = p->out_channels / CO;
    a.wg_per_cc = wg_per_cc_for(p->batch, p->in_channels, p->out_channels, p->height, p->width);
    const int grid = a.wg_per_cc * a.n_cic * a.n_cot;
    if (xb)
        conv3x3_wgrad_mfma_kernel<true><<<grid, NT, LDS_BYTES, st>>>(a);
    else
        conv3x3_wgrad_mfma_kernel<false><<<grid, NT, LDS_BYTES, st>>>(a);
    MMU_HIP_LAUNCH_CHECK("conv3x3_wgrad_mfma");
    const long n = (long)p->out_channels * p->in_channels * 9;
    {   // inside a deferred scope: with the other weight-gradient sums of the pass (deferred_reduce.hip, kind 6)
        const long job[8] = {6, (long)a.ws, (long)p->out, (long)p->in_channels | ((long)p->out_channels << 32), 0, a.n_cic,
                             a.wg_per_cc, (long)CO | ((long)CI << 32)};
        if (mmu_defer_job(job)) return 0;
    }
    conv3x3_wgrad_sum_kernel<<<(unsigned)((n + 15) / 16), 256, 0, st>>>(a.ws, (float *)p->out, p->in_channels, p->out_channels,
                                                                         a.n_cic, a.wg_per_cc);
    MMU_HIP_LAUNCH_CHECK("conv3x3_wgrad_mfma(sum)");
    return 0;
}
