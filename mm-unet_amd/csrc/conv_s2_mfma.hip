// conv_s2_mfma.hip -- the stride-2 dense convolutions of MM-UNet on the bf16 matrix cores with float32 accuracy:
// implicit GEMM, LDS-staged input patch, hi/lo bf16 split, three MFMAs per product (the scheme of conv3x3_mfma.hip).
//
// Where they sit (src/UM_Net/MMUNet.py): RCG's ConvTranspose2d(64, 64, 4, stride 2, padding 1) and
// Conv2d(64, 64, 4, stride 2, padding 1) around its tri-directional Mamba (:360-375, 15.8 % of the model's conv FLOPs),
// the Conv2d(C, 2C, 3, stride 2, padding 1) that opens every down-sampling ResidualBlock (:439-452).  MIOpen runs them
// as fp32 Winograd / implicit-GEMM kernels with NCHW <-> NHWC transposes around the weight gradients (2.6 ms per
// training step, profiles/r02_bench_graph_replay_summary.txt).
//
// A stride-2 convolution is a stride-1 2 x 2 convolution over the four PHASES of its input:
//     out[co][oy][ox] = sum_{kh, kw, c} w[co][c][kh][kw] in[c][2 oy - 1 + kh][2 ox - 1 + kw]            (padding 1)
//   with  2 oy - 1 + kh = 2 (oy + a) - py,  py in {0, 1}, a in {0, 1}   <=>   kh = 2 a + 1 - py :
//     out[co][oy][ox] = sum_{(py, px), (a, b), c} w[co][c][2a + 1 - py][2b + 1 - px] in[c][2 (oy + a) - py][2 (ox + b) - px]
//   (a tap with kh or kw >= K has weight 0: the 3 x 3 kernel fills 9 of the 16 slots, the 4 x 4 kernel all of them).
// GATHER form: K dimension = 4 phases x Cin "virtual channels", 4 shifts (a, b); the patch of a 16-channel chunk is the
// phase's pixels (real pixel (2 vy - py, 2 vx - px) at patch position (vy, vx)), so a shift is again just a pixel
// offset in LDS.  Serves: Conv2d(.., stride 2) forward, ConvTranspose2d(.., stride 2) input gradient.
//
// The transposed convolution (= the input gradient of the strided one) is four stride-1 2 x 2 convolutions, one per
// OUTPUT phase:
//     out[co][2 vy + qy][2 vx + qx] = sum_{(dy, dx), c} wT[c][co][kh(qy, dy)][kw(qx, dx)] in[c][vy - 1 + qy + dy][vx - 1 + qx + dx]
//   with kh(0, 0) = 3, kh(0, 1) = 1, kh(1, 0) = 2, kh(1, 1) = 0 (a tap >= K has weight 0).
// SCATTER form: the output phase rides in the tile index (virtual output-channel tile = phase x Cout / 64); the patch is
// the plain (8 + 2) x (64 + 2) one, the phase shifts the window by (qy, qx), the store interleaves.  Serves:
// ConvTranspose2d forward, Conv2d(.., stride 2) input gradient.
//
// Tile, wave layout, chunk pipeline, fragment addressing: conv3x3_mfma.hip (8 x 64 pixels x 64 channels per workgroup,
// wave = tile row, 2 x 2 tiles of 32 x 32, persistent workgroups, double-buffered LDS stage, loads two chunks ahead).
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TH = 8, TW = 64, CK = 16, NS = 4;
constexpr int WCH_BYTES = NS * 64 * CK * 2;          // one bf16 image of a chunk's weights: 8,192 B

template <int SCATTER>
struct Geo {
    static constexpr int PH = TH + 1 + SCATTER, PW = TW + 1 + SCATTER, NPX = PH * PW;
    static constexpr int PATCH_BYTES = NPX * CK * 2;
    static constexpr int STAGE_BYTES = 2 * PATCH_BYTES + 2 * WCH_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;      // gather 107,264 B, scatter 117,248 B
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo) {
    hi = pack_bf16(a, b);
    const float ah = __builtin_bit_cast(float, hi << 16), bh = __builtin_bit_cast(float, hi & 0xffff0000u);
    float la, lb;   // two plain v_sub_f32 (the compiler's v_pk_add_f32 costs more beside MFMAs: conv3x3_mfma.hip)
    asm("v_sub_f32 %0, %1, %2" : "=v"(la) : "v"(a), "v"(ah));
    asm("v_sub_f32 %0, %1, %2" : "=v"(lb) : "v"(b), "v"(bh));
    lo = pack_bf16(la, lb);
}

// Weight source wsrc[X][Y][K][K] float32.
//   gather  (SCATTER = 0): out channel = X (Cout), in channel = Y (Cin): conv weight [Cout][Cin][K][K], or a transposed
//                          convolution's [Cin_T][Cout_T][K][K] read for its input gradient (X = its Cin_T).
//                          K chunks run over the 4 Cin virtual channels v = phase * Cin + c.
//   scatter (SCATTER = 1): contraction channel = X, out channel = Y: ConvTranspose2d weight [Cin][Cout][K][K], or a
//                          convolution's [Cout][Cin][K][K] read for its input gradient (X = its Cout).
//                          Virtual output tiles cot' = phase * (Y / 64) + cot.
// -> [cot'][chunk][hi|lo][4 shifts][ci half][64 co][8 ci] bf16
template <int SCATTER>
__global__ __launch_bounds__(256) void conv_s2_prep_kernel(const float *__restrict__ w, unsigned short *__restrict__ out,
                                                           int X, int Y, int K) {
    const int Cin = SCATTER ? X : Y, Cout = SCATTER ? Y : X;   // of THIS operation
    const int nch = (SCATTER ? Cin : 4 * Cin) / CK, ncotv = (SCATTER ? 4 : 1) * (Cout / 64);
    const long n = (long)ncotv * nch * NS * 64 * CK;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int k = (int)(idx & 15), co = (int)((idx >> 4) & 63);
    long r = idx >> 10;
    const int s = (int)(r & 3);
    r >>= 2;
    const int ch = (int)(r % nch), cotv = (int)(r / nch);
    const int dy = s >> 1, dx = s & 1;
    float v = 0.f;
    if (SCATTER) {
        const int q = cotv / (Cout / 64), cot = cotv - q * (Cout / 64);
        const int qy = q >> 1, qx = q & 1;
        const int kh = qy ? (dy ? 0 : 2) : (dy ? 1 : 3), kw = qx ? (dx ? 0 : 2) : (dx ? 1 : 3);
        const int c = ch * CK + k, oc = cot * 64 + co;
        if (kh < K && kw < K) v = w[(((long)c * Cout + oc) * K + kh) * K + kw];
    } else {
        const int vc = ch * CK + k, ph = vc / Cin, c = vc - ph * Cin;
        const int py = ph >> 1, px = ph & 1;
        const int kh = 2 * dy + 1 - py, kw = 2 * dx + 1 - px;
        const int oc = cotv * 64 + co;
        if (kh < K && kw < K) v = w[(((long)oc * Cin + c) * K + kh) * K + kw];
    }
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    const long base = ((long)(cotv * nch + ch) * 2) * (NS * 64 * 16) + (((long)s * 2 + (k >> 3)) * 64 + co) * 8 + (k & 7);
    out[base] = __builtin_bit_cast(unsigned short, h);
    out[base + NS * 64 * 16] = __builtin_bit_cast(unsigned short, l);
}

struct S2Args {
    const void *x;            // float32 (bfloat16 in the XB instantiations of the producer / consumer kernel)
    const unsigned short *wp;
    const float *bias;
    void *out;                // same type as x
    // Cin / Cout of this operation; Hi x Wi = input map, Ho x Wo = output map; tiles over the Ho x Wo map (gather) or
    // over the Hi x Wi map (scatter: a tile covers one output phase of 8 x 64 positions)
    int B, Cin, Cout, Hi, Wi, Ho, Wo, tiles_x, tiles_y, ncot, ncotv, nch, total_tiles;
};

template <int SCATTER>
__global__ __launch_bounds__(512, 2) void conv_s2_mfma_kernel(S2Args p) {
    using G = Geo<SCATTER>;
    constexpr int PW = G::PW, NPX = G::NPX, PATCH_BYTES = G::PATCH_BYTES, STAGE_BYTES = G::STAGE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nch = p.nch;
    const long HWi = (long)p.Hi * p.Wi, HWo = (long)p.Ho * p.Wo;
    const int GD = gridDim.x;
    const int ntl = (p.total_tiles - (int)blockIdx.x + GD - 1) / GD;   // tiles of this workgroup (>= 1)
    const int niter = ntl * nch;
    const int nchp = p.Cin / CK;                                       // chunks per phase (gather)

    // patch work items of this thread: (channel half, patch pixel), pixel fastest over the lanes
    constexpr int NITEM = 2 * NPX;
    static_assert(NITEM <= 3 * 512, "three staging rounds");
    int ipr[3], ipc[3], ihalf[3], ioff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int q = tid + 512 * k;
        const bool live = q < NITEM;
        q = live ? q : NITEM - 1;
        ihalf[k] = q / NPX;
        const int pxi = q - ihalf[k] * NPX;
        ipr[k] = pxi / PW;
        ipc[k] = pxi - ipr[k] * PW;
        ioff[k] = live ? ihalf[k] * (NPX * 16) + pxi * 16 : -1;
    }
    auto decode = [&](int tj, int &b, int &cotv, int &y0, int &x0) {
        int t = (int)blockIdx.x + tj * GD;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        t /= p.tiles_y;
        cotv = t % p.ncotv;
        b = t / p.ncotv;
        y0 = ty * TH;
        x0 = tx * TW;
    };

    // ---- load stream (two chunks ahead of the MFMAs)
    float px[3][8], pmask[3];
    v4u wr[2];
    int l_tj = 0, l_ch = 0, l_b, l_cotv, l_y0, l_x0;
    decode(0, l_b, l_cotv, l_y0, l_x0);
    auto prefetch = [&]() {
        int py = 0, pxx = 0, cbase = l_ch * CK;
        if (!SCATTER) {   // chunk -> (phase, channel base)
            const int ph = l_ch / nchp;
            cbase = (l_ch - ph * nchp) * CK;
            py = ph >> 1;
            pxx = ph & 1;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int gy = SCATTER ? l_y0 - 1 + ipr[k] : 2 * (l_y0 + ipr[k]) - py;
            const int gx = SCATTER ? l_x0 - 1 + ipc[k] : 2 * (l_x0 + ipc[k]) - pxx;
            const bool inb = gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
            pmask[k] = inb ? 1.f : 0.f;
            const float *s = (const float *)p.x + ((long)l_b * p.Cin + cbase + 8 * ihalf[k]) * HWi + (inb ? (long)gy * p.Wi + gx : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) px[k][j] = s[j * HWi];
        }
        const v4u *ws = reinterpret_cast<const v4u *>(p.wp + ((long)l_cotv * nch + l_ch) * (2 * NS * 64 * 16));
#pragma unroll
        for (int j = 0; j < 2; ++j) wr[j] = ws[tid + 512 * j];       // 2 * 8,192 B = 1,024 x 16 B
        if (l_ch + 1 < nch) {
            ++l_ch;
        } else if (l_tj + 1 < ntl) {
            l_ch = 0;
            ++l_tj;
            decode(l_tj, l_b, l_cotv, l_y0, l_x0);
        }
    };
    auto stage = [&](unsigned char *buf) {
        unsigned char *patch_hi = buf, *patch_lo = buf + PATCH_BYTES, *w_hi = buf + 2 * PATCH_BYTES;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            unsigned hw[4], lw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                split2(pmask[k] != 0.f ? px[k][2 * j] : 0.f, pmask[k] != 0.f ? px[k][2 * j + 1] : 0.f, hw[j], lw[j]);
            const v4u h = {hw[0], hw[1], hw[2], hw[3]}, l = {lw[0], lw[1], lw[2], lw[3]};
            if (ioff[k] >= 0) {
                *reinterpret_cast<v4u *>(patch_hi + ioff[k]) = h;
                *reinterpret_cast<v4u *>(patch_lo + ioff[k]) = l;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<v4u *>(w_hi + (tid + 512 * j) * 16) = wr[j];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

    const int a_lane = (lane >> 5) * (64 * 16) + (lane & 31) * 16;        // weights: [shift][plane][64 co][8 ci]
    const int b_lane = (lane >> 5) * (NPX * 16) + (lane & 31) * 16;       // patch:   [plane][pixel][8 ci]

    prefetch();
    stage(lds);
    prefetch();
    MMU_LDS_BARRIER();
    int c_tj = 0, c_ch = 0;
    int c_b, c_cotv, c_y0, c_x0;
    decode(0, c_b, c_cotv, c_y0, c_x0);
    for (int it = 0; it < niter; ++it) {
        const unsigned char *cur = lds + (it & 1) * STAGE_BYTES;
        const unsigned char *patch_hi = cur, *patch_lo = cur + PATCH_BYTES, *w_hi = cur + 2 * PATCH_BYTES;
        stage(lds + ((it + 1) & 1) * STAGE_BYTES);
        prefetch();
        __builtin_amdgcn_sched_barrier(0);
        // scatter: the tile's output phase moves the 2 x 2 window inside the (8 + 2) x (64 + 2) patch
        const int q = SCATTER ? c_cotv / p.ncot : 0;
        const int qy = q >> 1, qx = q & 1;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int dy = s >> 1, dx = s & 1;
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int off = s * (2 * 64 * 16) + m * (32 * 16) + a_lane;
                ah[m] = *reinterpret_cast<const bf16x8 *>(w_hi + off);
                al[m] = *reinterpret_cast<const bf16x8 *>(w_hi + WCH_BYTES + off);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int off = ((wv + dy + qy) * PW + n * 32 + dx + qx) * 16 + b_lane;
                bh[n] = *reinterpret_cast<const bf16x8 *>(patch_hi + off);
                bl[n] = *reinterpret_cast<const bf16x8 *>(patch_lo + off);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
                }
        }
        if (++c_ch == nch) {
            // tile done: C layout col = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel)
            const int cot = SCATTER ? c_cotv - q * p.ncot : c_cotv;
            const int vy = c_y0 + wv;
            float bv[2][16];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 16; ++e) bv[m][e] = 0.f;
            if (p.bias != nullptr) {
                const float *bp = p.bias + cot * 64 + 4 * (lane >> 5);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int e = 0; e < 16; ++e) bv[m][e] = bp[m * 32 + (e & 3) + 8 * (e >> 2)];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int vx = c_x0 + n * 32 + (lane & 31);
                    const int oy = SCATTER ? 2 * vy + qy : vy, ox = SCATTER ? 2 * vx + qx : vx;
                    const bool ok = SCATTER ? (vy < p.Hi && vx < p.Wi) : (vy < p.Ho && vx < p.Wo);
                    if (ok) {
                        float *op = (float *)p.out + ((long)c_b * p.Cout + cot * 64 + m * 32 + 4 * (lane >> 5)) * HWo + (long)oy * p.Wo + ox;
#pragma unroll
                        for (int e = 0; e < 16; ++e) op[((e & 3) + 8 * (e >> 2)) * HWo] = acc[m][n][e] + bv[m][e];
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
                }
            c_ch = 0;
            ++c_tj;
            if (c_tj < ntl) decode(c_tj, c_b, c_cotv, c_y0, c_x0);
        }
        MMU_LDS_BARRIER();
    }
}

// ---- the same two forms with the staging and the matrix work on DIFFERENT waves (round 4; conv3x3_mfma.hip has the
// measurements behind it: with all eight waves staging and then multiplying, the two phases of a chunk ADD).  A chunk here
// has only 4 shifts (48 MFMAs per wave against the same ~340 staging instructions as the 3 x 3 kernel's 108), so the
// staging was more than half of the kernel.  Waves 0-3: producers (buffer loads with hardware zero padding two chunks
// ahead, hi/lo split, LDS stores of the next chunk); waves 4-7: consumers, two tile rows each, the next (shift, row) step's
// fragments read under the current MFMAs; buffer stores in the epilogue.
// XB: bfloat16 activations (autocast): a bf16 value is its own hi part -- no lo image of the patch, two MFMAs per product
// (W lo x X, W hi x X), bf16 stores (conv3x3_mfma.hip's XB form).
template <int SCATTER, bool XB>
__global__ __launch_bounds__(512, 2) void conv_s2_mfma_ws_kernel(S2Args p) {
    constexpr unsigned ES = XB ? 2 : 4;
    using G = Geo<SCATTER>;
    constexpr int PW = G::PW, NPX = G::NPX, PATCH_BYTES = G::PATCH_BYTES, STAGE_BYTES = G::STAGE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nch = p.nch;
    const long HWi = (long)p.Hi * p.Wi, HWo = (long)p.Ho * p.Wo;
    const int GD = gridDim.x;
    const int ntl = (p.total_tiles - (int)blockIdx.x + GD - 1) / GD;   // tiles of this workgroup (>= 1)
    const int niter = ntl * nch;
    const int nchp = p.Cin / CK;                                       // chunks per phase (gather)
    constexpr unsigned OOB = 0x80000000u;
    auto decode = [&](int tj, int &b, int &cotv, int &y0, int &x0) {
        int t = (int)blockIdx.x + tj * GD;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        t /= p.tiles_y;
        cotv = t % p.ncotv;
        b = t / p.ncotv;
        y0 = ty * TH;
        x0 = tx * TW;
    };
    if (wv < 4) {
        // ================= producers
        constexpr int NITEM = 2 * NPX, PR = (NITEM + 255) / 256, WR = 4;   // 2 x 8,192 B of weights = 1,024 x 16 B
        int ipr[PR], ipc[PR], ihalf[PR], ioff[PR];
#pragma unroll
        for (int k = 0; k < PR; ++k) {
            int q = tid + 256 * k;
            const bool live = q < NITEM;
            q = live ? q : NITEM - 1;
            ihalf[k] = q / NPX;
            const int pxi = q - ihalf[k] * NPX;
            ipr[k] = pxi / PW;
            ipc[k] = pxi - ipr[k] * PW;
            ioff[k] = live ? ihalf[k] * (NPX * 16) + pxi * 16 : -1;
        }
        unsigned px[PR][8];
        v4u wr[WR];
        int l_tj = 0, l_ch = 0, l_b, l_cotv, l_y0, l_x0;
        decode(0, l_b, l_cotv, l_y0, l_x0);
        auto prefetch = [&]() {
            int py = 0, pxx = 0, cbase = l_ch * CK;
            if (!SCATTER) {   // chunk -> (phase, channel base)
                const int ph = l_ch / nchp;
                cbase = (l_ch - ph * nchp) * CK;
                py = ph >> 1;
                pxx = ph & 1;
            }
            const char *base = (const char *)p.x + ((long)l_b * p.Cin + cbase) * HWi * ES;
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, (int)OOB, 0x00020000);
#pragma unroll
            for (int k = 0; k < PR; ++k) {
                const int gy = SCATTER ? l_y0 - 1 + ipr[k] : 2 * (l_y0 + ipr[k]) - py;
                const int gx = SCATTER ? l_x0 - 1 + ipc[k] : 2 * (l_x0 + ipc[k]) - pxx;
                const bool inb = gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
                const unsigned voff = inb ? ((unsigned)(8 * ihalf[k]) * (unsigned)HWi + (unsigned)(gy * p.Wi + gx)) * ES : OOB;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (XB)
                        px[k][j] = __builtin_amdgcn_raw_buffer_load_b16(rs, voff, (unsigned)(j * HWi) * ES, 0);
                    else
                        px[k][j] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (unsigned)(j * HWi) * ES, 0);
                }
            }
            const v4u *ws = reinterpret_cast<const v4u *>(p.wp + ((long)l_cotv * nch + l_ch) * (2 * NS * 64 * 16));
#pragma unroll
            for (int j = 0; j < WR; ++j) wr[j] = ws[tid + 256 * j];
            if (l_ch + 1 < nch) {
                ++l_ch;
            } else if (l_tj + 1 < ntl) {
                l_ch = 0;
                ++l_tj;
                decode(l_tj, l_b, l_cotv, l_y0, l_x0);
            }
        };
        auto stage = [&](unsigned char *buf) {
            unsigned char *patch_hi = buf, *patch_lo = buf + PATCH_BYTES, *w_hi = buf + 2 * PATCH_BYTES;
#pragma unroll
            for (int k = 0; k < PR; ++k) {
                unsigned hw[4], lw[4];
                if constexpr (XB) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) hw[j] = px[k][2 * j] | (px[k][2 * j + 1] << 16);
                    if (ioff[k] >= 0) *reinterpret_cast<v4u *>(patch_hi + ioff[k]) = v4u{hw[0], hw[1], hw[2], hw[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) split2(__uint_as_float(px[k][2 * j]), __uint_as_float(px[k][2 * j + 1]), hw[j], lw[j]);
                    const v4u h = {hw[0], hw[1], hw[2], hw[3]}, l = {lw[0], lw[1], lw[2], lw[3]};
                    if (ioff[k] >= 0) {
                        *reinterpret_cast<v4u *>(patch_hi + ioff[k]) = h;
                        *reinterpret_cast<v4u *>(patch_lo + ioff[k]) = l;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < WR; ++j) *reinterpret_cast<v4u *>(w_hi + (tid + 256 * j) * 16) = wr[j];
        };
        prefetch();
        stage(lds);
        prefetch();
        MMU_LDS_BARRIER();
        for (int it = 0; it < niter; ++it) {
            stage(lds + ((it + 1) & 1) * STAGE_BYTES);
            prefetch();
            MMU_LDS_BARRIER();
        }
    } else {
        // ================= consumers: wave cw owns tile rows 2 cw, 2 cw + 1
        const int cw = wv - 4;
        f32x16 acc[2][2][2];   // [row][m][n]
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[r][m][n][e] = 0.f;
        const int a_lane = (lane >> 5) * (64 * 16) + (lane & 31) * 16;        // weights: [shift][plane][64 co][8 ci]
        const int b_lane = (lane >> 5) * (NPX * 16) + (lane & 31) * 16;       // patch:   [plane][pixel][8 ci]
        MMU_LDS_BARRIER();
        int c_tj = 0, c_ch = 0;
        int c_b, c_cotv, c_y0, c_x0;
        decode(0, c_b, c_cotv, c_y0, c_x0);
        for (int it = 0; it < niter; ++it) {
            const unsigned char *cur = lds + (it & 1) * STAGE_BYTES;
            const unsigned char *patch_hi = cur, *patch_lo = cur + PATCH_BYTES, *w_hi = cur + 2 * PATCH_BYTES;
            // scatter: the tile's output phase moves the 2 x 2 window inside the (8 + 2) x (64 + 2) patch
            const int q = SCATTER ? c_cotv / p.ncot : 0;
            const int qy = q >> 1, qx = q & 1;
            bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];   // [register set][m or n]
            auto frag_a = [&](int s, int set) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int off = s * (2 * 64 * 16) + m * (32 * 16) + a_lane;
                    ah[set][m] = *reinterpret_cast<const bf16x8 *>(w_hi + off);
                    al[set][m] = *reinterpret_cast<const bf16x8 *>(w_hi + WCH_BYTES + off);
                }
            };
            auto frag_b = [&](int s, int r, int set) {
                const int dy = s >> 1, dx = s & 1;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int off = ((2 * cw + r + dy + qy) * PW + n * 32 + dx + qx) * 16 + b_lane;
                    bh[set][n] = *reinterpret_cast<const bf16x8 *>(patch_hi + off);
                    if constexpr (!XB) bl[set][n] = *reinterpret_cast<const bf16x8 *>(patch_lo + off);
                }
            };
            frag_a(0, 0);
            frag_b(0, 0, 0);
#pragma unroll
            for (int step = 0; step < 2 * NS; ++step) {      // step = (shift, row)
                const int s = step >> 1, r = step & 1, ca = s & 1, cb = step & 1;
                if (step + 1 < 2 * NS) {
                    const int ns = (step + 1) >> 1, nr = (step + 1) & 1;
                    frag_b(ns, nr, cb ^ 1);
                    if (nr == 0) frag_a(ns, ca ^ 1);
                    __builtin_amdgcn_sched_barrier(0);   // the next step's reads are issued before this step's MFMAs
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        acc[r][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ca][m], bh[cb][n], acc[r][m][n], 0, 0, 0);
                        if constexpr (!XB)
                            acc[r][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ca][m], bl[cb][n], acc[r][m][n], 0, 0, 0);
                        acc[r][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ca][m], bh[cb][n], acc[r][m][n], 0, 0, 0);
                    }
                if (step + 1 < 2 * NS) __builtin_amdgcn_sched_barrier(0);
            }
            if (++c_ch == nch) {
                // tile done: C layout col = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel)
                const int cot = SCATTER ? c_cotv - q * p.ncot : c_cotv;
                float bv[2][16];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int e = 0; e < 16; ++e) bv[m][e] = 0.f;
                if (p.bias != nullptr) {
                    const float *bp = p.bias + cot * 64 + 4 * (lane >> 5);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int e = 0; e < 16; ++e) bv[m][e] = bp[m * 32 + (e & 3) + 8 * (e >> 2)];
                }
                char *obase = (char *)p.out + ((long)c_b * p.Cout + cot * 64) * HWo * ES;
                const rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)OOB, 0x00020000);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int vy = c_y0 + 2 * cw + r;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int vx = c_x0 + n * 32 + (lane & 31);
                        const int oy = SCATTER ? 2 * vy + qy : vy, ox = SCATTER ? 2 * vx + qx : vx;
                        const bool ok = SCATTER ? (vy < p.Hi && vx < p.Wi) : (vy < p.Ho && vx < p.Wo);
                        const unsigned voff = ok ? ((unsigned)(4 * (lane >> 5)) * (unsigned)HWo + (unsigned)(oy * p.Wo + ox)) * ES : OOB;
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const unsigned soff = (unsigned)((m * 32 + (e & 3) + 8 * (e >> 2)) * HWo) * ES;
                                const float v = acc[r][m][n][e] + bv[m][e];
                                if constexpr (XB)
                                    __builtin_amdgcn_raw_buffer_store_b16(from_f32<bf16_t>(v).bits, ors, voff, soff, 0);
                                else
                                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ors, voff, soff, 0);
                                acc[r][m][n][e] = 0.f;
                            }
                    }
                }
                c_ch = 0;
                ++c_tj;
                if (c_tj < ntl) decode(c_tj, c_b, c_cotv, c_y0, c_x0);
            }
            MMU_LDS_BARRIER();
        }
    }
}

int check(const mmu_conv_s2_params *p, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->batch > 0 && p->in_height > 0 && p->in_width > 0, "%s: empty tensor", name);
    MMU_CHECK(p->kernel == 3 || p->kernel == 4, "%s: kernel size 3 or 4 (stride 2, padding 1) supported (got %d)", name,
              p->kernel);
    MMU_CHECK(p->in_channels > 0 && p->in_channels % 16 == 0 && p->out_channels > 0 && p->out_channels % 64 == 0,
              "%s: in_channels must be a multiple of 16 and out_channels of 64 (got %d, %d)", name, p->in_channels,
              p->out_channels);
    MMU_CHECK(p->input && p->weight && p->out && p->workspace, "%s: input, weight, out, workspace are required", name);
    MMU_CHECK(((uintptr_t)p->workspace & 15) == 0, "%s: workspace must be 16-byte aligned", name);
    return 0;
}

template <int SCATTER>
int launch(const mmu_conv_s2_params *p, hipStream_t st, const char *name) {
    using G = Geo<SCATTER>;
    const int K = p->kernel;
    S2Args a;
    a.x = p->input; a.wp = (const unsigned short *)p->workspace; a.bias = p->bias; a.out = p->out;
    a.B = p->batch; a.Cin = p->in_channels; a.Cout = p->out_channels; a.Hi = p->in_height; a.Wi = p->in_width;
    if (SCATTER) {   // transposed: out = (Hi - 1) * 2 - 2 + K + output_padding; the tiles run over the input grid
        a.Ho = p->out_height; a.Wo = p->out_width;
        MMU_CHECK(a.Ho >= 2 * a.Hi - 1 && a.Ho <= 2 * a.Hi && a.Wo >= 2 * a.Wi - 1 && a.Wo <= 2 * a.Wi,
                  "%s: output %d x %d does not belong to a stride-2 / padding-1 transposed convolution of %d x %d", name,
                  a.Ho, a.Wo, a.Hi, a.Wi);
        MMU_CHECK(a.Ho == 2 * a.Hi && a.Wo == 2 * a.Wi, "%s: output must be exactly twice the input (got %d x %d from %d x %d)",
                  name, a.Ho, a.Wo, a.Hi, a.Wi);
        a.tiles_x = (a.Wi + TW - 1) / TW; a.tiles_y = (a.Hi + TH - 1) / TH;
        a.nch = a.Cin / CK;
    } else {
        a.Ho = (a.Hi + 2 - K) / 2 + 1; a.Wo = (a.Wi + 2 - K) / 2 + 1;
        MMU_CHECK(p->out_height == a.Ho && p->out_width == a.Wo, "%s: output must be %d x %d (got %d x %d)", name, a.Ho,
                  a.Wo, p->out_height, p->out_width);
        a.tiles_x = (a.Wo + TW - 1) / TW; a.tiles_y = (a.Ho + TH - 1) / TH;
        a.nch = 4 * a.Cin / CK;
    }
    a.ncot = a.Cout / 64;
    a.ncotv = (SCATTER ? 4 : 1) * a.ncot;
    const long total = (long)a.tiles_x * a.tiles_y * a.ncotv * a.B;
    MMU_CHECK(total < (1L << 30), "%s: too many tiles", name);
    a.total_tiles = (int)total;
    // X, Y of the weight source: gather reads [Cout][Cin], scatter [Cin][Cout]
    const int X = SCATTER ? a.Cin : a.Cout, Y = SCATTER ? a.Cout : a.Cin;
    const long nw = (long)a.ncotv * a.nch * NS * 64 * CK;
    conv_s2_prep_kernel<SCATTER><<<(unsigned)((nw + 255) / 256), 256, 0, st>>>((const float *)p->weight, (unsigned short *)p->workspace, X,
                                                                               Y, K);
    MMU_HIP_LAUNCH_CHECK(name);
    // MMU_CONV_S2_WS=0: the kernel whose eight waves all stage and multiply (A/B); the producer / consumer kernel's buffer
    // addressing keeps 9 input rows-of-channels and a 64-channel output tile of one batch item within 32-bit byte offsets
    static const bool ws_on = []() { const char *e = getenv("MMU_CONV_S2_WS"); return !e || e[0] != '0'; }();
    const bool fits = 9L * a.Hi * a.Wi * 4 < (1L << 31) && 64L * a.Ho * a.Wo * 4 < (1L << 31);
    const bool xb = p->io_dtype == MMU_DTYPE_BF16;
    MMU_CHECK(xb || p->io_dtype == MMU_DTYPE_F32, "%s: io_dtype must be float32 or bfloat16 (got %d)", name, p->io_dtype);
    MMU_CHECK(!xb || fits, "%s: bfloat16 activations need maps within 32-bit byte offsets", name);
    const bool ws = xb || (ws_on && fits);
    static unsigned long long attr_mask = 0, attr_mask_ws = 0, attr_mask_xb = 0;  // per device
    hipError_t e;
    if (xb)
        e = mmu_set_lds_once(conv_s2_mfma_ws_kernel<SCATTER, true>, G::LDS_BYTES, attr_mask_xb);
    else if (ws)
        e = mmu_set_lds_once(conv_s2_mfma_ws_kernel<SCATTER, false>, G::LDS_BYTES, attr_mask_ws);
    else
        e = mmu_set_lds_once(conv_s2_mfma_kernel<SCATTER>, G::LDS_BYTES, attr_mask);
    if (e != hipSuccess) return mmu_fail("%s: LDS attribute: %s", name, hipGetErrorString(e));
    const int n_cu = mmu_cu_count();
    const int grid = total < n_cu ? (int)total : n_cu;
    if (xb)
        conv_s2_mfma_ws_kernel<SCATTER, true><<<grid, 512, G::LDS_BYTES, st>>>(a);
    else if (ws)
        conv_s2_mfma_ws_kernel<SCATTER, false><<<grid, 512, G::LDS_BYTES, st>>>(a);
    else
        conv_s2_mfma_kernel<SCATTER><<<grid, 512, G::LDS_BYTES, st>>>(a);
    MMU_HIP_LAUNCH_CHECK(name);
    return 0;
}

}  // namespace

// bytes of the prepared-weight workspace: 16 taps' worth of hi + lo bf16 images whatever the kernel size
extern "C" size_t mmu_conv_s2_workspace_bytes(int in_channels, int out_channels) {
    if (in_channels <= 0 || out_channels <= 0) return 0;
    return (size_t)in_channels * out_channels * 16 * 2 * sizeof(unsigned short);
}

// out = conv2d(input, weight [Cout][Cin][K][K], stride 2, padding 1) (+ bias); with weight = a transposed convolution's
// [Cin_T][Cout_T][K][K] and input = its output gradient this is that convolution's input gradient.
extern "C" int mmu_conv_s2_mfma(const mmu_conv_s2_params *p, void *stream) {
    if (int r = check(p, "conv_s2_mfma")) return r;
    return launch<0>(p, (hipStream_t)stream, "conv_s2_mfma");
}

// out = conv_transpose2d(input, weight [Cin][Cout][K][K], stride 2, padding 1) with out = 2 x input size (K = 4: no output
// padding; K = 3: output_padding 1) (+ bias); with weight = a strided convolution's [Cout_c][Cin_c][K][K] and input = its
// output gradient this is that convolution's input gradient.
extern "C" int mmu_conv_s2_transposed_mfma(const mmu_conv_s2_params *p, void *stream) {
    if (int r = check(p, "conv_s2_transposed_mfma")) return r;
    return launch<1>(p, (hipStream_t)stream, "conv_s2_transposed_mfma");
}

// ================================================================================================================
// weight gradient of the stride-2 convolution (and, operands swapped, of the transposed one)
// ================================================================================================================
//   dW[co][c][kh][kw] = sum_{b, oy, ox} dout[b][co][oy][ox] * in[b][c][2 oy - 1 + kh][2 ox - 1 + kw]
// In the phase form of the header: per virtual channel v = (py, px, c) and shift (a, b)
//   dW'[co][v][a][b] = sum dout[co][o] * in[c][2 (oy + a) - py][2 (ox + b) - px] ,   dW[co][c][2a+1-py][2b+1-px] = dW'
// -- the stride-1 weight-gradient GEMM of conv3x3_wgrad_mfma.hip (M = co, N = virtual ci, contraction over output pixels,
// patch operand read with ds_read_b64_tr_b16) with 4 shifts instead of 9 and the patch gathered per phase.
// Workgroup = 8 waves = 4 shifts x 2 halves of a 64-virtual-channel chunk; tile = 4 x 64 output pixels.
// The ConvTranspose2d weight gradient dWT[c][co][kh][kw] = sum in[c][y][x] * dout[co][2y - 1 + kh][2x - 1 + kw] is the
// same sum with the low-resolution tensor in the role of `dout` and the high-resolution one in the role of `in`.
namespace {

typedef __attribute__((ext_vector_type(8))) short s8;
typedef __attribute__((ext_vector_type(4))) short s4;

constexpr int WTH = 4, WTW = 64, WPH = WTH + 1, WPW = WTW + 1, WNPX = WPH * WPW;   // 325 patch positions
constexpr int WCI = 64, WCO = 64, WNT = 512;
constexpr int WROW = WCI * 2;                       // patch row: 64 virtual channels bf16 = 128 B
constexpr int WPATCH_IMG = WNPX * WROW;             // 41,600 B
constexpr int WDROW = WTH * WTW * 2 + 16;           // dout image row: 256 pixels bf16 + pad: 528 B
constexpr int WDOUT_IMG = WCO * WDROW;              // 33,792 B
constexpr int WLDS_BYTES = 2 * WPATCH_IMG + 2 * WDOUT_IMG;   // 150,784 B

struct S2WgArgs {
    const void *x, *g;       // x: high-resolution [B, Cin, Hi, Wi]; g: low-resolution [B, Cout, Ho, Wo]; float32, or (XB)
                             // both bfloat16: exact bf16 operands, ONE MFMA per product
    float *ws;
    int B, Cin, Cout, Hi, Wi, Ho, Wo, tiles_x, tiles_y, n_cic, n_cot, wg_per_cc;
};

template <bool XB>
__global__ __launch_bounds__(WNT, 1) void conv_s2_wgrad_kernel(S2WgArgs p) {
    using px_t = typename std::conditional<XB, unsigned short, float>::type;   // raw activations as loaded
    using dv_t = typename std::conditional<XB, v2u, float4>::type;            // four dout pixels as loaded
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *patch_hi = lds, *patch_lo = lds + WPATCH_IMG;
    unsigned char *dout_hi = lds + 2 * WPATCH_IMG, *dout_lo = dout_hi + WDOUT_IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tap = wv >> 1, half = wv & 1;                          // shift (a, b) and 32-channel half of this wave
    const int cc = blockIdx.x / p.wg_per_cc, wl = blockIdx.x - cc * p.wg_per_cc;
    const int cot = cc / p.n_cic, cic = cc - cot * p.n_cic;         // cic: chunk of 64 VIRTUAL channels
    const int vc0 = cic * WCI, ph = vc0 / p.Cin, c0 = vc0 - ph * p.Cin;   // (Cin % 64 == 0: a chunk lies in one phase)
    const int py = ph >> 1, pxx = ph & 1;
    const long HWi = (long)p.Hi * p.Wi, HWo = (long)p.Ho * p.Wo;
    const int tiles_img = p.tiles_x * p.tiles_y, ntiles = tiles_img * p.B;

    // staging items.  patch: (position, group of 8 ci): 325 x 8 = 2,600 items, 6 rounds.  dout: (co, 4 pixels): 8 rounds.
    constexpr int PR = 6;
    int p_pr[PR], p_pc[PR], p_cg[PR], p_off[PR];
#pragma unroll
    for (int k = 0; k < PR; ++k) {
        int q = tid + WNT * k;
        const bool live = q < WNPX * 8;
        q = live ? q : WNPX * 8 - 1;
        p_cg[k] = q / WNPX;
        const int pxi = q - p_cg[k] * WNPX;
        p_pr[k] = pxi / WPW;
        p_pc[k] = pxi - p_pr[k] * WPW;
        p_off[k] = live ? pxi * WROW + p_cg[k] * 16 : -1;
    }
    int d_co[8], d_px[8], d_off[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int q = tid + WNT * k;                  // 4,096 items exactly
        d_co[k] = q >> 6;
        d_px[k] = (q & 63) * 4;
        d_off[k] = d_co[k] * WDROW + d_px[k] * 2;
    }

    px_t px[PR][8];
    float pm[PR];
    dv_t dv[8];
    auto prefetch = [&](int t) {
        t = t < ntiles ? t : ntiles - 1;
        const int b = t / tiles_img, r = t - b * tiles_img;
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
        const int y0 = ty * WTH, x0 = tx * WTW;
#pragma unroll
        for (int k = 0; k < PR; ++k) {
            const int gy = 2 * (y0 + p_pr[k]) - py, gx = 2 * (x0 + p_pc[k]) - pxx;
            const bool inb = gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
            pm[k] = inb ? 1.f : 0.f;
            const px_t *s = (const px_t *)p.x + ((long)b * p.Cin + c0 + 8 * p_cg[k]) * HWi + (inb ? (long)gy * p.Wi + gx : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) px[k][j] = s[j * HWi];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int gy = y0 + (d_px[k] >> 6), gx = x0 + (d_px[k] & 63);
            const bool inb = gy < p.Ho && gx < p.Wo;   // Wo % 4 == 0: a group of 4 is in or out as a whole
            const long go = ((long)b * p.Cout + cot * WCO + d_co[k]) * HWo + (inb ? (long)gy * p.Wo + gx : 0);
            if constexpr (XB) {
                const v2u v = *reinterpret_cast<const v2u *>((const unsigned short *)p.g + go);
                dv[k] = inb ? v : v2u{0u, 0u};
            } else {
                const float4 v = *reinterpret_cast<const float4 *>((const float *)p.g + go);
                dv[k] = make_float4(inb ? v.x : 0.f, inb ? v.y : 0.f, inb ? v.z : 0.f, inb ? v.w : 0.f);
            }
        }
    };
    // The conversion (hi/lo split: the vector work of the staging) runs on the prefetched registers INSIDE the k-loop of the
    // current tile, beside the MFMAs; between the two barriers only the LDS stores are left.  MMU_S2WG_EARLY=0: the
    // conversion between the barriers, as before (A/B).
#ifndef MMU_S2WG_EARLY
#define MMU_S2WG_EARLY 1
#endif
    // (converted in place: the packed hi / lo words take the registers of the raw values they came from)
    auto convert = [&]() {
#pragma unroll
        for (int k = 0; k < PR; ++k) {
            unsigned hw[4], lw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (XB)
                    hw[j] = pm[k] != 0.f ? (unsigned)px[k][2 * j] | ((unsigned)px[k][2 * j + 1] << 16) : 0u;
                else
                    split2(pm[k] != 0.f ? px[k][2 * j] : 0.f, pm[k] != 0.f ? px[k][2 * j + 1] : 0.f, hw[j], lw[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (XB) {
                    px[k][2 * j] = (px_t)(hw[j] & 0xffffu);
                    px[k][2 * j + 1] = (px_t)(hw[j] >> 16);
                } else {
                    px[k][j] = __uint_as_float(hw[j]);
                    px[k][4 + j] = __uint_as_float(lw[j]);
                }
            }
        }
        if constexpr (!XB) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                unsigned h0, l0, h1, l1;
                split2(dv[k].x, dv[k].y, h0, l0);
                split2(dv[k].z, dv[k].w, h1, l1);
                dv[k] = make_float4(__uint_as_float(h0), __uint_as_float(h1), __uint_as_float(l0), __uint_as_float(l1));
            }
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int k = 0; k < PR; ++k)
            if (p_off[k] >= 0) {
                if constexpr (XB) {
                    unsigned hw[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) hw[j] = (unsigned)px[k][2 * j] | ((unsigned)px[k][2 * j + 1] << 16);
                    *reinterpret_cast<v4u *>(patch_hi + p_off[k]) = v4u{hw[0], hw[1], hw[2], hw[3]};
                } else {
                    *reinterpret_cast<v4u *>(patch_hi + p_off[k]) = v4u{__float_as_uint(px[k][0]), __float_as_uint(px[k][1]),
                                                                        __float_as_uint(px[k][2]), __float_as_uint(px[k][3])};
                    *reinterpret_cast<v4u *>(patch_lo + p_off[k]) = v4u{__float_as_uint(px[k][4]), __float_as_uint(px[k][5]),
                                                                        __float_as_uint(px[k][6]), __float_as_uint(px[k][7])};
                }
            }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if constexpr (XB) {
                *reinterpret_cast<v2u *>(dout_hi + d_off[k]) = dv[k];
            } else {
                *reinterpret_cast<v2u *>(dout_hi + d_off[k]) = v2u{__float_as_uint(dv[k].x), __float_as_uint(dv[k].y)};
                *reinterpret_cast<v2u *>(dout_lo + d_off[k]) = v2u{__float_as_uint(dv[k].z), __float_as_uint(dv[k].w)};
            }
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    const int sa = tap >> 1, sb = tap & 1;
    const int a_lane = (lane & 31) * WDROW + (lane >> 5) * 16;
    // transposed read of the patch (conv3x3_wgrad_mfma.hip): lane 4q + pp of a 16-lane group addresses position row q,
    // channels 4 pp .. 4 pp + 3 of the group's 16-channel block; two reads cover the lane half's 8 positions
    const int li = lane & 15, bq = li >> 2, bp = li & 3;
    const int b_lane = bq * WROW + (half * 32 + ((lane >> 4) & 1) * 16 + 4 * bp) * 2 + (lane >> 5) * 8 * WROW;

    auto ksteps = [&](int k0, int k1) {
#pragma unroll 4
        for (int ks = k0; ks < k1; ++ks) {
            const int row = ks >> 2, xk = (ks & 3) * 16;
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int off = m * 32 * WDROW + (row * WTW + xk) * 2 + a_lane;
                ah[m] = *reinterpret_cast<const bf16x8 *>(dout_hi + off);
                if constexpr (!XB) al[m] = *reinterpret_cast<const bf16x8 *>(dout_lo + off);
            }
            const int poff = ((row + sa) * WPW + xk + sb) * WROW + b_lane;
            auto tr = [&](const unsigned char *img, int o) {
                return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s4 *)(uintptr_t)(unsigned)(uintptr_t)(img + o));
            };
            const s4 h0 = tr(patch_hi, poff), h1 = tr(patch_hi, poff + 4 * WROW);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, s8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]});
            if constexpr (XB) {
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh, acc[m], 0, 0, 0);
            } else {
                const s4 l0 = tr(patch_lo, poff), l1 = tr(patch_lo, poff + 4 * WROW);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, s8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]});
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh, acc[m], 0, 0, 0);
                }
            }
        }
    };
    int t = wl;
    prefetch(t);
    if (MMU_S2WG_EARLY) convert();
    for (; t < ntiles; t += p.wg_per_cc) {
        __syncthreads();
        if (!MMU_S2WG_EARLY) convert();
        store();
        __syncthreads();
        prefetch(t + p.wg_per_cc);
        __builtin_amdgcn_sched_barrier(0);
        if (MMU_S2WG_EARLY) {
            ksteps(0, 8);
            convert();       // (the loads have had half a tile's MFMAs to arrive; the other half runs beside the split.
                             //  Spread over the eight remaining steps instead: 256 VGPRs, spills, no faster)
            ksteps(8, 16);
        } else {
            ksteps(0, 16);
        }
    }
    // partial of this workgroup: ws[wg][co 64][vci 64][4 shifts]
    float *wp = p.ws + (long)blockIdx.x * (WCO * WCI * 4);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), ci = half * 32 + (lane & 31);
            wp[(co * WCI + ci) * 4 + tap] = acc[m][e];
        }
}

// dW[X][Y][kh][kw] from the partials; swap = 0: X = co (the low-resolution tensor's channels), Y = c: conv weight
// [Cout][Cin][K][K]; swap = 1: dW is [c... the same (low, high) order -- a ConvTranspose2d weight [Cin_T][Cout_T] has the
// low-resolution tensor's channels first as well, so no swap is ever needed: kept as a plain sum.
__global__ __launch_bounds__(256) void conv_s2_wgrad_sum_kernel(const float *__restrict__ ws, float *__restrict__ dW, int Cin,
                                                                int Cout, int K, int n_cic, int wg_per_cc) {
    const int sub = threadIdx.x & 15;
    const long n = (long)Cout * Cin * K * K;
    long i = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = i < n;
    i = live ? i : n - 1;
    const int kw = (int)(i % K), kh = (int)((i / K) % K);
    const long r = i / (K * K);
    const int c = (int)(r % Cin), co = (int)(r / Cin);
    // kh = 2a + 1 - py  <=>  py = (kh + 1) & 1, a = (kh - 1 + py) / 2
    const int py = (kh + 1) & 1, a = (kh - 1 + py) >> 1, pxx = (kw + 1) & 1, b = (kw - 1 + pxx) >> 1;
    const int v = (py * 2 + pxx) * Cin + c;
    const int cc = (co / WCO) * n_cic + v / WCI;
    const float *src = ws + ((long)cc * wg_per_cc) * (WCO * WCI * 4) + ((co % WCO) * WCI + v % WCI) * 4 + (a * 2 + b);
    float s = 0.f;
    for (int k = sub; k < wg_per_cc; k += 16) s += src[(long)k * (WCO * WCI * 4)];
    s += dpp_mov<MMU_DPP_ROW_SHR(1), 0xf>(0.f, s);
    s += dpp_mov<MMU_DPP_ROW_SHR(2), 0xf>(0.f, s);
    s += dpp_mov<MMU_DPP_ROW_SHR(4), 0xf>(0.f, s);
    s += dpp_mov<MMU_DPP_ROW_SHR(8), 0xf>(0.f, s);
    if (live && sub == 15) dW[i] = s;
}

int s2_wg_per_cc(int batch, int cin, int cout, int ho, int wo) {
    const int ncc = (4 * cin / WCI) * (cout / WCO);
    const long ntiles = (long)batch * ((ho + WTH - 1) / WTH) * ((wo + WTW - 1) / WTW);
    long per = mmu_cu_count() / ncc;
    per = per < 1 ? 1 : per;
    return (int)(per > ntiles ? ntiles : per);
}

}  // namespace

extern "C" size_t mmu_conv_s2_wgrad_workspace_floats(int batch, int in_channels, int out_channels, int out_height,
                                                     int out_width) {
    if (batch <= 0 || in_channels <= 0 || out_channels <= 0 || out_height <= 0 || out_width <= 0 || in_channels % WCI ||
        out_channels % WCO)
        return 0;
    return (size_t)s2_wg_per_cc(batch, in_channels, out_channels, out_height, out_width) * (4 * in_channels / WCI) *
           (out_channels / WCO) * (WCO * WCI * 4);
}

// dweight [out_channels][in_channels][K][K] of conv2d(x, w, stride 2, padding 1): input = x [B, in_channels, in_height,
// in_width], weight field = dout [B, out_channels, out_height, out_width], out = dweight.  For ConvTranspose2d (weight
// [Cin_T][Cout_T][K][K]): input = its output gradient (the high-resolution tensor, in_channels = Cout_T), weight field =
// its input (low resolution, out_channels = Cin_T).
extern "C" int mmu_conv_s2_wgrad_mfma(const mmu_conv_s2_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "conv_s2_wgrad_mfma: null params");
    MMU_CHECK(p->kernel == 3 || p->kernel == 4, "conv_s2_wgrad_mfma: kernel size 3 or 4 (got %d)", p->kernel);
    MMU_CHECK(p->batch > 0 && p->in_height > 0 && p->in_width > 0, "conv_s2_wgrad_mfma: empty tensor");
    MMU_CHECK(p->in_channels > 0 && p->in_channels % WCI == 0 && p->out_channels > 0 && p->out_channels % WCO == 0,
              "conv_s2_wgrad_mfma: in_channels and out_channels must be multiples of 64 (got %d, %d)", p->in_channels,
              p->out_channels);
    const int Ho = (p->in_height + 2 - p->kernel) / 2 + 1, Wo = (p->in_width + 2 - p->kernel) / 2 + 1;
    MMU_CHECK(p->out_height == Ho && p->out_width == Wo, "conv_s2_wgrad_mfma: dout must be %d x %d (got %d x %d)", Ho, Wo,
              p->out_height, p->out_width);
    MMU_CHECK(Wo % 4 == 0, "conv_s2_wgrad_mfma: the output width must be a multiple of 4 (got %d)", Wo);
    MMU_CHECK(p->input && p->weight && p->out && p->workspace, "conv_s2_wgrad_mfma: input, dout, dweight, workspace required");
    const bool xb = p->io_dtype == MMU_DTYPE_BF16;
    MMU_CHECK(xb || p->io_dtype == MMU_DTYPE_F32, "conv_s2_wgrad_mfma: io_dtype must be float32 or bfloat16 (got %d)", p->io_dtype);
    MMU_CHECK(((uintptr_t)p->weight & (xb ? 7 : 15)) == 0, "conv_s2_wgrad_mfma: dout must be %d-byte aligned", xb ? 8 : 16);
    hipStream_t st = (hipStream_t)stream;
    static unsigned long long attr_mask = 0, attr_mask_xb = 0;  // per device
    if (hipError_t e = xb ? mmu_set_lds_once(conv_s2_wgrad_kernel<true>, WLDS_BYTES, attr_mask_xb)
                          : mmu_set_lds_once(conv_s2_wgrad_kernel<false>, WLDS_BYTES, attr_mask);
        e != hipSuccess)
        return mmu_fail("conv_s2_wgrad_mfma: LDS attribute: %s", hipGetErrorString(e));
    S2WgArgs a;
    a.x = p->input; a.g = p->weight; a.ws = (float *)p->workspace;
    a.B = p->batch; a.Cin = p->in_channels; a.Cout = p->out_channels; a.Hi = p->in_height; a.Wi = p->in_width;
    a.Ho = Ho; a.Wo = Wo;
    a.tiles_x = (Wo + WTW - 1) / WTW; a.tiles_y = (Ho + WTH - 1) / WTH;
    a.n_cic = 4 * p->in_channels / WCI; a.n_cot = p->out_channels / WCO;
    a.wg_per_cc = s2_wg_per_cc(p->batch, p->in_channels, p->out_channels, Ho, Wo);
    const int grid = a.wg_per_cc * a.n_cic * a.n_cot;
    if (xb)
        conv_s2_wgrad_kernel<true><<<grid, WNT, WLDS_BYTES, st>>>(a);
    else
        conv_s2_wgrad_kernel<false><<<grid, WNT, WLDS_BYTES, st>>>(a);
    MMU_HIP_LAUNCH_CHECK("conv_s2_wgrad_mfma");
    const long n = (long)p->out_channels * p->in_channels * p->kernel * p->kernel;
    {   // inside a deferred scope: with the other weight-gradient sums of the pass (deferred_reduce.hip, kind 6)
        const long job[8] = {6, (long)a.ws, (long)p->out, (long)p->in_channels | ((long)p->out_channels << 32), p->kernel,
                             a.n_cic, a.wg_per_cc, (long)WCO | ((long)WCI << 32)};
        if (mmu_defer_job(job)) return 0;
    }
    conv_s2_wgrad_sum_kernel<<<(unsigned)((n + 15) / 16), 256, 0, st>>>(a.ws, (float *)p->out, p->in_channels, p->out_channels,
                                                                         p->kernel, a.n_cic, a.wg_per_cc);
    MMU_HIP_LAUNCH_CHECK("conv_s2_wgrad_mfma(sum)");
    return 0;
}
