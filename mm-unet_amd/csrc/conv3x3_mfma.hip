// conv3x3_mfma.hip -- dense 3x3 / stride 1 / padding 1 convolution (Cin % 16 == 0, Cout % 64 == 0) as an
// implicit GEMM on the bf16 matrix cores with float32 accuracy.
//
// Where it sits: CBAM's two 3x3 64->64 convolutions on [8, 64, 256, 256] (src/UM_Net/MMUNet.py:313-338; 27 % of the
// model's conv FLOPs) and every 3x3 conv of model.py's plain Unet.  MIOpen runs them as
// fp32 Winograd at ~87 TFLOP/s effective; the f32-input MFMA peaks at 157 TFLOP/s, the bf16 MFMA at 2.5 PFLOP/s.
//
// float32 accuracy on the bf16 pipe: x = xh + xl with xh = bf16(x), xl = bf16(x - xh) (16 mantissa bits kept), and
//   x*w  ~=  xh*wh + xh*wl + xl*wh        (the dropped xl*wl term and the residuals are ~2^-16 relative)
// -- three v_mfma_f32_32x32x16_bf16 per product, float32 accumulation: 833 TFLOP/s of fp32-grade peak.
//
// GEMM view: M = output channels (A = weights), N = pixels (B = input), K = (3x3 shift) x input channel.
//   workgroup = 512 threads = 8 waves = an 8-row x 64-column pixel tile x 64 output channels; wave w owns tile row w:
//   2 (M) x 2 (N) tiles of 32 x 32, 64 accumulator registers.
//   K loop: chunks of 16 input channels.  Per chunk the workgroup stages in LDS
//     * the (8+2) x (64+2) input patch, converted on the fly to [pixel][16 ci] bf16 (hi and lo images): the B
//       fragment of a lane (8 consecutive ci of one pixel) is one ds_read_b128, a wave reads 2 KiB contiguous, and a
//       3x3 shift is just another pixel offset -- this is the im2col stage, done by addressing;
//     * the chunk's weights [shift][co][16 ci] bf16 hi / lo (prepared once per call by a small kernel).
//   Per shift and wave: 8 ds_read_b128 feed 12 MFMAs (0.67 per MFMA; the LDS pipe allows 2).  The next chunk's
//   global loads are issued before the MFMAs of the current one and held in registers (52 VGPRs).
// The input gradient is the same kernel on flipped / transposed weights (prepared by the same small kernel).
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TH = 8, TW = 64, PW = TW + 2, PH = TH + 2, CK = 16;
constexpr int PATCH_BYTES = PH * PW * CK * 2;        // one bf16 image of the patch: 21,120 B
constexpr int WCH_BYTES = 9 * 64 * CK * 2;           // one bf16 image of a chunk's weights: 18,432 B
constexpr int STAGE_BYTES = 2 * PATCH_BYTES + 2 * WCH_BYTES;   // patch hi | patch lo | weights hi | weights lo: 79,104 B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;                      // double-buffered: 158,208 B of the CU's 160 KiB

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
// hi / lo split of two floats -> one packed word each
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo) {
    hi = pack_bf16(a, b);
    const float ah = __builtin_bit_cast(float, hi << 16), bh = __builtin_bit_cast(float, hi & 0xffff0000u);
    // two plain v_sub_f32: left to the compiler the pair becomes ONE v_pk_add_f32, which beside MFMAs costs more than
    // the two it replaces (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
    float la, lb;
    asm("v_sub_f32 %0, %1, %2" : "=v"(la) : "v"(a), "v"(ah));
    asm("v_sub_f32 %0, %1, %2" : "=v"(lb) : "v"(b), "v"(bh));
    lo = pack_bf16(la, lb);
}

// weights [Cout][Cin][3][3] f32 (flip = 0) or, for the input gradient, the ORIGINAL weight [Cin_eff][Cout_eff][3][3]
// read transposed and spatially flipped (flip = 1)  ->  [Cout/64][Cin/16][hi|lo][9][ci half][64 co][8 ci] bf16
__global__ __launch_bounds__(256) void conv3x3_mfma_prep_kernel(const float *__restrict__ w, unsigned short *__restrict__ out,
                                                                int Cin, int Cout, int flip) {
    const long n = (long)Cout * Cin * 9;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    // idx = (((cot * nch + ch) * 9 + s) * 64 + co) * 16 + k   (hi image; lo image is WCH elements further)
    const int k = (int)(idx & 15), co = (int)((idx >> 4) & 63);
    long r = idx >> 10;
    const int s = (int)(r % 9);
    r /= 9;
    const int nch = Cin / CK;
    const int ch = (int)(r % nch), cot = (int)(r / nch);
    const int oc = cot * 64 + co, ic = ch * CK + k;
    const float v = flip ? w[((long)ic * Cout + oc) * 9 + (8 - s)] : w[((long)oc * Cin + ic) * 9 + s];
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    const long base = ((long)(cot * nch + ch) * 2) * (9 * 64 * 16) + (((long)s * 2 + (k >> 3)) * 64 + co) * 8 + (k & 7);
    out[base] = __builtin_bit_cast(unsigned short, h);
    out[base + 9 * 64 * 16] = __builtin_bit_cast(unsigned short, l);
}

struct ConvArgsM {
    const void *x;            // float, or bf16_t in the XB instantiation
    const unsigned short *wp;
    const float *bias;
    void *out;                // same type as x
    int B, Cin, Cout, H, W, tiles_x, tiles_y, ncot, total_tiles;
};

// Persistent: one workgroup per CU walks tiles blockIdx.x, + gridDim.x, ...; the chunk pipeline (registers hold
// chunk i+2, LDS buffer (i+1)&1 is written while the MFMAs read buffer i&1) runs straight across tile boundaries,
// so a tile's first loads, its staging and the previous tile's output stores all sit under MFMAs.  (One workgroup
// per launch slot instead: every tile exposed a memory latency + staging + 64 stores per lane -- 32 % MFMA
// utilisation at 64 -> 64 channels, where a tile has only 4 chunks.)
// XB: input and output are bfloat16 (activations under autocast): a bf16 value is its own hi part -- the staged patch
// has no lo image, a product is two MFMAs (W lo x X, W hi x X), the result is rounded to bf16 where it is stored.
template <bool XB>
__global__ __launch_bounds__(512, 2) void conv3x3_mfma_kernel(ConvArgsM p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nch = p.Cin / CK;
    const long HW = (long)p.H * p.W;
    const int G = gridDim.x;
    const int ntl = (p.total_tiles - (int)blockIdx.x + G - 1) / G;   // tiles of this workgroup (>= 1)
    const int niter = ntl * nch;

    // ---- patch work items of this thread: (channel half, patch pixel), pixel fastest over the lanes -- the global
    // loads are dword-coalesced along x and each lane's 8 channels land as ONE 16-byte LDS write, consecutive
    // lanes on consecutive pixels (conflict-free).  1,320 items over 512 threads: 3 rounds.
    constexpr int NPX = PH * PW, NITEM = 2 * NPX;
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
    auto decode = [&](int tj, int &b, int &cot, int &y0, int &x0) {
        int t = (int)blockIdx.x + tj * G;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        t /= p.tiles_y;
        cot = t % p.ncot;
        b = t / p.ncot;
        y0 = ty * TH;
        x0 = tx * TW;
    };

    // ---- load stream (two chunks ahead of the MFMAs)
    unsigned px[3][8], pmask[3];   // raw bits of the loaded values (converted where they are staged); 0 / ~0 masks
    v4u wr[5];
    int l_tj = 0, l_ch = 0, l_b, l_cot, l_y0, l_x0;
    decode(0, l_b, l_cot, l_y0, l_x0);
    auto prefetch = [&]() {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int gy = l_y0 - 1 + ipr[k], gx = l_x0 - 1 + ipc[k];
            const bool inb = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            pmask[k] = inb ? 0xffffffffu : 0u;
            const long off = ((long)l_b * p.Cin + l_ch * CK + 8 * ihalf[k]) * HW + (inb ? (long)gy * p.W + gx : 0);
            if constexpr (XB) {
                const unsigned short *s = (const unsigned short *)p.x + off;
#pragma unroll
                for (int j = 0; j < 8; ++j) px[k][j] = s[j * HW];
            } else {
                const unsigned *s = (const unsigned *)p.x + off;
#pragma unroll
                for (int j = 0; j < 8; ++j) px[k][j] = s[j * HW];
            }
        }
        const v4u *ws = reinterpret_cast<const v4u *>(p.wp + ((long)l_cot * nch + l_ch) * (2 * 9 * 64 * 16));
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int q = tid + 512 * j;
            wr[j] = ws[q < 2304 ? q : 2303];
        }
        // advance; past the end the stream keeps re-reading the last chunk (nothing consumes it)
        if (l_ch + 1 < nch) {
            ++l_ch;
        } else if (l_tj + 1 < ntl) {
            l_ch = 0;
            ++l_tj;
            decode(l_tj, l_b, l_cot, l_y0, l_x0);
        }
    };
    auto stage = [&](unsigned char *buf) {
        unsigned char *patch_hi = buf, *patch_lo = buf + PATCH_BYTES, *w_hi = buf + 2 * PATCH_BYTES;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            unsigned hw[4], lw[4];
            if constexpr (XB) {
#pragma unroll
                for (int j = 0; j < 4; ++j) hw[j] = (px[k][2 * j] | (px[k][2 * j + 1] << 16)) & pmask[k];
                const v4u h = {hw[0], hw[1], hw[2], hw[3]};
                if (ioff[k] >= 0) *reinterpret_cast<v4u *>(patch_hi + ioff[k]) = h;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    split2(__uint_as_float(px[k][2 * j] & pmask[k]), __uint_as_float(px[k][2 * j + 1] & pmask[k]), hw[j], lw[j]);
                const v4u h = {hw[0], hw[1], hw[2], hw[3]}, l = {lw[0], lw[1], lw[2], lw[3]};
                if (ioff[k] >= 0) {
                    *reinterpret_cast<v4u *>(patch_hi + ioff[k]) = h;
                    *reinterpret_cast<v4u *>(patch_lo + ioff[k]) = l;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int q = tid + 512 * j;
            if (q < 2304) *reinterpret_cast<v4u *>(w_hi + q * 16) = wr[j];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

    // fragment addressing: lanes 0-31 (k 0-7) and lanes 32-63 (k 8-15) read from separate planes, 16 B per row /
    // pixel, so each half-wave reads 512 contiguous bytes (a [row][16 ci] image is a 2-way bank conflict)
    const int a_lane = (lane >> 5) * (64 * 16) + (lane & 31) * 16;        // weights: [shift][plane][64 co][8 ci]
    const int b_lane = (lane >> 5) * (NPX * 16) + (lane & 31) * 16;       // patch:   [plane][pixel][8 ci]

    // Nothing in the loop that issues loads is conditional (a branch around loads costs a vmcnt(0) at its merge);
    // the barrier waits for LDS only (lgkmcnt), so loads and the epilogue's stores stay in flight across it.
    prefetch();
    stage(lds);
    prefetch();
    MMU_LDS_BARRIER();
    int c_tj = 0, c_ch = 0;
    // The two waves of a SIMD (w and w + 4) run the same program between the same barriers: both convert and store the next
    // chunk (~340 vector instructions per 108 MFMAs) at the top of an iteration, then both queue on the matrix pipe.
    // MMU_CONV3_STAGGER=1 makes waves 4-7 stage BEHIND their MFMAs, so that on every SIMD one wave's staging runs beside the
    // other's matrix work -- measured (round 4): 118.2 / 118.4 us against 118.6 / 121.7 without, i.e. nothing; off.
#ifndef MMU_CONV3_STAGGER
#define MMU_CONV3_STAGGER 0
#endif
    const bool late = MMU_CONV3_STAGGER && wv >= 4;
    for (int it = 0; it < niter; ++it) {
        const unsigned char *cur = lds + (it & 1) * STAGE_BYTES;
        const unsigned char *patch_hi = cur, *patch_lo = cur + PATCH_BYTES, *w_hi = cur + 2 * PATCH_BYTES;
        // MMU_CONV3_EXP (timing experiments only, results are wrong): 1 = no staging and no loads inside the loop,
        // 2 = no MFMAs / fragment reads, 3 = staging but no global loads
#ifndef MMU_CONV3_EXP
#define MMU_CONV3_EXP 0
#endif
        if (!late && MMU_CONV3_EXP != 1) {
            stage(lds + ((it + 1) & 1) * STAGE_BYTES);
            if (MMU_CONV3_EXP != 3) prefetch();
        }
        // keep the loads ahead of the MFMAs: left alone, the scheduler sinks them behind the last MFMA (their
        // destination registers then double as fragment registers) and every chunk waits out a full memory latency
        __builtin_amdgcn_sched_barrier(0);
        // fragments of shift s + 1 are read while the MFMAs of shift s run (two register sets; left to itself the compiler
        // keeps ONE set and issues the next shift's reads behind the last MFMA: an exposed LDS round trip per shift)
        bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
        auto frags = [&](int s, int set) {
            const int kh = s / 3, kw = s - 3 * kh;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int off = s * (2 * 64 * 16) + m * (32 * 16) + a_lane;
                ah[set][m] = *reinterpret_cast<const bf16x8 *>(w_hi + off);
                al[set][m] = *reinterpret_cast<const bf16x8 *>(w_hi + WCH_BYTES + off);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int off = ((wv + kh) * PW + n * 32 + kw) * 16 + b_lane;
                bh[set][n] = *reinterpret_cast<const bf16x8 *>(patch_hi + off);
                if constexpr (!XB) bl[set][n] = *reinterpret_cast<const bf16x8 *>(patch_lo + off);
            }
        };
        frags(0, 0);
#pragma unroll
        for (int s = 0; s < (MMU_CONV3_EXP == 2 ? 0 : 9); ++s) {
            const int c = s & 1;
            if (s + 1 < 9) {
                frags(s + 1, c ^ 1);
                __builtin_amdgcn_sched_barrier(0);   // the reads are issued BEFORE this shift's MFMAs
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[c][m], bh[c][n], acc[m][n], 0, 0, 0);
                    if constexpr (!XB) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c][m], bl[c][n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c][m], bh[c][n], acc[m][n], 0, 0, 0);
                }
            // pin the order: this shift's MFMAs stay behind the reads of the next one
            if (s + 1 < 9) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (late) {
            stage(lds + ((it + 1) & 1) * STAGE_BYTES);
            prefetch();
        }
        if (++c_ch == nch) {
            // ---- tile done: C layout col = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel)
            int b, cot, y0, x0;
            decode(c_tj, b, cot, y0, x0);
            const int oy = y0 + wv;
            // bias values of this lane's 32 channel rows first, as one batch of loads (a load inside the store loop
            // is a load -> vmcnt(0) -> store round trip per element: 64 serialised memory latencies per tile)
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
                    const int ox = x0 + n * 32 + (lane & 31);
                    if (oy < p.H && ox < p.W) {
                        const long o0 = ((long)b * p.Cout + cot * 64 + m * 32 + 4 * (lane >> 5)) * HW + (long)oy * p.W + ox;
                        if constexpr (XB) {
                            bf16_t *op = (bf16_t *)p.out + o0;
#pragma unroll
                            for (int e = 0; e < 16; ++e) op[((e & 3) + 8 * (e >> 2)) * HW] = from_f32<bf16_t>(acc[m][n][e] + bv[m][e]);
                        } else {
                            float *op = (float *)p.out + o0;
#pragma unroll
                            for (int e = 0; e < 16; ++e) op[((e & 3) + 8 * (e >> 2)) * HW] = acc[m][n][e] + bv[m][e];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
                }
            c_ch = 0;
            ++c_tj;
        }
        MMU_LDS_BARRIER();
    }
}


// ---- the same convolution with the staging and the matrix work on DIFFERENT waves (round 4) ---------------------------
// Measured on the kernel above (tools/ab_conv3.sh; [8, 256, 64, 64] -> 256, no HBM influence): 110 us in all = 76 us
// with the staging and loads removed + 38 us with the MFMAs removed -- the two phases add up, nothing overlaps: the two
// waves of a SIMD run the same program between the same barriers, so both convert / store the next chunk, then both queue
// on the matrix pipe.  (Staggering one of them behind its MFMAs put a branch around the loads -- a vmcnt(0) at each
// merge -- and gained nothing.)  Here waves 0-3 (one per SIMD) are PRODUCERS: global loads two chunks ahead, hi/lo
// split, LDS stores of the next chunk; waves 4-7 (one per SIMD) are CONSUMERS: two tile rows each, 128 accumulator
// registers, 24 MFMAs per 3x3 shift on 12 fragment reads (the weights' fragments serve both rows), the next step's
// fragments read while the current MFMAs run.  Two separate loops with the same number of barriers: a SIMD's vector
// issue slots between the consumer's MFMAs (24 of every 32 cycles) carry the producer's instructions.
template <bool XB>
__global__ __launch_bounds__(512, 2) void conv3x3_mfma_ws_kernel(ConvArgsM p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nch = p.Cin / CK;
    const long HW = (long)p.H * p.W;
    // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs, each with its own L2.  XCD k takes the k-th
    // contiguous eighth of the tile list (x fastest, then y: at 256 x 256 one whole image), so the halo rows and the
    // 128-byte lines that neighbouring tiles share are fetched from HBM once, not once per XCD.
    const int G = gridDim.x;
    const bool xcd_map = (G & 7) == 0;
    const int per_xcd = (p.total_tiles + 7) >> 3;
    const int t_first = xcd_map ? ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int t_step = xcd_map ? G >> 3 : G;
    const int t_end = xcd_map ? min((((int)blockIdx.x & 7) + 1) * per_xcd, p.total_tiles) : p.total_tiles;
    const int ntl = t_first < t_end ? (t_end - t_first + t_step - 1) / t_step : 0;   // tiles of this workgroup
    const int niter = ntl * nch;
    constexpr int NPX = PH * PW, NITEM = 2 * NPX;
    auto decode = [&](int tj, int &b, int &cot, int &y0, int &x0) {
        int t = t_first + tj * t_step;
        t = t < p.total_tiles ? t : p.total_tiles - 1;   // (a workgroup without tiles decodes the last one and loads it once)
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        t /= p.tiles_y;
        cot = t % p.ncot;
        b = t / p.ncot;
        y0 = ty * TH;
        x0 = tx * TW;
    };
    if (wv < 4) {
        // ================= producers: 256 threads stage a chunk (1,320 patch items: 6 rounds; 2,304 weight pieces: 9)
        constexpr int PR = (NITEM + 255) / 256, WR = 2304 / 256;
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
        int l_tj = 0, l_ch = 0, l_b, l_cot, l_y0, l_x0;
        decode(0, l_b, l_cot, l_y0, l_x0);
        // Patch loads through a buffer resource: base = (batch item, chunk) in SGPRs, the channel row j as a scalar
        // offset, ONE per-lane byte offset per item (non-negative by construction) -- no 64-bit address arithmetic per
        // load -- and padded pixels ask for offset 2^31 >= num_records, which the hardware answers with 0: no mask
        // instructions either.  (The host checks 9 H W elements < 2^31 bytes.)
        constexpr unsigned ES = XB ? 2 : 4, OOB = 0x80000000u;
        auto prefetch = [&]() {
            const char *base = (const char *)p.x + ((long)l_b * p.Cin + l_ch * CK) * HW * ES;
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, (int)OOB, 0x00020000);
#pragma unroll
            for (int k = 0; k < PR; ++k) {
                const int gy = l_y0 - 1 + ipr[k], gx = l_x0 - 1 + ipc[k];
                const bool inb = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                const unsigned voff = inb ? ((unsigned)(8 * ihalf[k]) * (unsigned)HW + (unsigned)(gy * p.W + gx)) * ES : OOB;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (XB)
                        px[k][j] = __builtin_amdgcn_raw_buffer_load_b16(rs, voff, (unsigned)(j * HW) * ES, 0);
                    else
                        px[k][j] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (unsigned)(j * HW) * ES, 0);
                }
            }
            const v4u *ws = reinterpret_cast<const v4u *>(p.wp + ((long)l_cot * nch + l_ch) * (2 * 9 * 64 * 16));
#pragma unroll
            for (int j = 0; j < WR; ++j) wr[j] = ws[tid + 256 * j];
            if (l_ch + 1 < nch) {           // past the end the stream re-reads the last chunk (nothing consumes it)
                ++l_ch;
            } else if (l_tj + 1 < ntl) {
                l_ch = 0;
                ++l_tj;
                decode(l_tj, l_b, l_cot, l_y0, l_x0);
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
                    const v4u h = {hw[0], hw[1], hw[2], hw[3]};
                    if (ioff[k] >= 0) *reinterpret_cast<v4u *>(patch_hi + ioff[k]) = h;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        split2(__uint_as_float(px[k][2 * j]), __uint_as_float(px[k][2 * j + 1]), hw[j], lw[j]);
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
            if (MMU_CONV3_EXP != 1) {        // (timing experiments: see the kernel above)
                stage(lds + ((it + 1) & 1) * STAGE_BYTES);
                if (MMU_CONV3_EXP != 3) prefetch();
            }
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
        for (int it = 0; it < niter; ++it) {
            const unsigned char *cur = lds + (it & 1) * STAGE_BYTES;
            const unsigned char *patch_hi = cur, *patch_lo = cur + PATCH_BYTES, *w_hi = cur + 2 * PATCH_BYTES;
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
                const int kh = s / 3, kw = s - 3 * kh;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int off = ((2 * cw + r + kh) * PW + n * 32 + kw) * 16 + b_lane;
                    bh[set][n] = *reinterpret_cast<const bf16x8 *>(patch_hi + off);
                    if constexpr (!XB) bl[set][n] = *reinterpret_cast<const bf16x8 *>(patch_lo + off);
                }
            };
            frag_a(0, 0);
            frag_b(0, 0, 0);
#pragma unroll
            for (int step = 0; step < (MMU_CONV3_EXP == 2 ? 0 : 18); ++step) {      // step = (shift, row)
                const int s = step >> 1, r = step & 1, ca = s & 1, cb = step & 1;
                if (step + 1 < 18) {
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
                if (step + 1 < 18) __builtin_amdgcn_sched_barrier(0);
            }
            if (++c_ch == nch) {
                // ---- tile done: C layout col = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel)
                int b, cot, y0, x0;
                decode(c_tj, b, cot, y0, x0);
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
                // stores through a buffer resource too: base = (batch item, channel tile) in SGPRs, the channel row as a
                // scalar offset, one byte offset per lane and pixel; lanes outside the image store to offset 2^31 (dropped)
                constexpr unsigned ES = XB ? 2 : 4, OOB = 0x80000000u;
                char *obase = (char *)p.out + ((long)b * p.Cout + cot * 64) * HW * ES;
                const rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)OOB, 0x00020000);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int oy = y0 + 2 * cw + r;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int ox = x0 + n * 32 + (lane & 31);
                        const unsigned voff = (oy < p.H && ox < p.W)
                                                  ? ((unsigned)(4 * (lane >> 5)) * (unsigned)HW + (unsigned)(oy * p.W + ox)) * ES
                                                  : OOB;
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const unsigned soff = (unsigned)((m * 32 + (e & 3) + 8 * (e >> 2)) * HW) * ES;
                                const float v = acc[r][m][n][e] + bv[m][e];
                                if constexpr (XB)
                                    __builtin_amdgcn_raw_buffer_store_b16(from_f32<bf16_t>(v).bits, ors, voff, soff, 0);
                                else
                                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ors, voff, soff, 0);
                                acc[r][m][n][e] = 0.f;
                            }
                        }
                    }
                }
                c_ch = 0;
                ++c_tj;
            }
            MMU_LDS_BARRIER();
        }
    }
}

}  // namespace

extern "C" size_t mmu_conv3x3_mfma_workspace_bytes(int in_channels, int out_channels) {
    if (in_channels <= 0 || out_channels <= 0) return 0;
    return (size_t)in_channels * out_channels * 9 * 2 * sizeof(unsigned short);
}

extern "C" int mmu_conv3x3_mfma(const mmu_conv3x3_mfma_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "conv3x3_mfma: null params");
    MMU_CHECK(p->batch > 0 && p->height > 0 && p->width > 0, "conv3x3_mfma: empty tensor");
    MMU_CHECK(p->in_channels > 0 && p->in_channels % 16 == 0 && p->out_channels > 0 && p->out_channels % 64 == 0,
              "conv3x3_mfma: in_channels must be a multiple of 16 and out_channels of 64 (got %d, %d)", p->in_channels,
              p->out_channels);
    MMU_CHECK(p->width % 4 == 0, "conv3x3_mfma: width must be a multiple of 4 (got %d)", p->width);
    MMU_CHECK(p->input && p->weight && p->out && p->workspace, "conv3x3_mfma: input, weight, out, workspace are required");
    MMU_CHECK(((uintptr_t)p->input & 15) == 0 && ((uintptr_t)p->workspace & 15) == 0,
              "conv3x3_mfma: input and workspace must be 16-byte aligned");
    const bool xb = p->io_dtype == MMU_DTYPE_BF16;
    MMU_CHECK(xb || p->io_dtype == MMU_DTYPE_F32, "conv3x3_mfma: io_dtype must be float32 or bfloat16 (got %d)", p->io_dtype);
    hipStream_t st = (hipStream_t)stream;
    const long nw = (long)p->in_channels * p->out_channels * 9;
    conv3x3_mfma_prep_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, st>>>(
        p->weight, (unsigned short *)p->workspace, p->in_channels, p->out_channels, p->transposed ? 1 : 0);
    MMU_HIP_LAUNCH_CHECK("conv3x3_mfma(prep)");
    // MMU_CONV3_WS=0: the kernel whose eight waves all stage and multiply (A/B; default: producer / consumer waves)
    static const bool ws_on = []() { const char *e = getenv("MMU_CONV3_WS"); return !e || e[0] != '0'; }();
    // (its buffer-resource addressing keeps a channel tile of one batch item within 32-bit byte offsets)
    const bool ws = ws_on && 64L * p->height * p->width * (xb ? 2 : 4) < (1L << 31);
    static unsigned long long attr_mask = 0, attr_mask_b = 0, attr_mask_ws = 0, attr_mask_ws_b = 0;  // per device
    hipError_t e;
    if (ws)
        e = xb ? mmu_set_lds_once(conv3x3_mfma_ws_kernel<true>, LDS_BYTES, attr_mask_ws_b)
               : mmu_set_lds_once(conv3x3_mfma_ws_kernel<false>, LDS_BYTES, attr_mask_ws);
    else
        e = xb ? mmu_set_lds_once(conv3x3_mfma_kernel<true>, LDS_BYTES, attr_mask_b)
               : mmu_set_lds_once(conv3x3_mfma_kernel<false>, LDS_BYTES, attr_mask);
    if (e != hipSuccess) return mmu_fail("conv3x3_mfma: LDS attribute: %s", hipGetErrorString(e));
    ConvArgsM a;
    a.x = p->input; a.wp = (const unsigned short *)p->workspace; a.bias = p->bias; a.out = p->out;
    a.B = p->batch; a.Cin = p->in_channels; a.Cout = p->out_channels; a.H = p->height; a.W = p->width;
    a.tiles_x = (p->width + TW - 1) / TW;
    a.tiles_y = (p->height + TH - 1) / TH;
    a.ncot = p->out_channels / 64;
    const long total = (long)a.tiles_x * a.tiles_y * a.ncot * p->batch;
    MMU_CHECK(total < (1L << 30), "conv3x3_mfma: too many tiles");
    a.total_tiles = (int)total;
    const int n_cu = mmu_cu_count();
    const int grid = total < n_cu ? (int)total : n_cu;   // one workgroup (158 KB of LDS) per CU
    if (ws && xb)
        conv3x3_mfma_ws_kernel<true><<<grid, 512, LDS_BYTES, st>>>(a);
    else if (ws)
        conv3x3_mfma_ws_kernel<false><<<grid, 512, LDS_BYTES, st>>>(a);
    else if (xb)
        conv3x3_mfma_kernel<true><<<grid, 512, LDS_BYTES, st>>>(a);
    else
        conv3x3_mfma_kernel<false><<<grid, 512, LDS_BYTES, st>>>(a);
    MMU_HIP_LAUNCH_CHECK("conv3x3_mfma");
    return 0;
}
