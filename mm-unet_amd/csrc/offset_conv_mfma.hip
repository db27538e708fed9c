// offset_conv_mfma.hip -- input gradient of MMConv's offset convolution nn.Conv2d(Cin, 6, 3, padding=1)
// (src/UM_Net/MMUNet.py:46,250) on the bf16 matrix cores with float32-grade products.
//
//   dx[ci][y][x] = sum_co sum_{ty,tx in -1..1} W[co][ci][1-ty][1-tx] * dout[co][y+ty][x+tx]
//
// On the vector pipe (conv3x3_small.hip: 54 FMAs per input channel and pixel with scalar weights) this is 0.78 ms of a
// training step, 41 calls.  As a GEMM it is friendly: M = Cin (64..512: full 32-row tiles), N = pixels, K = 9 taps x 6
// channels -> 10 groups of 8 (k = tap * 8 + co; co 6, 7 and group 9 are zero) = 5 chunks of 16.  The whole B operand of
// a tile is its dout patch -- (8 + 2) x (32 + 2) pixels x 8 channel slots, split ONCE into three bf16 parts while it is
// staged in LDS as [part][pixel][8 co] (16 KB): a lane's 8 consecutive k of a tap are one ds_read_b128, a 3 x 3 shift
// is a pixel offset.  The A operand (the weights, flipped and transposed) is built by each wave in registers straight
// from the float32 weight -- 6 loads + one three-part split per 32-row tile and chunk, no prepared image, no extra
// launch -- and the wave walks the 32-row tiles of Cin with the B fragments of its pixels re-read from LDS.
// Products: three bf16 parts per operand, six MFMAs (as gemm_tokens' 32-token kernel and stem7_mfma.hip): the offset
// branch feeds the sampler's coordinates, the most rounding-sensitive path of the model.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TW = 32, PW = TW + 2;   // tile width; patch width

__device__ __forceinline__ unsigned pk(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split3(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    h = pk(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    m = pk(ra, rb);
    l = pk(ra - __builtin_bit_cast(float, m << 16), rb - __builtin_bit_cast(float, m & 0xffff0000u));
}

struct OdArgs {
    const float *dout, *w, *addend;
    float *dx;
    int B, Cin, CO, H, W, tiles_x, tiles_y;
    long w_co, w_ci, w_tap;   // element strides of W[co][ci][tap]
};

// R = rows of pixels per wave (tile = 4 R rows x 32 columns)
template <int R>
__global__ __launch_bounds__(256) void offset_dgrad_mfma_kernel(OdArgs p) {
    constexpr int TH = 4 * R, PH = TH + 2, NPX = PH * PW;
    __shared__ __attribute__((aligned(16))) unsigned short lds[3 * NPX * 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t = blockIdx.x;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y, b = t / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const long HW = (long)p.H * p.W;
    // ---- stage the dout patch: element e = pixel * 8 + co
    {
        const float *gb = p.dout + (long)b * p.CO * HW;
        constexpr int NE = NPX * 8, NI = (NE + 255) / 256;
        float v[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = tid + 256 * i;
            const int co = e & 7, px = e >> 3;
            const int pr = px / PW, pc = px - pr * PW;
            const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
            const bool inb = e < NE && co < p.CO && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            v[i] = inb ? gb[co * HW + (long)gy * p.W + gx] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = tid + 256 * i;
            if (e < NE) {
                const __bf16 h = (__bf16)v[i];
                const float r1 = v[i] - (float)h;
                const __bf16 m = (__bf16)r1;
                const __bf16 l = (__bf16)(r1 - (float)m);
                lds[e] = __builtin_bit_cast(unsigned short, h);
                lds[NE + e] = __builtin_bit_cast(unsigned short, m);
                lds[2 * NE + e] = __builtin_bit_cast(unsigned short, l);
            }
        }
    }
    __syncthreads();
    const int half = lane >> 5, l31 = lane & 31;
    // B fragments of this wave's pixels: [row r][chunk][part], read once, reused for every 32-row tile of Cin
    bf16x8 bf[R][5][3];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int ch = 0; ch < 5; ++ch) {
            const int s = 2 * ch + half;                 // tap 0..8; 9: zero weights (any pixel)
            const int sy = s < 9 ? s / 3 : 1, sx = s < 9 ? s - 3 * (s / 3) : 1;
            const int px = (wv * R + r + sy) * PW + (l31 + sx);
#pragma unroll
            for (int part = 0; part < 3; ++part)
                bf[r][ch][part] = *reinterpret_cast<const bf16x8 *>(lds + part * (NPX * 8) + px * 8);
        }
    const int ox = x0 + l31;
    for (int mt = 0; mt < p.Cin / 32; ++mt) {
        const int ci = mt * 32 + l31;
        f32x16 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
#pragma unroll
        for (int ch = 0; ch < 5; ++ch) {
            // A fragment: row ci, k = co of tap s = 2 ch + half, flipped: W[co][ci][8 - s]
            const int s = 2 * ch + half;
            float wv8[8];
#pragma unroll
            for (int co = 0; co < 8; ++co)
                wv8[co] = (s < 9 && co < p.CO) ? p.w[co * p.w_co + ci * p.w_ci + (8 - s) * p.w_tap] : 0.f;
            unsigned hw[4], mw[4], lw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) split3(wv8[2 * q], wv8[2 * q + 1], hw[q], mw[q], lw[q]);
            const v4u hq = {hw[0], hw[1], hw[2], hw[3]}, mq = {mw[0], mw[1], mw[2], mw[3]}, lq = {lw[0], lw[1], lw[2], lw[3]};
            const bf16x8 ah = __builtin_bit_cast(bf16x8, hq), am = __builtin_bit_cast(bf16x8, mq), al = __builtin_bit_cast(bf16x8, lq);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bf[r][ch][0], acc[r], 0, 0, 0);   // smallest terms first
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[r][ch][2], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bf[r][ch][1], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bf[r][ch][0], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[r][ch][1], acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[r][ch][0], acc[r], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int oy = y0 + wv * R + r;
            if (oy < p.H && ox < p.W) {
                const long off = ((long)b * p.Cin + mt * 32 + 4 * half) * HW + (long)oy * p.W + ox;
                float add[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) add[e] = p.addend ? p.addend[off + ((e & 3) + 8 * (e >> 2)) * HW] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) p.dx[off + ((e & 3) + 8 * (e >> 2)) * HW] = acc[r][e] + add[e];
            }
        }
    }
}

}  // namespace

// Called by mmu_conv3x3_small_bwd (conv3x3_small.hip) for the input gradient: 1 = launched, 0 = shape not covered (the
// caller's direct kernels take it), 2 = error (message set).  MMUNET_OFFSET_DGRAD_MFMA=0 switches it off, =2 takes every covered shape (tests, A/B).
int mmu_offset_dgrad_mfma_try(const mmu_conv3x3s_params *p, hipStream_t st) {
    static const int enabled = getenv("MMUNET_OFFSET_DGRAD_MFMA") ? atoi(getenv("MMUNET_OFFSET_DGRAD_MFMA")) : 1;
    if (!enabled || p->in_dtype != MMU_DTYPE_F32 || p->out_channels > 8 || p->in_channels % 32 != 0 ||
        !p->dinput)
        return 0;
    // measured against the direct kernel (tools/prof_offset_conv.py, batch 8): 32 x 32 x 256 channels 25 vs 13 us (64
    // workgroups walking 8 row tiles each), 64 x 64 x 128: 17 vs 20, 128 x 128 x 64: 22 vs 27, 256 x 256: 37-50 = 37-49
    // (both write-bound) -- taken where it wins
    const long hw = (long)p->height * p->width, lo = enabled == 2 ? 1024 : 4096, hi = enabled == 2 ? (1L << 40) : 65536;
    if (hw < lo || hw >= hi) return 0;
    OdArgs a = {};
    a.dout = p->dout; a.w = p->weight_t; a.addend = (const float *)p->dinput_addend; a.dx = (float *)p->dinput;
    a.B = p->batch; a.Cin = p->in_channels; a.CO = p->out_channels; a.H = p->height; a.W = p->width;
    if (p->weight_native) {   // [CO][Cin][3][3]
        a.w_co = (long)p->in_channels * 9; a.w_ci = 9; a.w_tap = 1;
    } else {                  // [Cin][3][3][CO]
        a.w_co = 1; a.w_ci = 9L * p->out_channels; a.w_tap = p->out_channels;
    }
    a.tiles_x = (p->width + TW - 1) / TW;
    // 8-row tiles when that still gives two workgroups per CU, 4-row tiles otherwise
    const long t8 = (long)a.tiles_x * ((p->height + 7) / 8) * p->batch;
    const int rows = t8 >= 2L * mmu_cu_count() ? 8 : 4;
    a.tiles_y = (p->height + rows - 1) / rows;
    const long total = (long)a.tiles_x * a.tiles_y * p->batch;
    if (total >= (1L << 30)) return 0;
    if (rows == 8)
        offset_dgrad_mfma_kernel<2><<<(unsigned)total, 256, 0, st>>>(a);
    else
        offset_dgrad_mfma_kernel<1><<<(unsigned)total, 256, 0, st>>>(a);
    if (hipGetLastError() != hipSuccess) {
        mmu_fail("offset_dgrad_mfma: launch failed");
        return 2;
    }
    return 1;
}
