// morph_mix.hip -- MMConv's deformable sampling AFTER the channel mixing, for gfx950.
//
// MMConv (src/UM_Net/MMUNet.py:245-274) samples its input at learned rows (one row map per tap k, shared by all
// channels) and then mixes channels and taps with the K x 1 DSC convolution:
//     out[o] = sum_c sum_k W[o][c][k] * S_k(x[c]),      S_k = bilinear row sampling at tap k's coordinates.
// S_k is linear and the same for every channel, so it commutes with the channel mixing:
//     out[o] = sum_k S_k( Y[k][o] ),                    Y[k][o] = sum_c W[o][c][k] x[c]   (a 1x1 convolution of x).
// morph_sample.hip materialises S_k(x[c]) -- K * Cin planes; this file samples Y -- K * Cout planes.  For the blocks that
// REDUCE the channel count (DecoderBlock.conv1 Cin -> Cin/4, SideoutBlock.conv1 64 -> 16, RCG.conv1 128 -> 64:
// MMUNet.py:344-349,357-359,424-430) the tensor that round-trips HBM shrinks by Cin / Cout: at 256 x 256 the side
// output's sample matrix was 403 MB, Y is 100 MB.  Same arithmetic up to the order of the float32 sums.
//
//     yc = clamp(y[b,k,h,w], 0, H-1);  y0 = floor(yc);  wy = yc - y0;  col = clamp(w + k - K/2, 0, W-1)
//     out[b,o,h,w]          = sum_k (1-wy) Y[b,kO+o,y0,col] + wy Y[b,kO+o,y0+1,col]         (row H contributes 0)
//     dY[b,kO+o,y0,col]    += (1-wy) g[b,o,h,w];   dY[b,kO+o,y0+1,col] += wy g[b,o,h,w]
//     dy[b,k,h,w]           = [0 <= y <= H-1] sum_o g[b,o,h,w] (Y[b,kO+o,y0+1,col] - Y[b,kO+o,y0,col])
// dY is a GATHER as in morph_sample.hip (the thread of a target element visits the sources within REACH rows that can
// reach it: plain stores, reproducible); sources further away are added by the d(row) kernel with float atomics.
// Y / dY carry (batch, channel) element strides: the GEMM that produces Y writes it channel-major.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

constexpr int REACH = 2;   // rows

struct MixArgs {
    int B, O, H, W;
    long y_bs, y_cs;        // element strides of Y / dY over batch and channel (planes are H x W contiguous)
    long g_bs;              // batch stride of out / dout ([B][O][H][W] contiguous: O * H * W)
    const float *Y, *rows, *dout;
    float *out, *dY, *drows;
};

template <int K>
struct Taps {
    int y0[K], col[K];
    float wy[K];
    bool has1[K], inside[K];
};

template <int K>
__device__ __forceinline__ void decode_taps(const MixArgs &p, int b, int h, int w, Taps<K> &t) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float yr = p.rows[((long)(b * K + k) * p.H + h) * p.W + w];
        const float yc = fminf(fmaxf(yr, 0.f), (float)(p.H - 1));
        t.y0[k] = (int)floorf(yc);
        t.wy[k] = yc - (float)t.y0[k];
        t.has1[k] = t.y0[k] + 1 <= p.H - 1;
        t.inside[k] = yr >= 0.f && yr <= (float)(p.H - 1);
        const int c = w + k - K / 2;
        t.col[k] = c < 0 ? 0 : (c > p.W - 1 ? p.W - 1 : c);
    }
}

// forward: thread = pixel (h, w) of batch item blockIdx.z; walks all O output channels
template <int K>
__global__ __launch_bounds__(256) void mix_sample_fwd_kernel(MixArgs p) {
    const int HW = p.H * p.W;
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= HW) return;
    const int b = blockIdx.z, h = pos / p.W, w = pos - h * p.W;
    Taps<K> t;
    decode_taps<K>(p, b, h, w, t);
    long off[K];
#pragma unroll
    for (int k = 0; k < K; ++k) off[k] = (long)b * p.y_bs + (long)k * p.O * p.y_cs + (long)t.y0[k] * p.W + t.col[k];
    float *op = p.out + (long)b * p.g_bs + pos;
#pragma unroll 4
    for (int o = 0; o < p.O; ++o) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float *s = p.Y + off[k] + (long)o * p.y_cs;
            const float v0 = s[0];
            const float v1 = t.has1[k] ? s[p.W] : 0.f;
            acc += fmaf(t.wy[k], v1 - v0, v0);
        }
        op[(long)o * HW] = acc;
    }
}

// d(row) + the far contributions to dY: thread = source pixel (h, w); runs AFTER the gather kernel
template <int K>
__global__ __launch_bounds__(256) void mix_sample_drows_kernel(MixArgs p) {
    const int HW = p.H * p.W;
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= HW) return;
    const int b = blockIdx.z, h = pos / p.W, w = pos - h * p.W;
    Taps<K> t;
    decode_taps<K>(p, b, h, w, t);
    long off[K];
    bool far0[K], far1[K];
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        off[k] = (long)b * p.y_bs + (long)k * p.O * p.y_cs + (long)t.y0[k] * p.W + t.col[k];
        far0[k] = t.y0[k] - h > REACH || h - t.y0[k] > REACH;
        far1[k] = t.has1[k] && (t.y0[k] + 1 - h > REACH || h - t.y0[k] - 1 > REACH);
        acc[k] = 0.f;
    }
    const float *gp = p.dout + (long)b * p.g_bs + pos;
#pragma unroll 2
    for (int o = 0; o < p.O; ++o) {
        const float g = gp[(long)o * HW];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long e = off[k] + (long)o * p.y_cs;
            const float v0 = p.Y[e];
            const float v1 = t.has1[k] ? p.Y[e + p.W] : 0.f;
            acc[k] = fmaf(g, v1 - v0, acc[k]);
            if (far0[k]) atomicAdd(p.dY + e, g * (1.f - t.wy[k]));
            if (far1[k]) atomicAdd(p.dY + e + p.W, g * t.wy[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        p.drows[((long)(b * K + k) * p.H + h) * p.W + w] = t.inside[k] ? acc[k] : 0.f;   // d clamp
}

// dY as a gather: grid (ceil(H*W / 256), K, B * ceil(O / CS)); thread = target (yy, col) of tap blockIdx.y
template <int K, int CS>
__global__ __launch_bounds__(256) void mix_gather_dY_kernel(MixArgs p) {
    const int HW = p.H * p.W;
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= HW) return;
    const int k = blockIdx.y;
    const int yy = pos / p.W, col = pos - yy * p.W;
    const int nsl = (p.O + CS - 1) / CS;
    const int b = blockIdx.z / nsl;
    const int o0 = (blockIdx.z - b * nsl) * CS;
    const int no = p.O - o0 < CS ? p.O - o0 : CS;
    float acc[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) acc[c] = 0.f;
    const int hlo = yy - REACH > 0 ? yy - REACH : 0;
    const int hhi = yy + REACH < p.H - 1 ? yy + REACH : p.H - 1;
    // source columns w with clamp(w + k - K/2, 0, W-1) == col
    const int we = col - k + K / 2;
    int wlo = col == 0 ? 0 : we, whi = col == p.W - 1 ? p.W - 1 : we;
    wlo = wlo < 0 ? 0 : wlo;
    whi = whi > p.W - 1 ? p.W - 1 : whi;
    for (int w = wlo; w <= whi; ++w) {
        const float *yp = p.rows + ((long)(b * K + k) * p.H + hlo) * p.W + w;
        const float *gp = p.dout + (long)b * p.g_bs + (long)o0 * HW + (long)hlo * p.W + w;
        for (int h = hlo; h <= hhi; ++h, yp += p.W, gp += p.W) {
            const float yc = fminf(fmaxf(*yp, 0.f), (float)(p.H - 1));
            const int y0 = (int)floorf(yc);
            const float wy = yc - (float)y0;
            const float coef = y0 == yy ? 1.f - wy : (y0 + 1 == yy ? wy : 0.f);
            if (coef != 0.f) {
                if (no == CS) {
#pragma unroll
                    for (int c = 0; c < CS; ++c) acc[c] = fmaf(coef, gp[(long)c * HW], acc[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < CS; ++c)
                        if (c < no) acc[c] = fmaf(coef, gp[(long)c * HW], acc[c]);
                }
            }
        }
    }
    float *dst = p.dY + (long)b * p.y_bs + ((long)k * p.O + o0) * p.y_cs + pos;
#pragma unroll
    for (int c = 0; c < CS; ++c)
        if (c < no) dst[(long)c * p.y_cs] = acc[c];
}

int fill(const mmu_morph_mix_params *p, MixArgs &a, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->batch > 0 && p->out_channels > 0 && p->height > 0 && p->width > 0, "%s: empty tensor", name);
    MMU_CHECK(p->taps == 1 || p->taps == 3, "%s: taps must be 1 or 3 (got %d)", name, p->taps);
    MMU_CHECK(p->batch <= 65535, "%s: batch too large for the launch grid", name);
    MMU_CHECK((long)p->height * p->width < (1L << 30), "%s: map too large", name);
    MMU_CHECK(p->mixed && p->y, "%s: mixed and y are required", name);
    a = MixArgs{};
    a.B = p->batch; a.O = p->out_channels; a.H = p->height; a.W = p->width;
    a.y_bs = p->mixed_bs; a.y_cs = p->mixed_cs;
    a.g_bs = (long)p->out_channels * p->height * p->width;
    a.Y = p->mixed; a.rows = p->y;
    return 0;
}

}  // namespace

extern "C" int mmu_morph_mix_sample_fwd(const mmu_morph_mix_params *p, void *stream) {
    MixArgs a;
    if (int r = fill(p, a, "morph_mix_sample_fwd")) return r;
    MMU_CHECK(p->out, "morph_mix_sample_fwd: out is required");
    a.out = p->out;
    dim3 grid((a.H * a.W + 255) / 256, 1, a.B);
    if (p->taps == 3)
        mix_sample_fwd_kernel<3><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    else
        mix_sample_fwd_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    MMU_HIP_LAUNCH_CHECK("morph_mix_sample_fwd");
    return 0;
}

extern "C" int mmu_morph_mix_sample_bwd(const mmu_morph_mix_params *p, void *stream) {
    MixArgs a;
    if (int r = fill(p, a, "morph_mix_sample_bwd")) return r;
    MMU_CHECK(p->dout && p->dmixed && p->dy, "morph_mix_sample_bwd: dout, dmixed, dy are required");
    a.dout = p->dout; a.dY = p->dmixed; a.drows = p->dy;
    hipStream_t st = (hipStream_t)stream;
    constexpr int CS = 16;
    dim3 gg((a.H * a.W + 255) / 256, p->taps, a.B * ((a.O + CS - 1) / CS));
    MMU_CHECK(gg.z <= 65535, "morph_mix_sample_bwd: batch * channel slices too large for the launch grid");
    dim3 grid((a.H * a.W + 255) / 256, 1, a.B);
    if (p->taps == 3) {
        mix_gather_dY_kernel<3, CS><<<gg, 256, 0, st>>>(a);
        MMU_HIP_LAUNCH_CHECK("morph_mix_sample_bwd(gather)");
        mix_sample_drows_kernel<3><<<grid, 256, 0, st>>>(a);
    } else {
        mix_gather_dY_kernel<1, CS><<<gg, 256, 0, st>>>(a);
        MMU_HIP_LAUNCH_CHECK("morph_mix_sample_bwd(gather)");
        mix_sample_drows_kernel<1><<<grid, 256, 0, st>>>(a);
    }
    MMU_HIP_LAUNCH_CHECK("morph_mix_sample_bwd");
    return 0;
}
