// morph_sample.hip -- the deformable sampling step of MMConv ("MorphMamba conv") for gfx950.
//
// Replaces, inside MMConv.forward (src/UM_Net/MMUNet.py:196-242, 259-263), the chain
//     clamp -> scale to [-1,1] (x and y) -> stack into a [B, H*K, W, 2] grid -> F.grid_sample(bilinear,
//     zeros, align_corners=True)
// and its autograd.  In MMConv the x coordinate of tap k at pixel (h, w) is the INTEGER column
// w + k - K/2 (MMUNet.py:141-151: arange + linspace(-c, c, K)); only the row coordinate y is learned.
// So the 4-tap bilinear gather degenerates to a 2-tap vertical lerp in one column:
//
//     yc  = clamp(y[b,k,h,w], 0, H-1) ;  y0 = floor(yc) ;  wy = yc - y0 ;  col = clamp(w + k - K/2, 0, W-1)
//     out[b,c,h*K+k,w] = (1-wy) * in[b,c,y0,col] + wy * in[b,c,y0+1,col]        (row H contributes 0)
//     d in[b,c,y0,col]   += (1-wy) * g ;   d in[b,c,y0+1,col] += wy * g         (g = dout[b,c,h*K+k,w])
//     d y[b,k,h,w]        = [0 <= y <= H-1] * sum_c g * (in[b,c,y0+1,col] - in[b,c,y0,col])
//
// ATen's generic grid_sampler_2d_backward spends 47 ms per MM-UNet training step at bs 8 (4 atomics per
// tap-channel, plus the x-gradient nobody uses).  Here the input gradient is a GATHER: the thread of input
// element (yy, col) looks at the sources (k, h, w) that can reach it -- w is fixed by col and k (integer
// columns), h within REACH rows of yy -- and takes the lerp weight of those whose y0 or y0 + 1 is yy.
// Plain stores, no memset, bit-reproducible.  A source that lands further than REACH rows away (possible:
// the row coordinate is unbounded) is not seen by any gatherer; the d(row) kernel, which visits every
// source anyway, adds exactly those contributions with float atomics afterwards.  (First version: two
// atomics per tap-channel for everything, 7.6 ms per training step -- the top kernel of the step.)
// One thread per (b, k, h, w) / (b, yy, col) walks a slice of the channels; rows are contiguous along w.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

struct MorphArgs {
    int B, C, H, W, K, cs;  // cs = channel slices (grid.z = B * cs)
    int reach;              // gather window in rows (< 0: scatter everything, din pre-zeroed)
    int xcd_swizzle;        // gather: workgroups re-numbered so that an XCD owns a band of rows
    int zero_dy;            // gather: also clears dy (the sampler backward that follows adds into it)
    long so_b, so_c, so_h, so_k;  // element strides of out / dout (unit stride along w)
    const void *in;         // [B, C, H, W], in_t (float or bf16_t: activations under autocast)
    const float *y;         // [B, K, H, W]  row coordinate in pixels (unclamped)
    int y_parts;            // forward: y is the SUM of y_parts maps y + j * y_ps (mamba_small_fused's state-range partials)
    long y_ps;
    float *y_sum;           // forward, y_parts > 1: receives the summed map (what the backward kernels read)
    float *out;             // [B, C, H*K, W] or [C, K, B, H, W] (so_* strides)
    const float *dout;      // same layout as out
    void *din;              // [B, C, H, W], in_t
    float *dy;              // [B, K, H, W]  zero-initialised when cs > 1
    const void *din_addend; // gather: added to d input ([B, C, H, W], in_t; another consumer's gradient of the input), or null
};

// += on an element of d input (the rare far-outlier contributions): float atomics, or for bf16 a CAS loop on the
// 32-bit word that holds the element
__device__ __forceinline__ void din_add(float *p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void din_add(bf16_t *p, float v) {
    unsigned *word = reinterpret_cast<unsigned *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)3);
    const bool hi = (reinterpret_cast<uintptr_t>(p) & 2) != 0;
    unsigned old = *word, assumed;
    do {
        assumed = old;
        bf16_t cur;
        cur.bits = (uint16_t)(hi ? assumed >> 16 : assumed & 0xffffu);
        const unsigned nb = from_f32<bf16_t>(to_f32(cur) + v).bits;
        const unsigned repl = hi ? ((assumed & 0x0000ffffu) | (nb << 16)) : ((assumed & 0xffff0000u) | nb);
        old = atomicCAS(word, assumed, repl);
    } while (old != assumed);
}

__device__ __forceinline__ bool decode(const MorphArgs &p, int &b, int &k, int &h, int &w, int &c0, int &c1) {
    const int HW = p.H * p.W;
    int bx = blockIdx.x;
    if (p.xcd_swizzle) {   // per tap: an XCD owns a band of rows (the taps and neighbouring rows share input rows)
        const int nbk = HW >> 8, kk = bx / nbk, r = bx - kk * nbk;
        bx = kk * nbk + (r & 7) * (nbk >> 3) + (r >> 3);
    }
    const int pos = bx * blockDim.x + threadIdx.x;
    if (pos >= p.K * HW) return false;
    k = pos / HW;
    const int r = pos - k * HW;
    h = r / p.W;
    w = r - h * p.W;
    b = blockIdx.z / p.cs;
    const int sl = blockIdx.z - b * p.cs;
    const int per = (p.C + p.cs - 1) / p.cs;
    c0 = sl * per;
    c1 = c0 + per < p.C ? c0 + per : p.C;
    return c0 < c1;
}

template <typename in_t>
__global__ __launch_bounds__(256) void morph_sample_fwd_kernel(MorphArgs p) {
    int b, k, h, w, c0, c1;
    if (!decode(p, b, k, h, w, c0, c1)) return;
    const int HW = p.H * p.W;
    const long yi = ((long)(b * p.K + k) * p.H + h) * p.W + w;
    float yr = p.y[yi];
    if (p.y_parts > 1) {   // the row map arrives as partial sums (fixed order: bit-reproducible); slice 0 keeps the total
        for (int j = 1; j < p.y_parts; ++j) yr += p.y[yi + j * p.y_ps];
        if (c0 == 0) p.y_sum[yi] = yr;
    }
    const float yc = fminf(fmaxf(yr, 0.f), (float)(p.H - 1));
    const int y0 = (int)floorf(yc);
    const float wy = yc - (float)y0;
    const bool has1 = y0 + 1 <= p.H - 1;
    int col = w + k - p.K / 2;
    col = col < 0 ? 0 : (col > p.W - 1 ? p.W - 1 : col);
    const in_t *src = static_cast<const in_t *>(p.in) + ((long)b * p.C + c0) * HW + (long)y0 * p.W + col;
    float *dst = p.out + b * p.so_b + c0 * p.so_c + h * p.so_h + k * p.so_k + w;
    const long ostride = p.so_c;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        const float v0 = to_f32(src[0]);
        const float v1 = has1 ? to_f32(src[p.W]) : 0.f;
        *dst = fmaf(wy, v1 - v0, v0);
        src += HW;
        dst += ostride;
    }
}

template <typename in_t>
__global__ __launch_bounds__(256) void morph_sample_bwd_kernel(MorphArgs p) {
    int b, k, h, w, c0, c1;
    if (!decode(p, b, k, h, w, c0, c1)) return;
    const int HW = p.H * p.W;
    const long yi = ((long)(b * p.K + k) * p.H + h) * p.W + w;
    const float yr = p.y[yi];
    const float yc = fminf(fmaxf(yr, 0.f), (float)(p.H - 1));
    const int y0 = (int)floorf(yc);
    const float wy = yc - (float)y0;
    const bool has1 = y0 + 1 <= p.H - 1;
    int col = w + k - p.K / 2;
    col = col < 0 ? 0 : (col > p.W - 1 ? p.W - 1 : col);
    const long ioff = ((long)b * p.C + c0) * HW + (long)y0 * p.W + col;
    const in_t *src = static_cast<const in_t *>(p.in) + ioff;
    in_t *gin = static_cast<in_t *>(p.din) + ioff;
    const float *g = p.dout + b * p.so_b + c0 * p.so_c + h * p.so_h + k * p.so_k + w;
    const long ostride = p.so_c;
    // targets the gather kernel does not see (further than `reach` rows from the source row h)
    const bool far0 = p.reach < 0 || y0 - h > p.reach || h - y0 > p.reach;
    const bool far1 = has1 && (p.reach < 0 || y0 + 1 - h > p.reach || h - y0 - 1 > p.reach);
    float acc = 0.f;
    if (!far0 && !far1) {
#pragma unroll 4
        for (int c = c0; c < c1; ++c) {
            const float v0 = to_f32(src[0]);
            const float v1 = has1 ? to_f32(src[p.W]) : 0.f;
            acc = fmaf(*g, v1 - v0, acc);
            src += HW;
            g += ostride;
        }
    } else {
        for (int c = c0; c < c1; ++c) {
            const float gv = *g;
            const float v0 = to_f32(src[0]);
            const float v1 = has1 ? to_f32(src[p.W]) : 0.f;
            acc = fmaf(gv, v1 - v0, acc);
            if (far0) din_add(gin, gv * (1.f - wy));
            if (far1) din_add(gin + p.W, gv * wy);
            src += HW;
            gin += HW;
            g += ostride;
        }
    }
    // d clamp: gradient passes where 0 <= y <= H-1 (torch.clamp)
    if (!(yr >= 0.f && yr <= (float)(p.H - 1))) acc = 0.f;
    if (p.cs > 1)
        atomicAdd(p.dy + yi, acc);
    else
        p.dy[yi] = acc;
}

// d input as a gather (see the header).  grid (ceil(H*W / 256), 1, B * ceil(C / CS)); thread = (yy, col).
template <int CS, typename in_t>
__global__ __launch_bounds__(256) void morph_gather_din_kernel(MorphArgs p) {
    // Workgroups go round-robin over the 8 XCDs, each with its own L2; a pixel reads the gradient rows yy - 2 .. yy + 2, so
    // with the natural order every XCD fetches (nearly) every row.  Re-numbered so that an XCD owns a contiguous band of
    // rows, the 5x row reuse stays inside one L2.
    int bx = blockIdx.x;
    if (p.xcd_swizzle) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    const int pos = bx * blockDim.x + threadIdx.x;
    const int HW = p.H * p.W;
    if (pos >= HW) return;
    const int yy = pos / p.W, col = pos - yy * p.W;
    const int nsl = (p.C + CS - 1) / CS;
    const int b = blockIdx.z / nsl;
    const int c0 = (blockIdx.z - b * nsl) * CS;
    const int nc = p.C - c0 < CS ? p.C - c0 : CS;
    float acc[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) acc[c] = 0.f;
    if (p.din_addend) {   // a residual connection's gradient: the sums start from it (its loads are in flight during the
                          // coordinate work below; added at the end they were 13 us of exposed latency per call)
        const in_t *ad = static_cast<const in_t *>(p.din_addend) + ((long)b * p.C + c0) * HW + pos;
#pragma unroll
        for (int c = 0; c < CS; ++c)
            if (c < nc) acc[c] = to_f32(ad[(long)c * HW]);
    }
    const int hlo = yy - p.reach > 0 ? yy - p.reach : 0;
    const int hhi = yy + p.reach < p.H - 1 ? yy + p.reach : p.H - 1;
    const long ostride = p.so_c;
    for (int k = 0; k < p.K; ++k) {
        // source columns w with clamp(w + k - K/2, 0, W-1) == col
        const int we = col - k + p.K / 2;
        int wlo = col == 0 ? 0 : we, whi = col == p.W - 1 ? p.W - 1 : we;
        wlo = wlo < 0 ? 0 : wlo;
        whi = whi > p.W - 1 ? p.W - 1 : whi;
        for (int w = wlo; w <= whi; ++w) {
            const float *yp = p.y + ((long)(b * p.K + k) * p.H + hlo) * p.W + w;
            const float *gp = p.dout + b * p.so_b + c0 * p.so_c + hlo * p.so_h + k * p.so_k + w;
            for (int h = hlo; h <= hhi; ++h, yp += p.W, gp += p.so_h) {
                const float yc = fminf(fmaxf(*yp, 0.f), (float)(p.H - 1));
                const int y0 = (int)floorf(yc);
                const float wy = yc - (float)y0;
                const float coef = y0 == yy ? 1.f - wy : (y0 + 1 == yy ? wy : 0.f);
                if (coef != 0.f) {
                    if (nc == CS) {
#pragma unroll
                        for (int c = 0; c < CS; ++c) acc[c] = fmaf(coef, gp[c * ostride], acc[c]);
                    } else {
#pragma unroll
                        for (int c = 0; c < CS; ++c)
                            if (c < nc) acc[c] = fmaf(coef, gp[c * ostride], acc[c]);
                    }
                }
            }
        }
    }
    in_t *dst = static_cast<in_t *>(p.din) + ((long)b * p.C + c0) * HW + pos;
#pragma unroll
    for (int c = 0; c < CS; ++c)
        if (c < nc) dst[(long)c * HW] = from_f32<in_t>(acc[c]);
    // d(row) is summed with float atomics by the channel slices of the kernel that runs next: its zero fill rides along
    // here (slice 0), one launch less per MMConv backward
    if (p.zero_dy && c0 == 0)
        for (int k = 0; k < p.K; ++k) p.dy[((long)(b * p.K + k) * p.H + yy) * p.W + col] = 0.f;
}

int channel_slices(int B, int C, int positions) {
    // enough threads to fill the chip, but never fewer than 8 channels per slice
    static const long target = []() { const char *e = getenv("MMU_MORPH_SLICE_THREADS"); return e ? atol(e) : 131072L; }();
    long threads = (long)B * positions;
    int cs = 1;
    while (threads * cs < target && C / (cs * 2) >= 8) cs *= 2;
    return cs;
}

// sampler kernels: blocks per tap a multiple of 8 (and enough of them) for the per-tap XCD banding of decode()
int sampler_swizzle(const MorphArgs &a) {
    static const bool on = []() { const char *e = getenv("MMU_MORPH_XCD"); return !e || e[0] != '0'; }();
    const long HW = (long)a.H * a.W;
    return (on && HW % 2048 == 0 && HW / 256 >= 64) ? 1 : 0;
}

void set_out_strides(MorphArgs &a, int layout) {
    const long HW = (long)a.H * a.W;
    if (layout == MMU_MORPH_TOKENS_LAST) {  // [C][K][B][H][W]
        a.so_c = (long)a.K * a.B * HW; a.so_k = (long)a.B * HW; a.so_b = HW; a.so_h = a.W;
    } else {                                // [B][C][H*K][W]
        a.so_b = (long)a.C * a.K * HW; a.so_c = (long)a.K * HW; a.so_h = (long)a.K * a.W; a.so_k = a.W;
    }
}

int check(const mmu_morph_params *p, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->batch > 0 && p->channels > 0 && p->height > 1 && p->width > 0 && p->taps > 0 && (p->taps & 1),
              "%s: need batch, channels > 0, height >= 2, odd number of taps", name);
    MMU_CHECK((long)p->batch * p->channels * p->height * p->taps * p->width < (1L << 31),
              "%s: tensor too large for 32-bit positions", name);
    MMU_CHECK(p->out_layout == MMU_MORPH_BCHW || p->out_layout == MMU_MORPH_TOKENS_LAST, "%s: unknown out_layout %d",
              name, p->out_layout);
    MMU_CHECK(p->in_dtype == MMU_DTYPE_F32 || p->in_dtype == MMU_DTYPE_BF16, "%s: unsupported in_dtype %d", name,
              p->in_dtype);
    return 0;
}

}  // namespace

extern "C" int mmu_morph_sample_fwd(const mmu_morph_params *p, void *stream) {
    if (int r = check(p, "morph_sample_fwd")) return r;
    MMU_CHECK(p->input && p->y && p->out, "morph_sample_fwd: input, y, out are required");
    MorphArgs a = {};
    a.B = p->batch; a.C = p->channels; a.H = p->height; a.W = p->width; a.K = p->taps;
    a.in = p->input; a.y = p->y; a.out = p->out;
    a.y_parts = p->y_parts > 1 ? p->y_parts : 1;
    a.y_ps = (long)a.B * a.K * a.H * a.W;
    a.y_sum = p->y_sum;
    MMU_CHECK(a.y_parts == 1 || a.y_sum != nullptr, "morph_sample_fwd: y_sum is required when y comes in parts");
    set_out_strides(a, p->out_layout);
    const int positions = a.K * a.H * a.W;
    a.cs = channel_slices(a.B, a.C, positions);
    dim3 grid((positions + 255) / 256, 1, a.B * a.cs);
    a.xcd_swizzle = 0;   // (banding measured on the forward: no change -- it is bound by its K-times larger output)
    if (p->in_dtype == MMU_DTYPE_BF16)
        morph_sample_fwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    else
        morph_sample_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    MMU_HIP_LAUNCH_CHECK("morph_sample_fwd");
    return 0;
}

extern "C" int mmu_morph_sample_bwd(const mmu_morph_params *p, void *stream) {
    if (int r = check(p, "morph_sample_bwd")) return r;
    MMU_CHECK(p->input && p->y && p->dout && p->dinput && p->dy,
              "morph_sample_bwd: input, y, dout, dinput, dy are required");
    MorphArgs a = {};
    a.B = p->batch; a.C = p->channels; a.H = p->height; a.W = p->width; a.K = p->taps;
    a.in = p->input; a.y = p->y; a.dout = p->dout; a.din = p->dinput; a.dy = p->dy;
    a.din_addend = p->dinput_addend;
    set_out_strides(a, p->out_layout);
    const int positions = a.K * a.H * a.W;
    a.cs = channel_slices(a.B, a.C, positions);
    hipStream_t st = (hipStream_t)stream;
    a.reach = 2;  // rows; anything further goes through the atomics of the d(row) kernel
    {
        constexpr int CS = 16;
        dim3 gg((a.H * a.W + 255) / 256, 1, a.B * ((a.C + CS - 1) / CS));
        static const bool swz = []() { const char *e = getenv("MMU_MORPH_XCD"); return !e || e[0] != '0'; }();
        a.xcd_swizzle = (swz && gg.x % 8 == 0 && gg.x >= 64) ? 1 : 0;
        a.zero_dy = a.cs > 1 ? 1 : 0;
        if (p->in_dtype == MMU_DTYPE_BF16)
            morph_gather_din_kernel<CS, bf16_t><<<gg, 256, 0, st>>>(a);
        else
            morph_gather_din_kernel<CS, float><<<gg, 256, 0, st>>>(a);
        MMU_HIP_LAUNCH_CHECK("morph_gather_din");
    }
    a.zero_dy = 0;
    dim3 grid((positions + 255) / 256, 1, a.B * a.cs);
    a.xcd_swizzle = sampler_swizzle(a);
    if (p->in_dtype == MMU_DTYPE_BF16)
        morph_sample_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>(a);
    else
        morph_sample_bwd_kernel<float><<<grid, 256, 0, st>>>(a);
    MMU_HIP_LAUNCH_CHECK("morph_sample_bwd");
    return 0;
}
