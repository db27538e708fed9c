// resize.hip -- bilinear resize with align_corners=True (F.interpolate / nn.Upsample as MM-UNet uses them:
// MMUNet.py:384 DecoderBlock x2, :362 RCG edge map, :571-575 side outputs to the input size) for gfx950.
//
//   sy = oy * (H-1)/(OH-1)   (0 when OH == 1);  y0 = floor(sy), y1 = min(y0+1, H-1), wy = sy - y0;  same in x
//   out[b,c,oy,ox] = (1-wy)(1-wx) in[y0,x0] + (1-wy) wx in[y0,x1] + wy (1-wx) in[y1,x0] + wy wx in[y1,x1]
//
// ATen's upsample_bilinear2d_out_frame takes 443 us for [8,64,128,128] -> 256x256 on MI355X (one thread per
// output pixel looping over batch and channels with 64-bit index arithmetic); this is a plain streaming
// kernel: thread = 4 consecutive output x of one (plane, oy), 16-byte stores.  Backward is a GATHER (each
// input pixel sums the outputs whose footprint covers it; the source ranges follow from monotonic sy, sx),
// so it needs neither atomics nor a zeroed buffer and is bit-reproducible (ATen: float atomics).
#include <type_traits>
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

struct ResizeArgs {
    int planes, H, W, OH, OW;
    float ry, rx;    // (H-1)/(OH-1), (W-1)/(OW-1)
    float iry, irx;  // their reciprocals (0 where the ratio is 0)
    const void *in;     // io_t (float or bf16_t), all four
    void *out;
    const void *dout;
    void *din;
    const void *addend;   // bwd: added to the gathered gradient (may be din itself)
};

__device__ __forceinline__ void tap(int o, float r, int n, int &i0, int &i1, float &w) {
    const float s = r * (float)o;
    i0 = (int)s;  // s >= 0
    i0 = i0 > n - 1 ? n - 1 : i0;
    i1 = i0 + 1 < n ? i0 + 1 : n - 1;
    w = s - (float)i0;
}

// block (bx, 256 / bx), bx = 8 ... 64 lanes along x by the output width: a thread = 4 consecutive output x of one row
// (plane, oy); at bx = 64 a WAVE owns the row -- its y tap and row pointers are wave-uniform -- and the row index is one
// 32-bit division where the flat form paid four 64-bit ones per thread.
template <typename io_t>
__global__ __launch_bounds__(256) void resize_fwd_kernel(ResizeArgs p) {
    const unsigned row = blockIdx.x * blockDim.y + threadIdx.y;        // (plane, oy); host: planes * OH < 2^32
    if (row >= (unsigned)p.planes * (unsigned)p.OH) return;
    const unsigned plane = row / (unsigned)p.OH;
    const int oy = (int)(row - plane * (unsigned)p.OH);
    const int q = blockIdx.y * blockDim.x + threadIdx.x;
    if (q * 4 >= p.OW) return;
    int y0, y1;
    float wy;
    tap(oy, p.ry, p.H, y0, y1, wy);
    const io_t *in = static_cast<const io_t *>(p.in);
    const io_t *r0 = in + ((long)plane * p.H + y0) * p.W, *r1 = in + ((long)plane * p.H + y1) * p.W;
    float v[4];
    int x0[4], x1[4];
    float wx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ox = q * 4 + j;
        tap(ox < p.OW ? ox : p.OW - 1, p.rx, p.W, x0[j], x1[j], wx[j]);
    }
    // the four outputs read a short run of consecutive input columns (4 when up-sampling, 8 when halving): one or two
    // 16-byte (unaligned) loads per row instead of eight dword gathers strided across the wave
    const int span = x1[3] - x0[0];
    if (std::is_same<io_t, float>::value && p.W >= 8 && span <= 7) {
        const int wide = span > 3;                       // wave-uniform in practice (the ratio decides)
        const int base = min(x0[0], p.W - (wide ? 8 : 4));
        const float *f0 = reinterpret_cast<const float *>(r0) + base, *f1 = reinterpret_cast<const float *>(r1) + base;
        float t0[8], t1[8];
        __builtin_memcpy(t0, f0, 16);
        __builtin_memcpy(t1, f1, 16);
        if (wide) {
            __builtin_memcpy(t0 + 4, f0 + 4, 16);
            __builtin_memcpy(t1 + 4, f1 + 4, 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i0 = x0[j] - base, i1 = x1[j] - base;
                float a0 = t0[0], a1 = t0[0], b0 = t1[0], b1 = t1[0];
#pragma unroll
                for (int k = 1; k < 8; ++k) {
                    a0 = i0 == k ? t0[k] : a0;
                    a1 = i1 == k ? t0[k] : a1;
                    b0 = i0 == k ? t1[k] : b0;
                    b1 = i1 == k ? t1[k] : b1;
                }
                const float a = fmaf(wx[j], a1 - a0, a0), b = fmaf(wx[j], b1 - b0, b0);
                v[j] = fmaf(wy, b - a, a);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i0 = x0[j] - base, i1 = x1[j] - base;
                float a0 = t0[0], a1 = t0[0], b0 = t1[0], b1 = t1[0];
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    a0 = i0 == k ? t0[k] : a0;
                    a1 = i1 == k ? t0[k] : a1;
                    b0 = i0 == k ? t1[k] : b0;
                    b1 = i1 == k ? t1[k] : b1;
                }
                const float a = fmaf(wx[j], a1 - a0, a0), b = fmaf(wx[j], b1 - b0, b0);
                v[j] = fmaf(wy, b - a, a);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a0 = to_f32(r0[x0[j]]), b0 = to_f32(r1[x0[j]]);
            const float a = fmaf(wx[j], to_f32(r0[x1[j]]) - a0, a0);
            const float b = fmaf(wx[j], to_f32(r1[x1[j]]) - b0, b0);
            v[j] = fmaf(wy, b - a, a);
        }
    }
    io_t *dst = static_cast<io_t *>(p.out) + (long)row * p.OW + q * 4;
    if ((p.OW & 3) == 0) {
        store_k<io_t, 4, true>(dst, 4, true, v);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < p.OW) dst[j] = from_f32<io_t>(v[j]);
    }
}

// outputs o with floor(r*o) in {i-1, i} contribute to input i: the range [lo, hi] of such o (r > 0), or all
// o when r == 0 (single input row/col; only i == 0 exists then)
__device__ __forceinline__ void src_range(int i, float r, float inv_r, int n_out, int &lo, int &hi) {
    if (r <= 0.f) {
        lo = 0;
        hi = n_out - 1;
        return;
    }
    // o with i-1 <= r*o < i+1, widened by one on each side against rounding (and the reciprocal's); the caller verifies
    // each candidate through its taps
    lo = (int)ceilf((float)(i - 1) * inv_r) - 1;
    hi = (int)ceilf((float)(i + 1) * inv_r);
    lo = lo < 0 ? 0 : lo;
    hi = hi > n_out - 1 ? n_out - 1 : hi;
}

template <typename io_t>
__global__ __launch_bounds__(256) void resize_bwd_kernel(ResizeArgs p) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)p.planes * p.H * p.W;
    if (idx >= total) return;
    const int x = (int)(idx % p.W);
    const long r = idx / p.W;
    const int y = (int)(r % p.H);
    const long plane = r / p.H;
    int ylo, yhi, xlo, xhi;
    src_range(y, p.ry, p.iry, p.OH, ylo, yhi);
    src_range(x, p.rx, p.irx, p.OW, xlo, xhi);
    // separable: the x weights do not depend on oy -- keep up to 8 of them in registers (x2 upsampling has 6)
    constexpr int XC = 8;
    const int nx = xhi - xlo + 1;
    float cxs[XC];
    if (nx <= XC) {
#pragma unroll
        for (int j = 0; j < XC; ++j) {
            int x0, x1;
            float wx;
            const int ox = xlo + j <= xhi ? xlo + j : xhi;
            tap(ox, p.rx, p.W, x0, x1, wx);
            const float cx = (x0 == x ? 1.f - wx : 0.f) + (x1 == x ? wx : 0.f);
            cxs[j] = xlo + j <= xhi ? cx : 0.f;
        }
    }
    float acc = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
        int y0, y1;
        float wy;
        tap(oy, p.ry, p.H, y0, y1, wy);
        const float cy = (y0 == y ? 1.f - wy : 0.f) + (y1 == y ? wy : 0.f);
        if (cy == 0.f) continue;
        const io_t *g = static_cast<const io_t *>(p.dout) + (plane * p.OH + oy) * p.OW;
        float row = 0.f;
        if (nx <= XC) {
#pragma unroll
            for (int j = 0; j < XC; ++j) {
                const int ox = xlo + j <= xhi ? xlo + j : xhi;
                row = fmaf(cxs[j], to_f32(g[ox]), row);
            }
        } else {
            for (int ox = xlo; ox <= xhi; ++ox) {
                int x0, x1;
                float wx;
                tap(ox, p.rx, p.W, x0, x1, wx);
                const float cx = (x0 == x ? 1.f - wx : 0.f) + (x1 == x ? wx : 0.f);
                row = fmaf(cx, to_f32(g[ox]), row);
            }
        }
        acc = fmaf(cy, row, acc);
    }
    if (p.addend) acc += to_f32(static_cast<const io_t *>(p.addend)[idx]);
    static_cast<io_t *>(p.din)[idx] = from_f32<io_t>(acc);
}

// Large up-sampling ratios on few pixels (the side outputs: 1 channel, 16 x 16 ... 128 x 128 -> 512 x 512): an input pixel
// gathers from (2 x ratio)^2 outputs -- 4,096 at x32 -- and there are only a few thousand pixels, so one THREAD per pixel
// (the kernel above) is a handful of waves in long loops: 51 us for 8 MB of gradient.  Here a WAVE owns the pixel, its
// lanes stride over the candidate rectangle (contiguous along x) and the sum is a wave reduction in a fixed order.
template <typename io_t>
__global__ __launch_bounds__(256) void resize_bwd_wave_kernel(ResizeArgs p) {
    const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (idx >= (long)p.planes * p.H * p.W) return;
    const int x = (int)(idx % p.W);
    const long r = idx / p.W;
    const int y = (int)(r % p.H);
    const long plane = r / p.H;
    int ylo, yhi, xlo, xhi;
    src_range(y, p.ry, p.iry, p.OH, ylo, yhi);
    src_range(x, p.rx, p.irx, p.OW, xlo, xhi);
    const int nx = xhi - xlo + 1, total = (yhi - ylo + 1) * nx;
    const io_t *g = static_cast<const io_t *>(p.dout) + plane * p.OH * p.OW;
    float acc = 0.f;
    for (int e = lane; e < total; e += 64) {
        const int ey = e / nx, oy = ylo + ey, ox = xlo + (e - ey * nx);
        int a0, a1, b0, b1;
        float wx, wy;
        tap(oy, p.ry, p.H, b0, b1, wy);
        tap(ox, p.rx, p.W, a0, a1, wx);
        const float cy = (b0 == y ? 1.f - wy : 0.f) + (b1 == y ? wy : 0.f);
        const float cx = (a0 == x ? 1.f - wx : 0.f) + (a1 == x ? wx : 0.f);
        acc = fmaf(cy * cx, to_f32(g[(long)oy * p.OW + ox]), acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        if (p.addend) acc += to_f32(static_cast<const io_t *>(p.addend)[idx]);
        static_cast<io_t *>(p.din)[idx] = from_f32<io_t>(acc);
    }
}

// The same gather through LDS, for the ratios the model uses (x2 up-sampling; the 256 x 256 edge map down to 128 / 64 /
// 32).  A thread of the kernel above reads every candidate output itself: 4-6 rows x 8 columns of scalar loads per input
// pixel -- the x2 up-sampling to 512 x 512 of 64 channels took 245 us for 671 MB (bound by load issue, not HBM).  Here a
// workgroup owns 64 x TY input pixels of a plane, loads the outputs their footprints cover once (coalesced along x) and
// gathers from LDS.  Round 3: the interpolation is separable, so the workgroup first writes, per input column and per
// input row of its tile, the list of outputs with a NON-ZERO coefficient (tile-relative index + coefficient, at most NZ
// of them: 2 when down-sampling, 5 for x2); a pixel is then NZ_y x NZ_x LDS reads and FMAs.  Before, every thread
// derived its 4-8 column and 2 x 4-8 row candidates itself (divisions, float->int conversions, ~150 instructions per
// pixel: 170 us for the 134 MB gradient of the edge map, 8x its write time).
constexpr int RT_TX = 64, RT_MAXR = 26, RT_MAXC = 152;

template <typename io_t, int NZ, int TY>
__global__ __launch_bounds__(256) void resize_bwd_tiled_kernel(ResizeArgs p) {
    __shared__ float tile[RT_MAXR][RT_MAXC + 1];
    __shared__ float xco[RT_TX][NZ], yco[TY][NZ];
    __shared__ int xix[RT_TX][NZ], yix[TY][NZ], ycnt[TY];
    const int x0 = blockIdx.x * RT_TX, y0 = blockIdx.y * TY;
    const long plane = blockIdx.z;
    const int xl = min(x0 + RT_TX, p.W) - 1, yl = min(y0 + TY, p.H) - 1;       // last input pixel of the tile
    int txlo, txhi, tylo, tyhi, t0, t1;
    src_range(x0, p.rx, p.irx, p.OW, txlo, t0);
    src_range(xl, p.rx, p.irx, p.OW, t1, txhi);
    src_range(y0, p.ry, p.iry, p.OH, tylo, t0);
    src_range(yl, p.ry, p.iry, p.OH, t1, tyhi);
    const int cols = txhi - txlo + 1, rows = tyhi - tylo + 1;                     // host: <= RT_MAXC, RT_MAXR
    const io_t *g = static_cast<const io_t *>(p.dout) + (plane * p.OH + tylo) * p.OW + txlo;
    for (int e = threadIdx.x; e < rows * cols; e += 256) {
        const int r = e / cols, c = e - r * cols;
        tile[r][c] = to_f32(g[(long)r * p.OW + c]);
    }
    if (threadIdx.x < RT_TX + TY) {                  // wave 0: the columns' lists; the first TY threads of wave 1: the rows'
        const bool isx = threadIdx.x < RT_TX;
        const int t = isx ? threadIdx.x : threadIdx.x - RT_TX;
        const int i = (isx ? x0 : y0) + t;
        const int n_in = isx ? p.W : p.H, n_out = isx ? p.OW : p.OH, origin = isx ? txlo : tylo;
        const float r = isx ? p.rx : p.ry, ir = isx ? p.irx : p.iry;
        float *co = isx ? xco[t] : yco[t];
        int *ix = isx ? xix[t] : yix[t];
        int n = 0;
        if (i < n_in) {
            int lo, hi;
            src_range(i, r, ir, n_out, lo, hi);
            for (int o = lo; o <= hi; ++o) {
                int a0, a1;
                float w;
                tap(o, r, n_in, a0, a1, w);
                const float c = (a0 == i ? 1.f - w : 0.f) + (a1 == i ? w : 0.f);
                if (c != 0.f && n < NZ) {            // (host: NZ bounds the outputs that can reach one input)
                    co[n] = c;
                    ix[n] = o - origin;
                    ++n;
                }
            }
        }
        if (!isx) ycnt[t] = n;
        for (; n < NZ; ++n) {
            co[n] = 0.f;
            ix[n] = 0;
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, x = x0 + tx;
    if (x >= p.W) return;
    float cx[NZ];
    int jx[NZ];
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
        cx[j] = xco[tx][j];
        jx[j] = xix[tx][j];
    }
    const io_t *add = static_cast<const io_t *>(p.addend);
#pragma unroll
    for (int k = 0; k < TY / 4; ++k) {
        const int ty = (threadIdx.x >> 6) + 4 * k, y = y0 + ty;
        if (y >= p.H) continue;
        const int ny = ycnt[ty];                     // (the same row for the whole wave: a uniform loop)
        float acc = 0.f;
        for (int j = 0; j < ny; ++j) {
            const float *row = tile[yix[ty][j]];
            float rs = 0.f;
#pragma unroll
            for (int i = 0; i < NZ; ++i) rs = fmaf(cx[i], row[jx[i]], rs);
            acc = fmaf(yco[ty][j], rs, acc);
        }
        const long o = (plane * p.H + y) * p.W + x;
        if (add) acc += to_f32(add[o]);
        static_cast<io_t *>(p.din)[o] = from_f32<io_t>(acc);
    }
}

// how many outputs can reach one input pixel along an axis: the multiples of r inside a half-open interval of length 2
inline int resize_bwd_reach(float r) { return (int)floorf(2.f / r + 1e-3f) + 1; }
inline bool resize_bwd_tiled_ok(const ResizeArgs &a, int ty) {
    if (a.rx <= 0.f || a.ry <= 0.f || a.planes > 65535) return false;
    const int cols = (int)ceilf((RT_TX + 1) / a.rx) + 3, rows = (int)ceilf((ty + 1) / a.ry) + 3;
    return cols <= RT_MAXC && rows <= RT_MAXR;
}

template <typename io_t>
bool resize_bwd_tiled_launch(const ResizeArgs &a, hipStream_t st) {
    if (a.rx <= 0.f || a.ry <= 0.f) return false;
    const int reach = max(resize_bwd_reach(a.rx), resize_bwd_reach(a.ry));
    const int ty = reach <= 2 && resize_bwd_tiled_ok(a, 32) ? 32 : 8;
    if (!resize_bwd_tiled_ok(a, ty) || reach > 8) return false;
    dim3 grid((a.W + RT_TX - 1) / RT_TX, (a.H + ty - 1) / ty, a.planes);
    if (reach <= 2 && ty == 32) resize_bwd_tiled_kernel<io_t, 2, 32><<<grid, 256, 0, st>>>(a);
    else if (reach <= 2) resize_bwd_tiled_kernel<io_t, 2, 8><<<grid, 256, 0, st>>>(a);
    else if (reach <= 5) resize_bwd_tiled_kernel<io_t, 5, 8><<<grid, 256, 0, st>>>(a);
    else resize_bwd_tiled_kernel<io_t, 8, 8><<<grid, 256, 0, st>>>(a);
    return true;
}

int fill(const mmu_resize_params *p, ResizeArgs &a, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->planes > 0 && p->in_h > 0 && p->in_w > 0 && p->out_h > 0 && p->out_w > 0, "%s: empty tensor", name);
    a.planes = p->planes; a.H = p->in_h; a.W = p->in_w; a.OH = p->out_h; a.OW = p->out_w;
    MMU_CHECK(p->dtype == MMU_DTYPE_F32 || p->dtype == MMU_DTYPE_BF16, "%s: unsupported dtype %d", name, p->dtype);
    a.ry = a.OH > 1 ? (float)(a.H - 1) / (float)(a.OH - 1) : 0.f;
    a.rx = a.OW > 1 ? (float)(a.W - 1) / (float)(a.OW - 1) : 0.f;
    a.iry = a.ry > 0.f ? 1.f / a.ry : 0.f;
    a.irx = a.rx > 0.f ? 1.f / a.rx : 0.f;
    return 0;
}

}  // namespace

extern "C" int mmu_bilinear_resize_fwd(const mmu_resize_params *p, void *stream) {
    ResizeArgs a = {};
    if (int r = fill(p, a, "bilinear_resize_fwd")) return r;
    MMU_CHECK(p->input && p->out, "bilinear_resize_fwd: input and out are required");
    a.in = p->input; a.out = p->out;
    const long rows = (long)a.planes * a.OH;
    MMU_CHECK(rows < (1L << 32), "bilinear_resize_fwd: too many output rows");
    const int xq = (a.OW + 3) / 4;
    MMU_CHECK((xq + 63) / 64 <= 65535, "bilinear_resize_fwd: output too wide");
    int bx = 8;
    while (bx < 64 && bx < xq) bx *= 2;
    const dim3 block(bx, 256 / bx), grid((unsigned)((rows + block.y - 1) / block.y), (unsigned)((xq + bx - 1) / bx));
    if (p->dtype == MMU_DTYPE_BF16)
        resize_fwd_kernel<bf16_t><<<grid, block, 0, (hipStream_t)stream>>>(a);
    else
        resize_fwd_kernel<float><<<grid, block, 0, (hipStream_t)stream>>>(a);
    MMU_HIP_LAUNCH_CHECK("bilinear_resize_fwd");
    return 0;
}

extern "C" int mmu_bilinear_resize_bwd(const mmu_resize_params *p, void *stream) {
    ResizeArgs a = {};
    if (int r = fill(p, a, "bilinear_resize_bwd")) return r;
    MMU_CHECK(p->dout && p->dinput, "bilinear_resize_bwd: dout and dinput are required");
    a.dout = p->dout; a.din = p->dinput;
    const long total = (long)a.planes * a.H * a.W;
    a.addend = p->dinput_addend;
    static const bool tiled_on = []() { const char *e = getenv("MMU_RESIZE_BWD_TILED"); return !e || e[0] != '0'; }();
    hipStream_t st = (hipStream_t)stream;
    bool done = false;
    if (tiled_on) done = p->dtype == MMU_DTYPE_BF16 ? resize_bwd_tiled_launch<bf16_t>(a, st) : resize_bwd_tiled_launch<float>(a, st);
    // more than 64 candidates per pixel and few enough pixels that a wave each still fills the chip
    if (!done && a.rx > 0.f && a.ry > 0.f && 4.f / (a.rx * a.ry) >= 64.f && total <= (1L << 20)) {
        const unsigned blocks = (unsigned)((total + 3) / 4);
        if (p->dtype == MMU_DTYPE_BF16) resize_bwd_wave_kernel<bf16_t><<<blocks, 256, 0, st>>>(a);
        else resize_bwd_wave_kernel<float><<<blocks, 256, 0, st>>>(a);
        done = true;
    }
    if (!done) {
        if (p->dtype == MMU_DTYPE_BF16) resize_bwd_kernel<bf16_t><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a);
        else resize_bwd_kernel<float><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a);
    }
    MMU_HIP_LAUNCH_CHECK("bilinear_resize_bwd");
    return 0;
}
