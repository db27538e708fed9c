// morph_coords.hip -- the small-tensor glue of MMConv around its K-channel Mamba, fused (gfx950).
//
// Reference (src/UM_Net/MMUNet.py:122-193 + requirements/mamba_simple.py:201-205,365): per MMConv block
//   y_off = offset[:, :K]                                   -> two-row zig-zag flatten (:68-93) -> (B, L, K)
//   xz    = in_proj(y_off tokens)                            (K -> 4K channels)
//   ...   conv1d / x_proj / dt_proj / selective scan ...     (kept as they are: HIP kernels + GEMMs)
//   seq   = out_proj(out_z)                                  (2K -> K channels)  -> inverse zig-zag (:95-121)
//   y     = clamp(softplus(altho), 0.01) * seq + row_index + extend_scope * cumsum-from-centre(y_off)
// which PyTorch runs as ~30 tiny kernels forward and as many backward (reshape copies, 3-channel GEMMs with
// a B*L-long reduction for their weight gradients, a dozen elementwise ops).  Here:
//
//   A  zigzag_inproj   fwd: xz[j][b][l]  = sum_k Win[j][k] * off[b,k,h(l),w(l)]                  (tokens-last)
//                      bwd: d off[b,k,h,w] = sum_j Win[j][k] * dxz[j][b][l] ;  dWin[j][k] = sum dxz * off
//   B  coords_outproj  fwd: y[b,k,h,w] = wgt * sum_d Wout[k][d] * oz[d][b][l] + h + scope * cum_k(off)
//                      bwd: d oz, dWout, d altho (through wgt = max(softplus(altho), 0.01)), d off (cum part)
//
// l = zig-zag token of pixel (h, w): rows are paired, a pair is walked column by column
// (2p,w),(2p+1,w),(2p,w+1),...; an odd last row is appended row-major.
// K (taps) is a template parameter (1 and 3 occur in MM-UNet); d_inner = 2K, in_proj has 4K rows.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

struct CoordArgs {
    int B, H, W;
    float scope;
    const float *off;    // [B, 2K, H, W]
    const float *win;    // [4K, K]
    const float *wout;   // [K, 2K]
    const float *altho;  // scalar
    float *xz;           // [4K][B][L]
    const float *dxz;    // [4K][B][L]
    const float *oz;     // [2K][B][L]
    float *y;            // [B, K, H, W]
    const float *dy;     // [B, K, H, W]
    float *doff;         // [B, 2K, H, W]  (channels K..2K-1 receive 0)
    float *dwin;         // [4K, K]   zeroed by the launcher, float atomics (one per block and entry)
    float *doz;          // [2K][B][L]
    float *dwout;        // [K, 2K]   zeroed, atomics
    float *daltho;       // scalar    zeroed, atomics
    int acc_doff;        // zigzag_inproj_bwd: add to doff instead of writing it
    float *ws;           // NULL, or per-block partial rows [gridDim.x][NV4]: plain stores, summed in a fixed order afterwards
};

__device__ __forceinline__ int zig_of(int h, int w, int H, int W) {
    const int He = H & ~1;
    return h < He ? (h >> 1) * (2 * W) + 2 * w + (h & 1) : He * W + w;
}
__device__ __forceinline__ void unzig(int l, int H, int W, int &h, int &w) {
    const int He = H & ~1;
    if (l < He * W) {
        const int p = l / (2 * W), r = l - p * 2 * W;
        h = 2 * p + (r & 1);
        w = r >> 1;
    } else {
        h = He;
        w = l - He * W;
    }
}

// Block-wide sums of NV per-thread values at once (256 threads), added to dst[0..NV) with one float atomic
// each.  Four values share one wave reduction (wave_sum4), the four waves meet in LDS once: one barrier for
// all NV values.  (One block_sum per value -- 36 of them for the in_proj weight gradient, two barriers each --
// made these kernels 20 us for a single workgroup.)  scale: optional factor for the LAST value (NV - 1).
template <int NV>
__device__ __forceinline__ void block_sum_many_atomic(const float (&v)[NV], float *lds /* [4][NV4] */, float *dst,
                                                      int ndst, float last_scale = 1.f) {
    constexpr int NV4 = (NV + 3) & ~3;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV4; i += 4) {
        const float r = wave_sum4_swap(v[i], i + 1 < NV ? v[i + 1] : 0.f, i + 2 < NV ? v[i + 2] : 0.f,
                                  i + 3 < NV ? v[i + 3] : 0.f);
        if (lane >= 12 && lane < 16) lds[w * NV4 + i + lane - 12] = r;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ndst; i += blockDim.x) {
        float s = lds[i] + lds[NV4 + i] + lds[2 * NV4 + i] + lds[3 * NV4 + i];
        if (i == NV - 1) s *= last_scale;
        atomicAdd(dst + i, s);
    }
}

// The same sums written to this block's row of a partial buffer (plain stores: no zero fill, no atomics; the rows are
// added in a fixed order by coords_partials_sum_kernel / the deferred reduction).  All NV values are stored.
template <int NV>
__device__ __forceinline__ void block_sum_many_store(const float (&v)[NV], float *lds /* [4][NV4] */, float *row,
                                                     float last_scale = 1.f) {
    constexpr int NV4 = (NV + 3) & ~3;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV4; i += 4) {
        const float r = wave_sum4_swap(v[i], i + 1 < NV ? v[i + 1] : 0.f, i + 2 < NV ? v[i + 2] : 0.f,
                                  i + 3 < NV ? v[i + 3] : 0.f);
        if (lane >= 12 && lane < 16) lds[w * NV4 + i + lane - 12] = r;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NV; i += blockDim.x) {
        float s = lds[i] + lds[NV4 + i] + lds[2 * NV4 + i] + lds[3 * NV4 + i];
        if (i == NV - 1) s *= last_scale;
        row[i] = s;
    }
}

// dst[i] = sum over the rows of part[r][i] (i < n1), dst2[i - n1] = the same for the values n1 .. ntot - 1:
// 16 row groups in parallel, each a strided serial sum, the 16 group sums added in order.  One workgroup per 64 values.
__global__ __launch_bounds__(1024) void coords_partials_sum_kernel(const float *__restrict__ part, float *__restrict__ dst,
                                                                   float *__restrict__ dst2, int n1, int nparts,
                                                                   int stride, int ntot) {
    __shared__ float sums[16][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;
    float s = 0.f;
    if (i < ntot)
        for (int k = g; k < nparts; k += 16) s += part[(long)k * stride + i];
    sums[g][o] = s;
    __syncthreads();
    if (g == 0 && i < ntot) {
        float t = sums[0][o];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += sums[k][o];
        if (i < n1)
            dst[i] = t;
        else if (dst2 != nullptr)
            dst2[i - n1] = t;
    }
}

// block-wide sum of v (256 threads); result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float *red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---- A forward: one thread per token (b, l) ------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void zigzag_inproj_fwd_kernel(CoordArgs p) {
    const int L = p.H * p.W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)p.B * L) return;
    const int b = (int)(idx / L), l = (int)(idx - (long)b * L);
    int h, w;
    unzig(l, p.H, p.W, h, w);
    float x[K];
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = p.off[(((long)b * 2 * K + k) * p.H + h) * p.W + w];
#pragma unroll
    for (int j = 0; j < 4 * K; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) acc = fmaf(p.win[j * K + k], x[k], acc);
        p.xz[((long)j * p.B + b) * L + l] = acc;
    }
}

// ---- A backward: grid-stride over tokens so the 4K*K weight partials are reduced once per block ---
template <int K>
__global__ __launch_bounds__(256) void zigzag_inproj_bwd_kernel(CoordArgs p) {
    __shared__ float red[4 * ((4 * K * K + 3) & ~3)];
    const int L = p.H * p.W;
    const long total = (long)p.B * L;
    float dw[4 * K * K];
#pragma unroll
    for (int i = 0; i < 4 * K * K; ++i) dw[i] = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int b = (int)(idx / L), l = (int)(idx - (long)b * L);
        int h, w;
        unzig(l, p.H, p.W, h, w);
        float x[K], g[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            x[k] = p.off[(((long)b * 2 * K + k) * p.H + h) * p.W + w];
            g[k] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4 * K; ++j) {
            const float d = p.dxz[((long)j * p.B + b) * L + l];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                g[k] = fmaf(p.win[j * K + k], d, g[k]);
                dw[j * K + k] = fmaf(d, x[k], dw[j * K + k]);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (p.acc_doff) {   // the other consumer of the offsets (coords_outproj) has already written its share here
                p.doff[(((long)b * 2 * K + k) * p.H + h) * p.W + w] += g[k];
            } else {
                p.doff[(((long)b * 2 * K + k) * p.H + h) * p.W + w] = g[k];
                p.doff[(((long)b * 2 * K + K + k) * p.H + h) * p.W + w] = 0.f;
            }
        }
    }
    if (p.ws)
        block_sum_many_store<4 * K * K>(dw, red, p.ws + (long)blockIdx.x * ((4 * K * K + 3) & ~3));
    else
        block_sum_many_atomic<4 * K * K>(dw, red, p.dwin, 4 * K * K);
}

__device__ __forceinline__ float coord_weight(float altho, float &dwgt_daltho) {
    const float sp = altho <= 20.f ? log1pf(expf(altho)) : altho;  // F.softplus (threshold 20)
    const float sg = 1.f / (1.f + expf(-altho));
    dwgt_daltho = sp >= 0.01f ? (altho <= 20.f ? sg : 1.f) : 0.f;  // d max(softplus, 0.01) / d altho
    return fmaxf(sp, 0.01f);
}

// ---- B forward: one thread per pixel (b, h, w) -----------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void coords_outproj_fwd_kernel(CoordArgs p) {
    const int L = p.H * p.W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)p.B * L) return;
    const int b = (int)(idx / L), r = (int)(idx - (long)b * L);
    const int h = r / p.W, w = r - h * p.W;
    const int l = zig_of(h, w, p.H, p.W);
    float dummy;
    const float wgt = coord_weight(p.altho[0], dummy);
    float oz[2 * K], off[K], cum[K];
#pragma unroll
    for (int d = 0; d < 2 * K; ++d) oz[d] = p.oz[((long)d * p.B + b) * L + l];
#pragma unroll
    for (int k = 0; k < K; ++k) off[k] = p.off[(((long)b * 2 * K + k) * p.H + h) * p.W + w];
    constexpr int c = K / 2;
    cum[c] = 0.f;
#pragma unroll
    for (int i = 1; i <= c; ++i) {
        cum[c + i] = cum[c + i - 1] + off[c + i];
        cum[c - i] = cum[c - i + 1] + off[c - i];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 2 * K; ++d) s = fmaf(p.wout[k * 2 * K + d], oz[d], s);
        p.y[(((long)b * K + k) * p.H + h) * p.W + w] = fmaf(wgt, s, (float)h + p.scope * cum[k]);
    }
}

// ---- B backward ---------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void coords_outproj_bwd_kernel(CoordArgs p) {
    constexpr int NV = 2 * K * K + 1, NV4 = (NV + 3) & ~3;   // the out_proj weight gradient and d(altho)
    __shared__ float red[4 * NV4];
    const int L = p.H * p.W;
    const long total = (long)p.B * L;
    float dwgt_da;
    const float wgt = coord_weight(p.altho[0], dwgt_da);
    float dwo[2 * K * K], dwg = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * K * K; ++i) dwo[i] = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int b = (int)(idx / L), r = (int)(idx - (long)b * L);
        const int h = r / p.W, w = r - h * p.W;
        const int l = zig_of(h, w, p.H, p.W);
        float oz[2 * K], g[K];
#pragma unroll
        for (int d = 0; d < 2 * K; ++d) oz[d] = p.oz[((long)d * p.B + b) * L + l];
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] = p.dy[(((long)b * K + k) * p.H + h) * p.W + w];
        // through seq = Wout @ oz and y = wgt * seq + ...
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 2 * K; ++d) {
                s = fmaf(p.wout[k * 2 * K + d], oz[d], s);
                dwo[k * 2 * K + d] = fmaf(wgt * g[k], oz[d], dwo[k * 2 * K + d]);
            }
            dwg = fmaf(g[k], s, dwg);
        }
#pragma unroll
        for (int d = 0; d < 2 * K; ++d) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) s = fmaf(p.wout[k * 2 * K + d], g[k], s);
            p.doz[((long)d * p.B + b) * L + l] = wgt * s;
        }
        // through the cumulative offsets: tap j > c feeds cum[k] for k >= j, tap j < c for k <= j
        constexpr int c = K / 2;
        float go[K];
        go[c] = 0.f;
        {
            float run = 0.f;
#pragma unroll
            for (int j = K - 1; j > c; --j) {
                run += g[j];
                go[j] = p.scope * run;
            }
            run = 0.f;
#pragma unroll
            for (int j = 0; j < c; ++j) {
                run += g[j];
                go[j] = p.scope * run;
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            p.doff[(((long)b * 2 * K + k) * p.H + h) * p.W + w] = go[k];
            p.doff[(((long)b * 2 * K + K + k) * p.H + h) * p.W + w] = 0.f;
        }
    }
    float vals[NV];
#pragma unroll
    for (int i = 0; i < 2 * K * K; ++i) vals[i] = dwo[i];
    vals[NV - 1] = dwg;
    if (p.ws) {   // (the last value is this block's share of d altho)
        block_sum_many_store<NV>(vals, red, p.ws + (long)blockIdx.x * NV4, dwgt_da);
        return;
    }
    block_sum_many_atomic<NV>(vals, red, p.dwout, 2 * K * K);
    if (threadIdx.x == 0)
        atomicAdd(p.daltho, (red[NV - 1] + red[NV4 + NV - 1] + red[2 * NV4 + NV - 1] + red[3 * NV4 + NV - 1]) * dwgt_da);
}

int check(const mmu_coords_params *p, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->taps == 1 || p->taps == 3, "%s: fused path supports 1 or 3 taps (got %d)", name, p->taps);
    MMU_CHECK(p->batch > 0 && p->height > 0 && p->width > 0, "%s: empty tensor", name);
    MMU_CHECK((long)p->batch * p->height * p->width * 4 * p->taps < (1L << 31), "%s: tensor too large", name);
    return 0;
}

CoordArgs to_args(const mmu_coords_params *p) {
    CoordArgs a = {};
    a.B = p->batch; a.H = p->height; a.W = p->width; a.scope = p->extend_scope;
    a.off = p->offset; a.win = p->in_proj_weight; a.wout = p->out_proj_weight; a.altho = p->altho;
    a.xz = p->xz; a.dxz = p->dxz; a.oz = p->out_z; a.y = p->y; a.dy = p->dy; a.doff = p->doffset;
    a.dwin = p->din_proj_weight; a.doz = p->dout_z; a.dwout = p->dout_proj_weight; a.daltho = p->daltho;
    a.ws = p->workspace;
    a.acc_doff = p->accumulate_doffset;
    return a;
}

inline unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }
// grid-stride kernels that end in one float atomic per block and weight: one position per thread up to 256 blocks (a
// block's iterations are serial memory round trips -- 8 positions per thread made the 2,048-position calls 16-19 us
// for ONE workgroup, the forward kernels next to them take 5), more positions per thread beyond, so that no address
// sees more than 256 atomics
inline unsigned reduce_blocks(long n) {
    long b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

}  // namespace

#define DISPATCH_K(kernel, grid, st, a)           \
    do {                                          \
        if (p->taps == 3)                         \
            kernel<3><<<grid, 256, 0, st>>>(a);   \
        else                                      \
            kernel<1><<<grid, 256, 0, st>>>(a);   \
    } while (0)

extern "C" int mmu_zigzag_inproj_fwd(const mmu_coords_params *p, void *stream) {
    if (int r = check(p, "zigzag_inproj_fwd")) return r;
    MMU_CHECK(p->offset && p->in_proj_weight && p->xz, "zigzag_inproj_fwd: offset, in_proj_weight, xz required");
    CoordArgs a = to_args(p);
    DISPATCH_K(zigzag_inproj_fwd_kernel, blocks_for((long)a.B * a.H * a.W), (hipStream_t)stream, a);
    MMU_HIP_LAUNCH_CHECK("zigzag_inproj_fwd");
    return 0;
}

extern "C" int mmu_zigzag_inproj_bwd(const mmu_coords_params *p, void *stream) {
    if (int r = check(p, "zigzag_inproj_bwd")) return r;
    MMU_CHECK(p->offset && p->in_proj_weight && p->dxz && p->doffset && p->din_proj_weight,
              "zigzag_inproj_bwd: offset, in_proj_weight, dxz, doffset, din_proj_weight required");
    CoordArgs a = to_args(p);
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = reduce_blocks((long)a.B * a.H * a.W);
    if (a.ws == nullptr && mmu_zero_async(a.dwin, (size_t)4 * p->taps * p->taps, st) != hipSuccess)
        return mmu_fail("zigzag_inproj_bwd: memset failed");
    DISPATCH_K(zigzag_inproj_bwd_kernel, nb, st, a);
    MMU_HIP_LAUNCH_CHECK("zigzag_inproj_bwd");
    if (a.ws) {   // ordered sum of the per-block partials: with the other weight-gradient sums of the pass, or now
        const int nv = 4 * p->taps * p->taps, nv4 = (nv + 3) & ~3;
        const long job[8] = {3, (long)a.ws, (long)a.dwin, 0, nv, nb, nv4, nv};
        if (!mmu_defer_job(job)) {
            coords_partials_sum_kernel<<<(nv + 63) / 64, 1024, 0, st>>>(a.ws, a.dwin, nullptr, nv, (int)nb, nv4, nv);
            MMU_HIP_LAUNCH_CHECK("zigzag_inproj_bwd(sum)");
        }
    }
    return 0;
}

extern "C" int mmu_coords_outproj_fwd(const mmu_coords_params *p, void *stream) {
    if (int r = check(p, "coords_outproj_fwd")) return r;
    MMU_CHECK(p->offset && p->out_proj_weight && p->altho && p->out_z && p->y,
              "coords_outproj_fwd: offset, out_proj_weight, altho, out_z, y required");
    CoordArgs a = to_args(p);
    DISPATCH_K(coords_outproj_fwd_kernel, blocks_for((long)a.B * a.H * a.W), (hipStream_t)stream, a);
    MMU_HIP_LAUNCH_CHECK("coords_outproj_fwd");
    return 0;
}

extern "C" int mmu_coords_outproj_bwd(const mmu_coords_params *p, void *stream) {
    if (int r = check(p, "coords_outproj_bwd")) return r;
    MMU_CHECK(p->out_proj_weight && p->altho && p->out_z && p->dy && p->doffset && p->dout_z &&
                  p->dout_proj_weight && p->daltho,
              "coords_outproj_bwd: out_proj_weight, altho, out_z, dy, doffset, dout_z, dout_proj_weight, daltho required");
    CoordArgs a = to_args(p);
    hipStream_t st = (hipStream_t)stream;
    const size_t nw = (size_t)2 * p->taps * p->taps;
    const bool together = a.daltho == a.dwout + nw;   // one allocation: one zero fill
    const unsigned nb = reduce_blocks((long)a.B * a.H * a.W);
    if (a.ws == nullptr && (mmu_zero_async(a.dwout, nw + (together ? 1 : 0), st) != hipSuccess ||
                            (!together && mmu_zero_async(a.daltho, 1, st) != hipSuccess)))
        return mmu_fail("coords_outproj_bwd: memset failed");
    DISPATCH_K(coords_outproj_bwd_kernel, nb, st, a);
    MMU_HIP_LAUNCH_CHECK("coords_outproj_bwd");
    if (a.ws) {
        const int nv = (int)nw + 1, nv4 = (nv + 3) & ~3;
        const long job[8] = {3, (long)a.ws, (long)a.dwout, (long)a.daltho, (long)nw, nb, nv4, nv};
        if (!mmu_defer_job(job)) {
            coords_partials_sum_kernel<<<(nv + 63) / 64, 1024, 0, st>>>(a.ws, a.dwout, a.daltho, (int)nw, (int)nb, nv4, nv);
            MMU_HIP_LAUNCH_CHECK("coords_outproj_bwd(sum)");
        }
    }
    return 0;
}

extern "C" size_t mmu_coords_bwd_workspace_floats(int batch, int height, int width, int taps) {
    if (batch <= 0 || height <= 0 || width <= 0 || taps <= 0) return 0;
    return (size_t)reduce_blocks((long)batch * height * width) * ((4 * taps * taps + 3) & ~3);
}
