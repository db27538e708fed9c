// conv7x7_small.hip -- nn.Conv2d(2, 1, kernel_size=7, padding=3, bias=False): CBAM's spatial-attention convolution.
//
// Where it sits: src/UM_Net/MMUNet.py:323 (`self.conv`, applied at :335 to cat(max, mean) of the 64 x 256 x 256 map): 98
// weights, 1-2 MB of data.  MIOpen runs it as an implicit GEMM behind layout transposes: 76 us forward, 180 us backward.
//   fwd : out[b, y, x] = sum_{c, ky, kx} w[c][ky][kx] in[b, c, y + ky - 3, x + kx - 3]          (zero padding)
//   bwd : din[b, c, y, x] = sum_{ky, kx} w[c][ky][kx] g[b, y - ky + 3, x - kx + 3];
//         dw[c][ky][kx]   = sum_{b, y, x} g[b, y, x] in[b, c, y + ky - 3, x + kx - 3]
//         (98 sums in registers per thread, one partial row per workgroup, ordered sum by a second kernel: deterministic)
// One thread per pixel; the 7 x 7 neighbourhoods are L1 / L2 hits.  float32, contiguous.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

constexpr int K7 = 7, R7 = 3, NW7 = 2 * K7 * K7;   // 98 weights

__global__ __launch_bounds__(256) void conv7_fwd_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                        float *__restrict__ out, int B, int H, int W) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long HW = (long)H * W;
    if (t >= B * HW) return;
    const int b = (int)(t / HW);
    const int r = (int)(t - b * HW);
    const int y = r / W, x = r - y * W;
    const float *ip = in + (long)b * 2 * HW;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ky = 0; ky < K7; ++ky) {
            const int yy = y + ky - R7;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < K7; ++kx) {
                const int xx = x + kx - R7;
                if (xx >= 0 && xx < W) acc = fmaf(w[(c * K7 + ky) * K7 + kx], ip[c * HW + (long)yy * W + xx], acc);
            }
        }
    out[t] = acc;
}

__global__ __launch_bounds__(256) void conv7_bwd_data_kernel(const float *__restrict__ g, const float *__restrict__ w,
                                                             float *__restrict__ din, int B, int H, int W) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long HW = (long)H * W;
    if (t >= B * HW) return;
    const int b = (int)(t / HW);
    const int r = (int)(t - b * HW);
    const int y = r / W, x = r - y * W;
    const float *gp = g + (long)b * HW;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int ky = 0; ky < K7; ++ky) {
        const int yy = y - ky + R7;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < K7; ++kx) {
            const int xx = x - kx + R7;
            if (xx >= 0 && xx < W) {
                const float gv = gp[(long)yy * W + xx];
                a0 = fmaf(w[ky * K7 + kx], gv, a0);
                a1 = fmaf(w[(K7 + ky) * K7 + kx], gv, a1);
            }
        }
    }
    din[(long)b * 2 * HW + r] = a0;
    din[(long)b * 2 * HW + HW + r] = a1;
}

// grid = blocks of 256 pixels (2 pixels per thread, grid-stride); part[block][100]
__global__ __launch_bounds__(256) void conv7_bwd_weight_kernel(const float *__restrict__ in, const float *__restrict__ g,
                                                               float *__restrict__ part, int B, int H, int W) {
    constexpr int NV4 = (NW7 + 3) & ~3;
    __shared__ float red[4 * NV4];
    const long HW = (long)H * W, total = (long)B * HW;
    float v[NW7];
#pragma unroll
    for (int i = 0; i < NW7; ++i) v[i] = 0.f;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int b = (int)(t / HW);
        const int r = (int)(t - b * HW);
        const int y = r / W, x = r - y * W;
        const float gv = g[t];
        const float *ip = in + (long)b * 2 * HW;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ky = 0; ky < K7; ++ky) {
                const int yy = y + ky - R7;
                const bool rowok = yy >= 0 && yy < H;
#pragma unroll
                for (int kx = 0; kx < K7; ++kx) {
                    const int xx = x + kx - R7;
                    const float iv = (rowok && xx >= 0 && xx < W) ? ip[c * HW + (long)yy * W + xx] : 0.f;
                    v[(c * K7 + ky) * K7 + kx] = fmaf(gv, iv, v[(c * K7 + ky) * K7 + kx]);
                }
            }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV4; i += 4) {
        const float r = wave_sum4_swap(v[i], i + 1 < NW7 ? v[i + 1] : 0.f, i + 2 < NW7 ? v[i + 2] : 0.f,
                                       i + 3 < NW7 ? v[i + 3] : 0.f);
        if (lane >= 12 && lane < 16) red[wv * NV4 + i + lane - 12] = r;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NW7; i += 256)
        part[(long)blockIdx.x * NV4 + i] = (red[i] + red[NV4 + i]) + (red[2 * NV4 + i] + red[3 * NV4 + i]);
}

__global__ __launch_bounds__(256) void conv7_wsum_kernel(const float *__restrict__ part, float *__restrict__ dw, int nblk) {
    constexpr int NV4 = (NW7 + 3) & ~3;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= NW7) return;
    float s = 0.f;
    for (int k = lane; k < nblk; k += 64) s += part[(long)k * NV4 + i];
    s = wave_sum(s);
    if (lane == 0) dw[i] = s;
}

inline int conv7_wblocks(long pixels) {
    long b = (pixels + 511) / 512;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int check(const mmu_conv7x7_params *p, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->batch > 0 && p->height > 0 && p->width > 0 && (long)p->batch * p->height * p->width < (1L << 31),
              "%s: empty or too large tensor", name);
    MMU_CHECK(p->weight, "%s: weight is required", name);
    return 0;
}

}  // namespace

extern "C" size_t mmu_conv7x7_2to1_workspace_floats(int batch, int height, int width) {
    if (batch <= 0 || height <= 0 || width <= 0) return 0;
    return (size_t)conv7_wblocks((long)batch * height * width) * ((NW7 + 3) & ~3);
}

extern "C" int mmu_conv7x7_2to1_fwd(const mmu_conv7x7_params *p, void *stream) {
    if (int r = check(p, "conv7x7_2to1_fwd")) return r;
    MMU_CHECK(p->input && p->out, "conv7x7_2to1_fwd: input and out are required");
    const long n = (long)p->batch * p->height * p->width;
    conv7_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(p->input, p->weight, p->out, p->batch,
                                                                                 p->height, p->width);
    MMU_HIP_LAUNCH_CHECK("conv7x7_2to1_fwd");
    return 0;
}

extern "C" int mmu_conv7x7_2to1_bwd(const mmu_conv7x7_params *p, void *stream) {
    if (int r = check(p, "conv7x7_2to1_bwd")) return r;
    MMU_CHECK(p->dout, "conv7x7_2to1_bwd: dout is required");
    hipStream_t st = (hipStream_t)stream;
    const long n = (long)p->batch * p->height * p->width;
    if (p->dinput) {
        conv7_bwd_data_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(p->dout, p->weight, p->dinput, p->batch, p->height,
                                                                         p->width);
        MMU_HIP_LAUNCH_CHECK("conv7x7_2to1_bwd(data)");
    }
    if (p->dweight) {
        MMU_CHECK(p->input && p->workspace, "conv7x7_2to1_bwd: input and workspace are required for dweight");
        const int nblk = conv7_wblocks(n);
        conv7_bwd_weight_kernel<<<nblk, 256, 0, st>>>(p->input, p->dout, p->workspace, p->batch, p->height, p->width);
        MMU_HIP_LAUNCH_CHECK("conv7x7_2to1_bwd(weight)");
        const long job[8] = {7, (long)p->workspace, (long)p->dweight, 0, NW7, nblk, (NW7 + 3) & ~3, 0};
        if (mmu_defer_job(job)) return 0;   // (deferred_reduce.hip: with the other weight-gradient sums of the pass)
        conv7_wsum_kernel<<<(NW7 + 3) / 4, 256, 0, st>>>(p->workspace, p->dweight, nblk);
        MMU_HIP_LAUNCH_CHECK("conv7x7_2to1_bwd(weight sum)");
    }
    return 0;
}
