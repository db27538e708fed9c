// sum_parts.hip -- out = sum of up to four float32 tensors, written as float32 or bfloat16: one pass instead of a chain
// of in-place adds and a cast.
//
// Where it sits: a d_state-64 scan (BASELINE config 5) runs as four dstate-16 launches (selective_scan_hip._fwd_groups /
// _bwd_groups); every per-token output (y, gated y, du, ddelta, dz) is the SUM of the four groups' float32 partial
// outputs, returned in the I/O type.  As ATen ops that is three `add_` passes (read 2, write 1 each) and a cast: 42 bytes
// per element for a bf16 result; here 18.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

struct SumArgs {
    const float *p[4];
    void *out;
    long n;
    int np;
};

template <typename out_t>
__global__ __launch_bounds__(256) void sum_parts_kernel(SumArgs a) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= a.n) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (i + 4 <= a.n) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < a.np) {
                const float4 v = *reinterpret_cast<const float4 *>(a.p[k] + i);
                s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            }
        store_k<out_t, 4, true>(static_cast<out_t *>(a.out) + i, 4, true, s);
    } else {
        for (long j = i; j < a.n; ++j) {
            float t = 0.f;
            for (int k = 0; k < a.np; ++k) t += a.p[k][j];
            static_cast<out_t *>(a.out)[j] = from_f32<out_t>(t);
        }
    }
}

// Bias gradient of a convolution: out[c] = sum over (b, h, w) of g[b, c, h, w].  Stage 1: one workgroup per (b, c) row
// (1,024 threads, 16-byte loads: the rows are contiguous) writes the row sum; stage 2 adds the B row sums of a channel in
// order.  (ATen's generic reduction took 145 us for [8, 64, 256, 256] = 0.9 TB/s; this is a streaming pass.)
__global__ __launch_bounds__(1024) void channel_sum_rows_kernel(const float *__restrict__ g, float *__restrict__ rows, long hw) {
    __shared__ float red[16];
    const float *r = g + (long)blockIdx.x * hw;
    float s = 0.f;
    const long n4 = hw / 4;
    for (long i = threadIdx.x; i < n4; i += 1024) {
        const float4 v = *reinterpret_cast<const float4 *>(r + 4 * i);
        s += (v.x + v.y) + (v.z + v.w);
    }
    for (long i = 4 * n4 + threadIdx.x; i < hw; i += 1024) s += r[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += red[k];
        rows[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void channel_sum_batch_kernel(const float *__restrict__ rows, float *__restrict__ out,
                                                                int batch, int channels) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= channels) return;
    float t = 0.f;
    for (int b = 0; b < batch; ++b) t += rows[(long)b * channels + c];
    out[c] = t;
}

// Input gradient of a stride-2 1 x 1 convolution from the gradient of its gathered input: dst[p][2y][2x] = src[p][y][x], 0
// elsewhere (+ addend).  In place on a gradient another consumer of the input left (addend == dst) only the even pixels are
// touched: a quarter of the tensor instead of a zero fill, a strided copy and an add over all of it.
__global__ __launch_bounds__(256) void scatter_s2_inplace_kernel(const float *__restrict__ src, float *dst, int h, int w, int H,
                                                                 int W, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % w);
    const long r = i / w;
    const int y = (int)(r % h);
    const long p = r / h;
    dst[(p * H + 2 * y) * W + 2 * x] += src[i];
}

__global__ __launch_bounds__(256) void scatter_s2_full_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                              const float *__restrict__ addend, int h, int w, int H, int W,
                                                              long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // one output element
    if (i >= n) return;
    const int X = (int)(i % W);
    const long r = i / W;
    const int Y = (int)(r % H);
    const long p = r / H;
    float v = ((X | Y) & 1) ? 0.f : src[(p * h + (Y >> 1)) * w + (X >> 1)];
    if (addend) v += addend[i];
    dst[i] = v;
}

}  // namespace

extern "C" int mmu_scatter_stride2(const float *src, float *dst, const float *addend, int64_t planes, int height, int width,
                                   void *stream) {
    MMU_CHECK(src && dst && planes > 0 && height > 0 && width > 0, "scatter_stride2: src, dst and positive sizes required");
    const int h = (height + 1) / 2, w = (width + 1) / 2;
    hipStream_t st = (hipStream_t)stream;
    if (addend == dst) {
        const long n = planes * h * w;
        scatter_s2_inplace_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(src, dst, h, w, height, width, n);
    } else {
        const long n = planes * height * width;
        MMU_CHECK(n < (1L << 39), "scatter_stride2: tensor too large");
        scatter_s2_full_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(src, dst, addend, h, w, height, width, n);
    }
    MMU_HIP_LAUNCH_CHECK("scatter_stride2");
    return 0;
}

extern "C" int mmu_channel_sum(const float *g, int batch, int channels, int64_t hw, float *workspace, float *out, void *stream) {
    MMU_CHECK(g && workspace && out && batch > 0 && channels > 0 && hw > 0, "channel_sum: g, workspace, out and positive sizes required");
    MMU_CHECK(((uintptr_t)g & 15) == 0 && hw % 4 == 0, "channel_sum: g must be 16-byte aligned and hw a multiple of 4");
    MMU_CHECK((long)batch * channels < (1L << 31), "channel_sum: too many rows");
    hipStream_t st = (hipStream_t)stream;
    channel_sum_rows_kernel<<<(unsigned)(batch * channels), 1024, 0, st>>>(g, workspace, hw);
    MMU_HIP_LAUNCH_CHECK("channel_sum(rows)");
    if (batch <= 16) {   // (the deferred row sum adds up to 16 parts in the same order; more would change the rounding)
        const long job[8] = {3, (long)workspace, (long)out, 0, channels, batch, channels, channels};
        if (mmu_defer_job(job)) return 0;   // (deferred_reduce.hip: with the other parameter-gradient sums of the pass)
    }
    channel_sum_batch_kernel<<<(channels + 255) / 256, 256, 0, st>>>(workspace, out, batch, channels);
    MMU_HIP_LAUNCH_CHECK("channel_sum");
    return 0;
}

extern "C" int mmu_sum_parts(const mmu_sum_parts_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "sum_parts: null params");
    MMU_CHECK(p->n > 0 && p->nparts >= 1 && p->nparts <= 4, "sum_parts: n > 0 and 1..4 parts required");
    MMU_CHECK(p->out_dtype == MMU_DTYPE_F32 || p->out_dtype == MMU_DTYPE_BF16, "sum_parts: unsupported out_dtype %d",
              p->out_dtype);
    MMU_CHECK(p->out && ((uintptr_t)p->out & 15) == 0, "sum_parts: out (16-byte aligned) is required");
    SumArgs a;
    for (int k = 0; k < 4; ++k) {
        a.p[k] = k < p->nparts ? p->parts[k] : nullptr;
        MMU_CHECK(k >= p->nparts || (a.p[k] && ((uintptr_t)a.p[k] & 15) == 0), "sum_parts: part %d missing or not 16-byte aligned", k);
    }
    a.out = p->out; a.n = p->n; a.np = p->nparts;
    const unsigned blocks = (unsigned)((p->n + 1023) / 1024);
    if (p->out_dtype == MMU_DTYPE_BF16)
        sum_parts_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(a);
    else
        sum_parts_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(a);
    MMU_HIP_LAUNCH_CHECK("sum_parts");
    return 0;
}
