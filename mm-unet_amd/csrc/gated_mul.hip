// gated_mul.hip -- out = x * gate with a gate that is constant over the pixels of a channel ("channel gate", [B, C, 1, 1])
// or over the channels of a pixel ("spatial gate", [B, 1, H, W]), and its backward in ONE pass.
//
// Where it sits: CBAM (src/UM_Net/MMUNet.py:330,336: `c_out * x`, `s_out * y1` on the 64 x 256 x 256 stem map) and RCG's
// `x0 * gate * x2` (MMUNet.py:415).  As ATen ops the backward of each product is a broadcast multiply for dx, a second
// full-size multiply g * x and a reduction of it for d gate (the reduction alone: 83 us for 134 MB).  Here
//   bwd : dx = g * gate;  dgate = sum of g * x over the gate's broadcast axis          (read g, x once; write dx)
// channel gate: one workgroup per (b, c) row (block sum, no atomics); spatial gate: one thread per 4 pixels walking the
// channels.  float32, contiguous NCHW, H * W % 4 == 0.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

__device__ __forceinline__ float gm_block_sum(float v, float *red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// rows = B * C; gate[row]
__global__ __launch_bounds__(256) void gm_channel_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gate,
                                                             float *__restrict__ out, long HW4, long total4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const float gv = gate[i / HW4];
    const float4 v = reinterpret_cast<const float4 *>(x)[i];
    reinterpret_cast<float4 *>(out)[i] = make_float4(v.x * gv, v.y * gv, v.z * gv, v.w * gv);
}

__global__ __launch_bounds__(1024) void gm_channel_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gate,
                                                              const float *__restrict__ g, float *__restrict__ dx,
                                                              float *__restrict__ dgate, long HW4,
                                                              const float *__restrict__ sg, const int *__restrict__ sam,
                                                              int C) {
    __shared__ float red[16];
    const long row = blockIdx.x;
    const float gv = gate[row];
    // (sg, sam: the gradient through the channel statistics of this product, added to g on the fly)
    const int bi = (int)(row / C), c = (int)(row - (long)bi * C);
    const float4 *sgm = sg ? reinterpret_cast<const float4 *>(sg) + (long)bi * 2 * HW4 : nullptr;
    const int4 *samp = sg ? reinterpret_cast<const int4 *>(sam) + (long)bi * HW4 : nullptr;
    const float invC = 1.f / (float)C;
    const float4 *xp = reinterpret_cast<const float4 *>(x) + row * HW4, *gp = reinterpret_cast<const float4 *>(g) + row * HW4;
    float4 *dp = dx ? reinterpret_cast<float4 *>(dx) + row * HW4 : nullptr;
    float s = 0.f;
    for (long i = threadIdx.x; i < HW4; i += blockDim.x) {
        const float4 a = xp[i];
        float4 b = gp[i];
        if (sgm) {
            const float4 gm = sgm[i], ga = sgm[HW4 + i];
            const int4 am = samp[i];
            b.x += ga.x * invC + (am.x == c ? gm.x : 0.f);
            b.y += ga.y * invC + (am.y == c ? gm.y : 0.f);
            b.z += ga.z * invC + (am.z == c ? gm.z : 0.f);
            b.w += ga.w * invC + (am.w == c ? gm.w : 0.f);
        }
        s += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        if (dp) dp[i] = make_float4(b.x * gv, b.y * gv, b.z * gv, b.w * gv);
    }
    s = gm_block_sum(s, red);
    if (threadIdx.x == 0 && dgate) dgate[row] = s;
}

// gate[b][p]; thread = 4 pixels of one batch item
__global__ __launch_bounds__(256) void gm_spatial_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gate,
                                                             float *__restrict__ out, int C, long HW4) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= HW4) return;
    const int b = blockIdx.y;
    const float4 gv = reinterpret_cast<const float4 *>(gate)[b * HW4 + q];
    const float4 *xp = reinterpret_cast<const float4 *>(x) + (long)b * C * HW4 + q;
    float4 *op = reinterpret_cast<float4 *>(out) + (long)b * C * HW4 + q;
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        const float4 v = xp[c * HW4];
        op[c * HW4] = make_float4(v.x * gv.x, v.y * gv.y, v.z * gv.z, v.w * gv.w);
    }
}

__global__ __launch_bounds__(256) void gm_spatial_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gate,
                                                             const float *__restrict__ g, float *__restrict__ dx,
                                                             float *__restrict__ dgate, int C, long HW4) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= HW4) return;
    const int b = blockIdx.y;
    const float4 gv = reinterpret_cast<const float4 *>(gate)[b * HW4 + q];
    const float4 *xp = reinterpret_cast<const float4 *>(x) + (long)b * C * HW4 + q;
    const float4 *gp = reinterpret_cast<const float4 *>(g) + (long)b * C * HW4 + q;
    float4 *dp = dx ? reinterpret_cast<float4 *>(dx) + (long)b * C * HW4 + q : nullptr;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        const float4 a = xp[c * HW4], d = gp[c * HW4];
        s.x = fmaf(a.x, d.x, s.x); s.y = fmaf(a.y, d.y, s.y); s.z = fmaf(a.z, d.z, s.z); s.w = fmaf(a.w, d.w, s.w);
        if (dp) dp[c * HW4] = make_float4(d.x * gv.x, d.y * gv.y, d.z * gv.z, d.w * gv.w);
    }
    if (dgate) reinterpret_cast<float4 *>(dgate)[b * HW4 + q] = s;
}

int check(const mmu_gated_mul_params *p, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->batch > 0 && p->batch < 65536 && p->channels > 0 && p->hw > 0 && p->hw % 4 == 0,
              "%s: batch in 1..65535, channels > 0 and hw a positive multiple of 4 required", name);
    MMU_CHECK(p->mode == MMU_GATE_CHANNEL || p->mode == MMU_GATE_SPATIAL, "%s: unknown mode %d", name, p->mode);
    MMU_CHECK(p->input && p->gate, "%s: input and gate are required", name);
    MMU_CHECK((long)p->batch * p->channels < (1L << 31), "%s: too many rows", name);
    return 0;
}

}  // namespace

extern "C" int mmu_gated_mul_fwd(const mmu_gated_mul_params *p, void *stream) {
    if (int r = check(p, "gated_mul_fwd")) return r;
    MMU_CHECK(p->out && ((uintptr_t)p->out & 15) == 0 && ((uintptr_t)p->input & 15) == 0 && ((uintptr_t)p->gate & 15) == 0,
              "gated_mul_fwd: out is required; input, gate, out 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long HW4 = p->hw / 4;
    if (p->mode == MMU_GATE_CHANNEL) {
        const long total4 = (long)p->batch * p->channels * HW4;
        gm_channel_fwd_kernel<<<(unsigned)((total4 + 255) / 256), 256, 0, st>>>(p->input, p->gate, p->out, HW4, total4);
    } else {
        gm_spatial_fwd_kernel<<<dim3((unsigned)((HW4 + 255) / 256), p->batch), 256, 0, st>>>(p->input, p->gate, p->out,
                                                                                            p->channels, HW4);
    }
    MMU_HIP_LAUNCH_CHECK("gated_mul_fwd");
    return 0;
}

extern "C" int mmu_gated_mul_bwd(const mmu_gated_mul_params *p, void *stream) {
    if (int r = check(p, "gated_mul_bwd")) return r;
    MMU_CHECK(p->dout && (p->dinput || p->dgate), "gated_mul_bwd: dout and at least one of dinput / dgate are required");
    const void *ptrs[] = {p->input, p->gate, p->dout, p->dinput, p->dgate};
    for (const void *q : ptrs) MMU_CHECK(((uintptr_t)q & 15) == 0, "gated_mul_bwd: tensors must be 16-byte aligned");
    MMU_CHECK((p->stats_dout == nullptr) == (p->stats_argmax == nullptr) && (!p->stats_dout || p->mode == MMU_GATE_CHANNEL) &&
                  ((uintptr_t)p->stats_dout & 15) == 0 && ((uintptr_t)p->stats_argmax & 15) == 0,
              "gated_mul_bwd: stats_dout / stats_argmax come together (16-byte aligned), for the channel gate only");
    hipStream_t st = (hipStream_t)stream;
    const long HW4 = p->hw / 4;
    if (p->mode == MMU_GATE_CHANNEL) {
        const int threads = HW4 >= 2048 ? 1024 : 256;
        gm_channel_bwd_kernel<<<(unsigned)(p->batch * p->channels), threads, 0, st>>>(p->input, p->gate, p->dout, p->dinput,
                                                                                     p->dgate, HW4, p->stats_dout,
                                                                                     p->stats_argmax, p->channels);
    } else {
        gm_spatial_bwd_kernel<<<dim3((unsigned)((HW4 + 255) / 256), p->batch), 256, 0, st>>>(p->input, p->gate, p->dout,
                                                                                            p->dinput, p->dgate, p->channels, HW4);
    }
    MMU_HIP_LAUNCH_CHECK("gated_mul_bwd");
    return 0;
}
