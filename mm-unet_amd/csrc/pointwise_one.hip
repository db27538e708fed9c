// pointwise_one.hip -- nn.Conv2d(C, 1, kernel_size=1): one output channel, a dot product over the channels per pixel.
//
// Where it sits: RCG's gate `mlp = Conv2d(64, 1, 1) -> Sigmoid` (src/UM_Net/MMUNet.py:386-387, applied at :414) and the
// side outputs' `conv2 = Conv2d(16, 1, 1)` (MMUNet.py:346,350): seven calls per step.  MIOpen runs them as implicit GEMMs
// behind NCHW <-> NHWC transposes: 31 us forward and 79 us backward for [8, 64, 128, 128] (33 MB).  Here:
//   fwd : out[b, p]   = bias + sum_c w[c] x[b, c, p]                      thread = 4 pixels, 16-byte loads
//   bwd : dx[b, c, p] = g[b, p] w[c];  dw[c] = sum_{b, p} g x[b, c, p];  dbias = sum g
//         one pass over x: thread = 4 pixels, C + 1 sums in registers, one partial per workgroup (wave_sum4_swap + LDS),
//         a second kernel adds the workgroups' partials in a fixed order (deterministic, no atomics, nothing to zero).
// HBM-bound streaming kernels: (C + 1) * 4 bytes per pixel forward, (2 C + 1) * 4 backward.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

template <int C, bool SCALED>
__global__ __launch_bounds__(256) void pw1_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ bias, float *__restrict__ out, int B,
                                                      long HW, const float *__restrict__ scale) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;      // group of 4 pixels of batch item blockIdx.y
    if (q * 4 >= HW) return;
    const int b = blockIdx.y;
    const float *sc = SCALED ? scale + (long)b * C : nullptr;  // (b, c) factors on the input: Dropout2d's mask / (1 - p)
    const float *xp = x + (long)b * C * HW + q * 4;
    const float bv = bias ? bias[0] : 0.f;
    float a0 = bv, a1 = bv, a2 = bv, a3 = bv;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const float4 v = *reinterpret_cast<const float4 *>(xp + (long)c * HW);
        float wv = w[c];
        if constexpr (SCALED) wv *= sc[c];
        a0 = fmaf(wv, v.x, a0); a1 = fmaf(wv, v.y, a1); a2 = fmaf(wv, v.z, a2); a3 = fmaf(wv, v.w, a3);
    }
    *reinterpret_cast<float4 *>(out + (long)b * HW + q * 4) = make_float4(a0, a1, a2, a3);
}

template <int C, bool SCALED>
__global__ __launch_bounds__(256) void pw1_bwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ g, float *__restrict__ dx,
                                                      float *__restrict__ part, int B, long HW,
                                                      const float *__restrict__ scale) {
    constexpr int NV = C + 1, NV4 = (NV + 3) & ~3;
    __shared__ float red[4 * NV4];
    const float *sc = SCALED ? scale + (long)blockIdx.y * C : nullptr;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const bool live = q * 4 < HW;
    const long qq = live ? q : 0;
    const float4 g4r = *reinterpret_cast<const float4 *>(g + (long)b * HW + qq * 4);
    const float m = live ? 1.f : 0.f;
    const float4 g4 = make_float4(g4r.x * m, g4r.y * m, g4r.z * m, g4r.w * m);
    const float *xp = x + (long)b * C * HW + qq * 4;
    float *dp = dx ? dx + (long)b * C * HW + qq * 4 : nullptr;
    float v[NV];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float4 xv = *reinterpret_cast<const float4 *>(xp + (long)c * HW);
        v[c] = fmaf(g4.x, xv.x, fmaf(g4.y, xv.y, fmaf(g4.z, xv.z, g4.w * xv.w)));
        if constexpr (SCALED) v[c] *= sc[c];
        if (dp && live) {
            float wv = w[c];
            if constexpr (SCALED) wv *= sc[c];
            *reinterpret_cast<float4 *>(dp + (long)c * HW) = make_float4(g4.x * wv, g4.y * wv, g4.z * wv, g4.w * wv);
        }
    }
    v[C] = (g4.x + g4.y) + (g4.z + g4.w);
    const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV4; i += 4) {
        const float r = wave_sum4_swap(v[i], i + 1 < NV ? v[i + 1] : 0.f, i + 2 < NV ? v[i + 2] : 0.f,
                                       i + 3 < NV ? v[i + 3] : 0.f);
        if (lane >= 12 && lane < 16) red[wv_ * NV4 + i + lane - 12] = r;
    }
    __syncthreads();
    const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
    for (int i = threadIdx.x; i < NV; i += 256)
        part[blk * NV4 + i] = (red[i] + red[NV4 + i]) + (red[2 * NV4 + i] + red[3 * NV4 + i]);
}

// one wave per result: lanes stride over the workgroups' partials, fixed-order tree at the end
template <int C>
__global__ __launch_bounds__(256) void pw1_sum_kernel(const float *__restrict__ part, float *__restrict__ dw,
                                                      float *__restrict__ db, int nblk) {
    constexpr int NV = C + 1, NV4 = (NV + 3) & ~3;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= NV) return;
    float s = 0.f;
    for (int k = lane; k < nblk; k += 64) s += part[(long)k * NV4 + i];
    s = wave_sum(s);
    if (lane == 0) {
        if (i < C) {
            if (dw) dw[i] = s;
        } else if (db) {
            db[0] = s;
        }
    }
}

int check(const mmu_conv1x1_one_params *p, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->batch > 0 && p->batch < 65536 && p->hw > 0 && p->hw % 4 == 0, "%s: batch in 1..65535 and hw a positive multiple of 4 required",
              name);
    MMU_CHECK(p->channels == 16 || p->channels == 64, "%s: 16 or 64 input channels (got %d)", name, p->channels);
    MMU_CHECK(p->input && p->weight, "%s: input and weight are required", name);
    MMU_CHECK(((uintptr_t)p->input & 15) == 0, "%s: input must be 16-byte aligned", name);
    return 0;
}

inline unsigned pw1_blocks(long hw) { return (unsigned)((hw / 4 + 255) / 256); }

}  // namespace

extern "C" size_t mmu_conv1x1_one_workspace_floats(int batch, int channels, long hw) {
    if (batch <= 0 || channels <= 0 || hw <= 0) return 0;
    return (size_t)batch * pw1_blocks(hw) * ((channels + 1 + 3) & ~3);
}

extern "C" int mmu_conv1x1_one_fwd(const mmu_conv1x1_one_params *p, void *stream) {
    if (int r = check(p, "conv1x1_one_fwd")) return r;
    MMU_CHECK(p->out && ((uintptr_t)p->out & 15) == 0, "conv1x1_one_fwd: out (16-byte aligned) is required");
    dim3 grid(pw1_blocks(p->hw), p->batch);
    hipStream_t st = (hipStream_t)stream;
    if (p->channels == 16) {
        if (p->scale) pw1_fwd_kernel<16, true><<<grid, 256, 0, st>>>(p->input, p->weight, p->bias, p->out, p->batch, p->hw, p->scale);
        else pw1_fwd_kernel<16, false><<<grid, 256, 0, st>>>(p->input, p->weight, p->bias, p->out, p->batch, p->hw, nullptr);
    } else {
        if (p->scale) pw1_fwd_kernel<64, true><<<grid, 256, 0, st>>>(p->input, p->weight, p->bias, p->out, p->batch, p->hw, p->scale);
        else pw1_fwd_kernel<64, false><<<grid, 256, 0, st>>>(p->input, p->weight, p->bias, p->out, p->batch, p->hw, nullptr);
    }
    MMU_HIP_LAUNCH_CHECK("conv1x1_one_fwd");
    return 0;
}

extern "C" int mmu_conv1x1_one_bwd(const mmu_conv1x1_one_params *p, void *stream) {
    if (int r = check(p, "conv1x1_one_bwd")) return r;
    MMU_CHECK(p->dout && p->workspace && ((uintptr_t)p->dout & 15) == 0 && (!p->dinput || ((uintptr_t)p->dinput & 15) == 0),
              "conv1x1_one_bwd: dout and workspace are required; dout / dinput 16-byte aligned");
    dim3 grid(pw1_blocks(p->hw), p->batch);
    const int nblk = (int)(grid.x * grid.y);
    hipStream_t st = (hipStream_t)stream;
    const int C = p->channels, NV4 = (C + 1 + 3) & ~3;
#define PW1_BWD(C_, S_) pw1_bwd_kernel<C_, S_><<<grid, 256, 0, st>>>(p->input, p->weight, p->dout, p->dinput, p->workspace, p->batch, p->hw, p->scale)
    if (C == 16) {
        if (p->scale) PW1_BWD(16, true); else PW1_BWD(16, false);
    } else {
        if (p->scale) PW1_BWD(64, true); else PW1_BWD(64, false);
    }
#undef PW1_BWD
    MMU_HIP_LAUNCH_CHECK("conv1x1_one_bwd");
    if (p->dweight || p->dbias) {
        // inside a deferred scope: with the other parameter-gradient sums of the pass (deferred_reduce.hip, kind 7)
        if (p->dweight && p->dbias) {
            const long job[8] = {7, (long)p->workspace, (long)p->dweight, (long)p->dbias, C + 1, nblk, NV4, C};
            if (mmu_defer_job(job)) return 0;
        }
        if (C == 16) pw1_sum_kernel<16><<<(17 + 3) / 4, 256, 0, st>>>(p->workspace, p->dweight, p->dbias, nblk);
        else pw1_sum_kernel<64><<<(65 + 3) / 4, 256, 0, st>>>(p->workspace, p->dweight, p->dbias, nblk);
    }
    MMU_HIP_LAUNCH_CHECK("conv1x1_one_bwd(sum)");
    return 0;
}
