// cbam_stats.hip -- the two pooled statistics of CBAM, each as one pass with its arg-max kept for a one-pass backward.
//
// Where it sits: src/UM_Net/MMUNet.py:327-333 -- channel attention pools the 64 x 256 x 256 map over its pixels
// (`avg_pool`, `max_pool` -> [B, C]), spatial attention over its channels (`torch.max(dim=1)`, `torch.mean(dim=1)` ->
// [B, 2, H, W], max first).  As ATen ops: a mean pass, a max pass (and a cat) forward; backward a zero fill + scatter
// for the max, an expand + divide for the mean and the add of the two gradients.  Here
//   pixels  : rows (b, c) of hw floats -> mean[row], max[row], argmax[row];   d x = g_mean / hw + [p == argmax] g_max
//   channels: per pixel over C channels -> out[b][0] = max, out[b][1] = mean, argmax[b][p];
//             d x[c] = g[b][1] / C + [c == argmax] g[b][0]
// Ties: the FIRST maximal element (what torch.max returns).  float32, contiguous NCHW, hw % 4 == 0.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

// ---- over the pixels of a row ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void cs_rows_fwd_kernel(const float *__restrict__ x, float *__restrict__ mean,
                                                           float *__restrict__ mx, int *__restrict__ amax, long hw) {
    __shared__ float s_sum[16], s_max[16];
    __shared__ int s_arg[16];
    const long row = blockIdx.x;
    const float4 *xp = reinterpret_cast<const float4 *>(x + row * hw);
    float sum = 0.f, best = -INFINITY;
    int arg = 0x7fffffff;
    for (long i = threadIdx.x; i < hw / 4; i += blockDim.x) {
        const float4 v = xp[i];
        const float e[4] = {v.x, v.y, v.z, v.w};
        sum += (v.x + v.y) + (v.z + v.w);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (e[j] > best) { best = e[j]; arg = (int)(4 * i + j); }   // increasing positions per thread: first maximum
    }
    // wave: maximum, ties to the smaller position
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oa = __shfl_xor(arg, off);
        sum += __shfl_xor(sum, off);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[w] = sum; s_max[w] = best; s_arg[w] = arg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f, b = s_max[0];
        int a = s_arg[0];
        for (int i = 0; i < nw; ++i) {
            t += s_sum[i];
            if (s_max[i] > b || (s_max[i] == b && s_arg[i] < a)) { b = s_max[i]; a = s_arg[i]; }
        }
        mean[row] = t / (float)hw;
        mx[row] = b;
        amax[row] = a;
    }
}

__global__ __launch_bounds__(256) void cs_rows_bwd_kernel(const float *__restrict__ gmean, const float *__restrict__ gmax,
                                                          const int *__restrict__ amax, float *dx, const float *addend,
                                                          long hw, long total4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const long row = i / (hw / 4);
    const int p0 = (int)((i - row * (hw / 4)) * 4);
    const float a = gmean[row] / (float)hw, m = gmax[row];
    const int am = amax[row];
    float4 v = make_float4(a + (am == p0 ? m : 0.f), a + (am == p0 + 1 ? m : 0.f), a + (am == p0 + 2 ? m : 0.f),
                           a + (am == p0 + 3 ? m : 0.f));
    if (addend) {
        const float4 o = reinterpret_cast<const float4 *>(addend)[i];
        v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
    }
    reinterpret_cast<float4 *>(dx)[i] = v;
}

// ---- over the channels of a pixel --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cs_chan_fwd_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                          int *__restrict__ amax, int C, long HW4) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= HW4) return;
    const int b = blockIdx.y;
    const float4 *xp = reinterpret_cast<const float4 *>(x) + (long)b * C * HW4 + q;
    float4 best = xp[0], sum = best;
    int4 arg = make_int4(0, 0, 0, 0);
    for (int c = 1; c < C; ++c) {
        const float4 v = xp[c * HW4];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        if (v.x > best.x) { best.x = v.x; arg.x = c; }
        if (v.y > best.y) { best.y = v.y; arg.y = c; }
        if (v.z > best.z) { best.z = v.z; arg.z = c; }
        if (v.w > best.w) { best.w = v.w; arg.w = c; }
    }
    const float inv = 1.f / (float)C;
    float4 *op = reinterpret_cast<float4 *>(out) + (long)b * 2 * HW4 + q;
    op[0] = best;
    op[HW4] = make_float4(sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv);
    reinterpret_cast<int4 *>(amax)[b * HW4 + q] = arg;
}

__global__ __launch_bounds__(256) void cs_chan_bwd_kernel(const float *__restrict__ g, const int *__restrict__ amax,
                                                          float *dx, const float *addend, int C, long HW4) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= HW4) return;
    const int b = blockIdx.y;
    const float4 *gp = reinterpret_cast<const float4 *>(g) + (long)b * 2 * HW4 + q;
    const float4 *ap = addend ? reinterpret_cast<const float4 *>(addend) + (long)b * C * HW4 + q : nullptr;
    const float4 gm = gp[0], ga = gp[HW4];
    const int4 am = reinterpret_cast<const int4 *>(amax)[b * HW4 + q];
    const float inv = 1.f / (float)C;
    const float4 a = make_float4(ga.x * inv, ga.y * inv, ga.z * inv, ga.w * inv);
    float4 *dp = reinterpret_cast<float4 *>(dx) + (long)b * C * HW4 + q;
    for (int c = 0; c < C; ++c) {
        float4 v = make_float4(a.x + (am.x == c ? gm.x : 0.f), a.y + (am.y == c ? gm.y : 0.f),
                               a.z + (am.z == c ? gm.z : 0.f), a.w + (am.w == c ? gm.w : 0.f));
        if (ap) {
            const float4 o = ap[c * HW4];
            v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
        }
        dp[c * HW4] = v;
    }
}

// ---- channel attention's shared MLP on the pooled vectors: gate = sigmoid(mlp(avg) + mlp(max)) ------------------------
// mlp = Conv2d(C, R, 1, bias=False) -> ReLU -> Conv2d(R, C, 1, bias=False) on [B, C, 1, 1] (MMUNet.py:319-329): 64 -> 4 ->
// 64 on 8 vectors.  As modules that is 8 launches forward and 15 backward (MIOpen's naive 1x1 kernels, two adds for the
// weights used twice), each at the launch floor; here one workgroup does either direction.
__device__ __forceinline__ void gate_hidden(const mmu_cbam_gate_params &p, float *h /* [2][B][R] */) {
    const int B = p.batch, C = p.channels, R = p.hidden;
    for (int e = threadIdx.x; e < 2 * B * R; e += blockDim.x) {
        const int sidx = e / (B * R), b = (e / R) % B, r = e % R;
        const float *v = (sidx ? p.max : p.avg) + (long)b * C, *w = p.w1 + (long)r * C;
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(w[c], v[c], a);
        h[e] = fmaxf(a, 0.f);
    }
}

__global__ __launch_bounds__(256) void cbam_gate_fwd_kernel(mmu_cbam_gate_params p) {
    extern __shared__ float gsm[];
    const int B = p.batch, C = p.channels, R = p.hidden;
    float *h = gsm;
    gate_hidden(p, h);
    __syncthreads();
    for (int e = threadIdx.x; e < B * C; e += blockDim.x) {
        const int b = e / C, c = e % C;
        const float *w = p.w2 + (long)c * R, *ha = h + b * R, *hm = h + (B + b) * R;
        float oa = 0.f, om = 0.f;
        for (int r = 0; r < R; ++r) {
            oa = fmaf(w[r], ha[r], oa);
            om = fmaf(w[r], hm[r], om);
        }
        p.gate[e] = 1.f / (1.f + __expf(-(oa + om)));
    }
}

__global__ __launch_bounds__(256) void cbam_gate_bwd_kernel(mmu_cbam_gate_params p) {
    extern __shared__ float gsm[];
    const int B = p.batch, C = p.channels, R = p.hidden;
    float *h = gsm, *dh = h + 2 * B * R, *dz = dh + 2 * B * R;     // [2][B][R] | [2][B][R] | [B][C]
    gate_hidden(p, h);
    for (int e = threadIdx.x; e < B * C; e += blockDim.x) {
        const float g = p.gate[e];
        dz[e] = p.dgate[e] * g * (1.f - g);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * B * R; e += blockDim.x) {
        const int b = (e / R) % B, r = e % R;
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(p.w2[(long)c * R + r], dz[b * C + c], a);
        dh[e] = h[e] > 0.f ? a : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < B * C; e += blockDim.x) {        // d avg, d max
        const int b = e / C, c = e % C;
        float da = 0.f, dm = 0.f;
        for (int r = 0; r < R; ++r) {
            const float w = p.w1[(long)r * C + c];
            da = fmaf(w, dh[b * R + r], da);
            dm = fmaf(w, dh[(B + b) * R + r], dm);
        }
        if (p.davg) p.davg[e] = da;
        if (p.dmax) p.dmax[e] = dm;
    }
    for (int e = threadIdx.x; e < C * R; e += blockDim.x) {        // d w2 [C][R], d w1 [R][C]
        if (p.dw2) {
            const int c = e / R, r = e % R;
            float a = 0.f;
            for (int b = 0; b < B; ++b) a = fmaf(dz[b * C + c], h[b * R + r] + h[(B + b) * R + r], a);
            p.dw2[e] = a;
        }
        if (p.dw1) {
            const int r = e / C, c = e % C;
            float a = 0.f;
            for (int b = 0; b < B; ++b)
                a += dh[b * R + r] * p.avg[(long)b * C + c] + dh[(B + b) * R + r] * p.max[(long)b * C + c];
            p.dw1[e] = a;
        }
    }
}

int gate_check(const mmu_cbam_gate_params *p, const char *name, size_t &lds) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->batch > 0 && p->channels > 0 && p->hidden > 0, "%s: empty problem", name);
    MMU_CHECK(p->avg && p->max && p->w1 && p->w2 && p->gate, "%s: avg, max, w1, w2, gate are required", name);
    lds = ((size_t)4 * p->batch * p->hidden + (size_t)p->batch * p->channels) * sizeof(float);
    MMU_CHECK(lds <= 48 * 1024, "%s: batch * (channels + 4 * hidden) floats must fit 48 KB of LDS", name);
    return 0;
}

int check(const mmu_cbam_stats_params *p, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->batch > 0 && p->batch < 65536 && p->channels > 0 && p->hw > 0 && p->hw % 4 == 0 && p->hw < (1L << 31),
              "%s: batch in 1..65535, channels > 0, hw a positive multiple of 4 below 2^31 required", name);
    MMU_CHECK(p->mode == MMU_STATS_PIXELS || p->mode == MMU_STATS_CHANNELS, "%s: unknown mode %d", name, p->mode);
    MMU_CHECK((long)p->batch * p->channels < (1L << 31), "%s: too many rows", name);
    return 0;
}

}  // namespace

extern "C" int mmu_cbam_stats_fwd(const mmu_cbam_stats_params *p, void *stream) {
    if (int r = check(p, "cbam_stats_fwd")) return r;
    MMU_CHECK(p->input && p->argmax && ((uintptr_t)p->input & 15) == 0 && ((uintptr_t)p->argmax & 15) == 0,
              "cbam_stats_fwd: input and argmax (16-byte aligned) are required");
    hipStream_t st = (hipStream_t)stream;
    if (p->mode == MMU_STATS_PIXELS) {
        MMU_CHECK(p->mean && p->max, "cbam_stats_fwd: mean and max are required");
        cs_rows_fwd_kernel<<<(unsigned)(p->batch * p->channels), p->hw >= 8192 ? 1024 : 256, 0, st>>>(
            p->input, p->mean, p->max, p->argmax, p->hw);
    } else {
        MMU_CHECK(p->out && ((uintptr_t)p->out & 15) == 0, "cbam_stats_fwd: out (16-byte aligned) is required");
        const long HW4 = p->hw / 4;
        cs_chan_fwd_kernel<<<dim3((unsigned)((HW4 + 255) / 256), p->batch), 256, 0, st>>>(p->input, p->out, p->argmax,
                                                                                         p->channels, HW4);
    }
    MMU_HIP_LAUNCH_CHECK("cbam_stats_fwd");
    return 0;
}

extern "C" int mmu_cbam_stats_bwd(const mmu_cbam_stats_params *p, void *stream) {
    if (int r = check(p, "cbam_stats_bwd")) return r;
    MMU_CHECK(p->argmax && p->dinput && ((uintptr_t)p->dinput & 15) == 0 && ((uintptr_t)p->argmax & 15) == 0,
              "cbam_stats_bwd: argmax and dinput (16-byte aligned) are required");
    MMU_CHECK(((uintptr_t)p->dinput_addend & 15) == 0, "cbam_stats_bwd: dinput_addend must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long HW4 = p->hw / 4;
    if (p->mode == MMU_STATS_PIXELS) {
        MMU_CHECK(p->dmean && p->dmax, "cbam_stats_bwd: dmean and dmax are required");
        const long total4 = (long)p->batch * p->channels * HW4;
        cs_rows_bwd_kernel<<<(unsigned)((total4 + 255) / 256), 256, 0, st>>>(p->dmean, p->dmax, p->argmax, p->dinput,
                                                                           p->dinput_addend, p->hw, total4);
    } else {
        MMU_CHECK(p->dout && ((uintptr_t)p->dout & 15) == 0, "cbam_stats_bwd: dout (16-byte aligned) is required");
        cs_chan_bwd_kernel<<<dim3((unsigned)((HW4 + 255) / 256), p->batch), 256, 0, st>>>(p->dout, p->argmax, p->dinput,
                                                                                         p->dinput_addend, p->channels, HW4);
    }
    MMU_HIP_LAUNCH_CHECK("cbam_stats_bwd");
    return 0;
}

extern "C" int mmu_cbam_gate_fwd(const mmu_cbam_gate_params *p, void *stream) {
    size_t lds = 0;
    if (int r = gate_check(p, "cbam_gate_fwd", lds)) return r;
    cbam_gate_fwd_kernel<<<1, 256, lds, (hipStream_t)stream>>>(*p);
    MMU_HIP_LAUNCH_CHECK("cbam_gate_fwd");
    return 0;
}

extern "C" int mmu_cbam_gate_bwd(const mmu_cbam_gate_params *p, void *stream) {
    size_t lds = 0;
    if (int r = gate_check(p, "cbam_gate_bwd", lds)) return r;
    MMU_CHECK(p->dgate, "cbam_gate_bwd: dgate is required");
    cbam_gate_bwd_kernel<<<1, 256, lds, (hipStream_t)stream>>>(*p);
    MMU_HIP_LAUNCH_CHECK("cbam_gate_bwd");
    return 0;
}
