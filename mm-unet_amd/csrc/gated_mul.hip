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

// gate[b][p].  x2 (optional): a second factor of the same shape (RCG's x0 * x2 * gate, MMUNet.py:415); addend (optional):
// added to the product (RCG's "+ f").  Forward: plain elementwise, one float4 per thread.  (Round 2's form -- a thread
// walking all channels of its four pixels -- ran 64 dependent iterations on 128 workgroups at 128 x 128: 45 us for what
// the elementwise kernels it replaced did in 14.)
__device__ __forceinline__ float4 gm_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

__global__ __launch_bounds__(256) void gm_spatial_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gate,
                                                             float *__restrict__ out, long CHW4, long HW4, long total4,
                                                             const float *__restrict__ x2, const float *__restrict__ addend) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const long b = i / CHW4, p = (i - b * CHW4) % HW4;
    float4 v = reinterpret_cast<const float4 *>(x)[i];
    if (x2) v = gm_mul(v, reinterpret_cast<const float4 *>(x2)[i]);
    v = gm_mul(v, reinterpret_cast<const float4 *>(gate)[b * HW4 + p]);
    if (addend) {
        const float4 a = reinterpret_cast<const float4 *>(addend)[i];
        v = make_float4(v.x + a.x, v.y + a.y, v.z + a.z, v.w + a.w);
    }
    reinterpret_cast<float4 *>(out)[i] = v;
}

// dx = g gate [x2];  dx2 = g gate x;  dgate = sum over the channels of g x [x2].  Thread = 4 pixels of one batch item and
// a CHUNK of GM_CCH channels (grid.z): the chunk sums go to part[chunk][b][p] and gm_chunk_sum_kernel adds them in order
// (one chunk: straight to dgate).
constexpr int GM_CCH = 8;

__global__ __launch_bounds__(256) void gm_spatial_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gate,
                                                             const float *__restrict__ g, float *__restrict__ dx,
                                                             float *__restrict__ dgate, int C, long HW4,
                                                             const float *__restrict__ x2, float *__restrict__ dx2,
                                                             float *__restrict__ part) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= HW4) return;
    const int b = blockIdx.y, c0 = blockIdx.z * GM_CCH, c1 = min(c0 + GM_CCH, C);
    const float4 gv = reinterpret_cast<const float4 *>(gate)[b * HW4 + q];
    const long base = (long)b * C * HW4 + q;
    const float4 *xp = reinterpret_cast<const float4 *>(x) + base;
    const float4 *gp = reinterpret_cast<const float4 *>(g) + base;
    const float4 *x2p = x2 ? reinterpret_cast<const float4 *>(x2) + base : nullptr;
    float4 *dp = dx ? reinterpret_cast<float4 *>(dx) + base : nullptr;
    float4 *d2p = dx2 ? reinterpret_cast<float4 *>(dx2) + base : nullptr;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        float4 a = xp[c * HW4];
        const float4 d = gp[c * HW4];
        const float4 dg = gm_mul(d, gv);
        if (x2p) {
            const float4 w = x2p[c * HW4];
            if (dp) dp[c * HW4] = gm_mul(dg, w);
            if (d2p) d2p[c * HW4] = gm_mul(dg, a);
            a = gm_mul(a, w);
        } else if (dp) {
            dp[c * HW4] = dg;
        }
        s.x = fmaf(a.x, d.x, s.x); s.y = fmaf(a.y, d.y, s.y); s.z = fmaf(a.z, d.z, s.z); s.w = fmaf(a.w, d.w, s.w);
    }
    if (gridDim.z == 1) {
        if (dgate) reinterpret_cast<float4 *>(dgate)[b * HW4 + q] = s;
    } else if (part) {
        reinterpret_cast<float4 *>(part)[((long)blockIdx.z * gridDim.y + b) * HW4 + q] = s;
    }
}

__global__ __launch_bounds__(256) void gm_chunk_sum_kernel(const float *__restrict__ part, float *__restrict__ dgate, int nchunk,
                                                           long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = reinterpret_cast<const float4 *>(part)[i];
    for (int k = 1; k < nchunk; ++k) {
        const float4 v = reinterpret_cast<const float4 *>(part)[(long)k * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4 *>(dgate)[i] = s;
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

extern "C" size_t mmu_gated_mul_bwd_workspace_floats(int batch, int channels, int64_t hw, int mode) {
    if (mode != MMU_GATE_SPATIAL || batch <= 0 || channels <= GM_CCH || hw <= 0) return 0;
    return (size_t)((channels + GM_CCH - 1) / GM_CCH) * batch * hw;
}

extern "C" int mmu_gated_mul_fwd(const mmu_gated_mul_params *p, void *stream) {
    if (int r = check(p, "gated_mul_fwd")) return r;
    MMU_CHECK(p->out && ((uintptr_t)p->out & 15) == 0 && ((uintptr_t)p->input & 15) == 0 && ((uintptr_t)p->gate & 15) == 0,
              "gated_mul_fwd: out is required; input, gate, out 16-byte aligned");
    MMU_CHECK((!p->input2 && !p->addend) || p->mode == MMU_GATE_SPATIAL, "gated_mul_fwd: input2 / addend need the spatial gate");
    MMU_CHECK(((uintptr_t)p->input2 & 15) == 0 && ((uintptr_t)p->addend & 15) == 0, "gated_mul_fwd: input2, addend 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long HW4 = p->hw / 4;
    if (p->mode == MMU_GATE_CHANNEL) {
        const long total4 = (long)p->batch * p->channels * HW4;
        gm_channel_fwd_kernel<<<(unsigned)((total4 + 255) / 256), 256, 0, st>>>(p->input, p->gate, p->out, HW4, total4);
    } else {
        const long CHW4 = (long)p->channels * HW4, total4 = (long)p->batch * CHW4;
        gm_spatial_fwd_kernel<<<(unsigned)((total4 + 255) / 256), 256, 0, st>>>(p->input, p->gate, p->out, CHW4, HW4, total4,
                                                                             p->input2, p->addend);
    }
    MMU_HIP_LAUNCH_CHECK("gated_mul_fwd");
    return 0;
}

extern "C" int mmu_gated_mul_bwd(const mmu_gated_mul_params *p, void *stream) {
    if (int r = check(p, "gated_mul_bwd")) return r;
    MMU_CHECK(p->dout && (p->dinput || p->dgate || p->dinput2), "gated_mul_bwd: dout and at least one output are required");
    MMU_CHECK((!p->input2 && !p->dinput2) || (p->mode == MMU_GATE_SPATIAL && p->input2),
              "gated_mul_bwd: input2 / dinput2 need the spatial gate (and dinput2 needs input2)");
    const void *ptrs[] = {p->input, p->gate, p->dout, p->dinput, p->dgate, p->input2, p->dinput2};
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
        const int nchunk = (p->channels + GM_CCH - 1) / GM_CCH;
        MMU_CHECK(nchunk == 1 || !p->dgate || (p->workspace && ((uintptr_t)p->workspace & 15) == 0),
                  "gated_mul_bwd: the spatial gate's dgate needs mmu_gated_mul_bwd_workspace_floats() floats of workspace");
        gm_spatial_bwd_kernel<<<dim3((unsigned)((HW4 + 255) / 256), p->batch, nchunk), 256, 0, st>>>(
            p->input, p->gate, p->dout, p->dinput, p->dgate, p->channels, HW4, p->input2, p->dinput2,
            p->dgate ? p->workspace : nullptr);
        if (nchunk > 1 && p->dgate) {
            MMU_HIP_LAUNCH_CHECK("gated_mul_bwd");
            const long n4 = (long)p->batch * HW4;
            gm_chunk_sum_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, st>>>(p->workspace, p->dgate, nchunk, n4);
        }
    }
    MMU_HIP_LAUNCH_CHECK("gated_mul_bwd");
    return 0;
}
