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

}  // namespace

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
