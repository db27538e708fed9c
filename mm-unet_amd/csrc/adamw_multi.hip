// adamw_multi.hip -- ONE launch of decoupled-weight-decay Adam over many parameter tensors, for gfx950.
//
// Where it sits: the optimizer step of the training step (train.py:197-201: timm create_optimizer_v2(opt="adamw",
// betas=(0.9, 0.95), weight_decay=0.05) -> torch.optim.AdamW).  torch's fused AdamW walks MM-UNet's 1,069 live
// parameters in 30 multi-tensor launches (its launch arguments hold a few dozen tensor addresses each) plus 11 for the
// step counters: 41 launches, 0.6 ms of a 37.5 ms step, for 270 MB of traffic.  Here the addresses live in a device
// table that is built once (inside a captured training step parameters, gradients and optimizer state are static), and
// a work list of (tensor, chunk) pairs gives every workgroup 4,096 elements: two launches (counters, update).
//
// Arithmetic = torch/aten/src/ATen/native/cuda/fused_adam_utils.cuh (adam_math, ADAMW, no amsgrad / maximize):
//     step <- step + 1;   p <- p - lr * wd * p;   m <- m + (1 - b1) (g - m);   v <- b2 v + (1 - b2) g g
//     p <- p - (lr / (1 - b1^step)) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)
// bias corrections in double, element math in float.
#include <math.h>
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

constexpr int CHUNK = 4096, COLS = 8;   // table row: {param, grad, exp_avg, exp_avg_sq, step, numel, lr, weight_decay bits}

__global__ __launch_bounds__(256) void adamw_steps_kernel(const long *__restrict__ table, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        float *step = reinterpret_cast<float *>(table[(long)i * COLS + 4]);
        *step += 1.f;
    }
}

__global__ __launch_bounds__(256) void adamw_multi_kernel(const long *__restrict__ table, const int *__restrict__ work,
                                                          double beta1, double beta2, float eps) {
    const int t = work[2 * blockIdx.x], chunk = work[2 * blockIdx.x + 1];
    const long *row = table + (long)t * COLS;
    float *__restrict__ p = reinterpret_cast<float *>(row[0]);
    const float *__restrict__ g = reinterpret_cast<const float *>(row[1]);
    float *__restrict__ m = reinterpret_cast<float *>(row[2]);
    float *__restrict__ v = reinterpret_cast<float *>(row[3]);
    const float step = *reinterpret_cast<const float *>(row[4]);   // already incremented (adamw_steps_kernel)
    const long n = row[5];
    const float lr = *reinterpret_cast<const float *>(row[6]);
    const float wd = __builtin_bit_cast(float, (int)row[7]);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float b1c = (float)(1.0 - beta1), b2 = (float)beta2, b2c = (float)(1.0 - beta2);
    const float decay = lr * wd;
    const long lo = (long)chunk * CHUNK, hi = lo + CHUNK < n ? lo + CHUNK : n;
    auto one = [&](float pv, float gv, float &mv, float &vv) {
        pv -= decay * pv;
        mv = mv + b1c * (gv - mv);
        vv = b2 * vv + b2c * gv * gv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        return pv - step_size * mv / denom;
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (vec) {
        for (long i = lo + 4 * threadIdx.x; i + 3 < hi; i += 4 * 256) {
            float4 pv = *reinterpret_cast<const float4 *>(p + i), mv = *reinterpret_cast<const float4 *>(m + i);
            float4 vv = *reinterpret_cast<const float4 *>(v + i);
            const float4 gv = *reinterpret_cast<const float4 *>(g + i);
            pv.x = one(pv.x, gv.x, mv.x, vv.x); pv.y = one(pv.y, gv.y, mv.y, vv.y);
            pv.z = one(pv.z, gv.z, mv.z, vv.z); pv.w = one(pv.w, gv.w, mv.w, vv.w);
            *reinterpret_cast<float4 *>(p + i) = pv;
            *reinterpret_cast<float4 *>(m + i) = mv;
            *reinterpret_cast<float4 *>(v + i) = vv;
        }
        const long tail = lo + ((hi - lo) & ~3L);
        for (long i = tail + threadIdx.x; i < hi; i += 256) {
            float mv = m[i], vv = v[i];
            p[i] = one(p[i], g[i], mv, vv);
            m[i] = mv;
            v[i] = vv;
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += 256) {
            float mv = m[i], vv = v[i];
            p[i] = one(p[i], g[i], mv, vv);
            m[i] = mv;
            v[i] = vv;
        }
    }
}

}  // namespace

extern "C" int mmu_adamw_multi(const mmu_adamw_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "adamw_multi: null params");
    MMU_CHECK(p->n_tensors > 0 && p->n_work > 0 && p->table && p->work, "adamw_multi: table and work list are required");
    MMU_CHECK(p->beta1 >= 0 && p->beta1 < 1 && p->beta2 >= 0 && p->beta2 < 1 && p->eps > 0, "adamw_multi: bad hyper-parameters");
    static_assert(sizeof(long) == sizeof(int64_t), "the table is read as long");
    hipStream_t st = (hipStream_t)stream;
    adamw_steps_kernel<<<(p->n_tensors + 255) / 256, 256, 0, st>>>(reinterpret_cast<const long *>(p->table), p->n_tensors);
    MMU_HIP_LAUNCH_CHECK("adamw_multi(steps)");
    adamw_multi_kernel<<<p->n_work, 256, 0, st>>>(reinterpret_cast<const long *>(p->table), p->work, p->beta1, p->beta2,
                                                  p->eps);
    MMU_HIP_LAUNCH_CHECK("adamw_multi");
    return 0;
}
