// dice_bce.hip -- the training loss of the reference (top-level loss.py:5-28) in three launches, for gfx950.
//
//   p = sigmoid(x);  loss = 1 - (2 sum(p t) + s) / (sum(p + t) + s) + mean(-(t max(log p, -100) + (1 - t) max(log(1 - p), -100)))
//
// (the Dice sums run over the whole batch; the -100 clamp is nn.BCELoss's).  As ATen ops this is 22 launches forward and
// 14 backward on a 2 M-element map -- sigmoid, two products, three reductions with their fills, a dozen scalar kernels --
// every one at the ~4.8 us dependent-launch floor of the replayed training step: 0.17 ms for 25 MB of traffic.  Here
//   partials: one pass, three sums per workgroup (sum p t, sum p + t, sum bce) in a fixed order;
//   finish  : one workgroup adds the partials in order and writes loss, sum(p t), sum(p + t);
//   backward: d x = g (d bce / d p + d dice / d p) p (1 - p) with ATen's d bce / d p = (p - t) / max((1 - p) p, 1e-12) / n.
// No atomics: bit-reproducible.  float32, contiguous, any shape (n elements).
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

constexpr int DB_THREADS = 256, DB_MAX_BLOCKS = 1024;

__device__ __forceinline__ float db_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ void db_block_sum3(float &a, float &b, float &c, float (*red)[3]) {
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[w][0] = a;
        red[w][1] = b;
        red[w][2] = c;
    }
    __syncthreads();
    a = b = c = 0.f;
    for (int i = 0; i < DB_THREADS / 64; ++i) {
        a += red[i][0];
        b += red[i][1];
        c += red[i][2];
    }
}

__global__ __launch_bounds__(DB_THREADS) void dice_bce_partials_kernel(const float *__restrict__ x, const float *__restrict__ t,
                                                                       float *__restrict__ part, long n) {
    __shared__ float red[DB_THREADS / 64][3];
    float s_pt = 0.f, s_u = 0.f, s_b = 0.f;
    const long stride = (long)gridDim.x * DB_THREADS;
    for (long i = (long)blockIdx.x * DB_THREADS + threadIdx.x; i < n; i += stride) {
        const float p = db_sigmoid(x[i]), tv = t[i];
        s_pt += p * tv;
        s_u += p + tv;
        s_b -= tv * fmaxf(logf(p), -100.f) + (1.f - tv) * fmaxf(logf(1.f - p), -100.f);
    }
    db_block_sum3(s_pt, s_u, s_b, red);
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x] = s_pt;
        part[3 * blockIdx.x + 1] = s_u;
        part[3 * blockIdx.x + 2] = s_b;
    }
}

// out[0] = loss, out[1] = sum(p t), out[2] = sum(p + t)
__global__ __launch_bounds__(DB_THREADS) void dice_bce_finish_kernel(const float *__restrict__ part, int nblocks, float smooth,
                                                                     float inv_n, float *__restrict__ out) {
    __shared__ float red[DB_THREADS / 64][3];
    float a = 0.f, b = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += DB_THREADS) {
        a += part[3 * i];
        b += part[3 * i + 1];
        c += part[3 * i + 2];
    }
    db_block_sum3(a, b, c, red);
    if (threadIdx.x == 0) {
        out[0] = (1.f - (2.f * a + smooth) / (b + smooth)) + c * inv_n;
        out[1] = a;
        out[2] = b;
    }
}

__global__ __launch_bounds__(DB_THREADS) void dice_bce_bwd_kernel(const float *__restrict__ x, const float *__restrict__ t,
                                                                  const float *__restrict__ sums, const float *__restrict__ g,
                                                                  float smooth, float inv_n, float *__restrict__ dx, long n) {
    const long i = (long)blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= n) return;
    const float I2 = 2.f * sums[1] + smooth, U = sums[2] + smooth, go = g[0];
    const float p = db_sigmoid(x[i]), tv = t[i];
    const float q = (1.f - p) * p;
    const float dbce = (p - tv) / fmaxf(q, 1e-12f) * inv_n;
    const float ddice = -(2.f * tv * U - I2) / (U * U);
    dx[i] = go * (dbce + ddice) * q;
}

inline int db_blocks(long n) {
    const long b = (n + 4 * DB_THREADS - 1) / (4 * DB_THREADS);
    return (int)(b < 1 ? 1 : (b > DB_MAX_BLOCKS ? DB_MAX_BLOCKS : b));
}

}  // namespace

extern "C" size_t mmu_dice_bce_workspace_floats(int64_t n) { return n > 0 ? (size_t)3 * db_blocks(n) : 0; }

extern "C" int mmu_dice_bce_fwd(const mmu_dice_bce_params *p, void *stream) {
    MMU_CHECK(p != nullptr && p->n > 0, "dice_bce_fwd: null params or empty tensor");
    MMU_CHECK(p->logits && p->targets && p->workspace && p->out, "dice_bce_fwd: logits, targets, workspace, out are required");
    hipStream_t st = (hipStream_t)stream;
    const int nb = db_blocks(p->n);
    dice_bce_partials_kernel<<<nb, DB_THREADS, 0, st>>>(p->logits, p->targets, p->workspace, p->n);
    MMU_HIP_LAUNCH_CHECK("dice_bce_fwd(partials)");
    dice_bce_finish_kernel<<<1, DB_THREADS, 0, st>>>(p->workspace, nb, p->smooth, 1.f / (float)p->n, p->out);
    MMU_HIP_LAUNCH_CHECK("dice_bce_fwd");
    return 0;
}

extern "C" int mmu_dice_bce_bwd(const mmu_dice_bce_params *p, void *stream) {
    MMU_CHECK(p != nullptr && p->n > 0, "dice_bce_bwd: null params or empty tensor");
    MMU_CHECK(p->logits && p->targets && p->out && p->dloss && p->dlogits,
              "dice_bce_bwd: logits, targets, out (the forward's three floats), dloss, dlogits are required");
    dice_bce_bwd_kernel<<<(unsigned)((p->n + DB_THREADS - 1) / DB_THREADS), DB_THREADS, 0, (hipStream_t)stream>>>(
        p->logits, p->targets, p->out, p->dloss, p->smooth, 1.f / (float)p->n, p->dlogits, p->n);
    MMU_HIP_LAUNCH_CHECK("dice_bce_bwd");
    return 0;
}
