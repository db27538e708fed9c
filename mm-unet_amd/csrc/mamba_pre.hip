// mamba_pre.hip -- everything between in_proj and the selective scan of a SMALL Mamba block, as one kernel.
//
// mamba_inner_fn (mamba_ssm/ops/selective_scan_interface.py:173-210) runs, on the x half of xz:
//     conv1d(width 4) + SiLU  ->  x_dbl = conv @ W_x^T  (r + 2N columns: dt | B | C)  ->  delta = dt @ W_dt^T
// For the 44 Mamba blocks inside MMConv the inner width is 2K = 6 (or 2) and dt_rank is 1: the two "GEMMs"
// have K = 6 and K = 1.  As three launches (conv1d, a 14 us hipBLASLt GEMM, an 8 us outer product) they cost
// 34 us per block forward and 20 us again in the backward recomputation (checkpoint_lvl 1) -- all launch /
// latency bound.  Here one thread takes 4 tokens: 6 x (4 + 3 halo) inputs, 6 x 4 conv outputs, (r + 2N) x 4
// projections with wave-uniform scalar weights, 6 x 4 deltas; every store is 16 bytes per lane.
// Layouts as the fused path keeps them: x / conv_out / delta [D][B][L] (strides passed), x_dbl [r+2N][B*L].
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

template <int D>
__global__ __launch_bounds__(256) void mamba_pre_small_kernel(const float *__restrict__ x, long x_bs, long x_ds,
                                                              const float *__restrict__ cw, const float *__restrict__ cb,
                                                              const float *__restrict__ wx, const float *__restrict__ wdt,
                                                              float *__restrict__ conv_out, long c_bs, long c_ds,
                                                              float *__restrict__ xdbl, float *__restrict__ delta,
                                                              long d_bs, long d_ds, int B, int L, int R) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = q * 4;
    if (t >= L) return;
    const int b = blockIdx.y;
    float conv[D][4];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float *xr = x + (long)b * x_bs + (long)d * x_ds + t;
        const float4 cur = *reinterpret_cast<const float4 *>(xr);
        float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 4) prev = *reinterpret_cast<const float4 *>(xr - 4);
        const float xs[7] = {prev.y, prev.z, prev.w, cur.x, cur.y, cur.z, cur.w};
        const float bv = cb ? cb[d] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float acc = bv;
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = fmaf(cw[d * 4 + m], xs[i + m], acc);
            conv[d][i] = acc * sigmoidf_(acc);
        }
        *reinterpret_cast<float4 *>(conv_out + (long)b * c_bs + (long)d * c_ds + t) =
            make_float4(conv[d][0], conv[d][1], conv[d][2], conv[d][3]);
    }
    // x_dbl rows (wave-uniform weights: scalar loads); row 0 is dt (dt_rank 1)
    const long col = (long)b * L + t;
    const long T = (long)B * L;
    float dt[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < R; ++j) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float wv = wx[j * D + d];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = fmaf(wv, conv[d][i], o[i]);
        }
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dt[i] = o[i];
        }
        if (xdbl) *reinterpret_cast<float4 *>(xdbl + (long)j * T + col) = make_float4(o[0], o[1], o[2], o[3]);
        if (!xdbl) break;  // recomputation: only conv_out and delta are needed (x_dbl was saved)
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float wv = wdt[d];
        *reinterpret_cast<float4 *>(delta + (long)b * d_bs + (long)d * d_ds + t) =
            make_float4(wv * dt[0], wv * dt[1], wv * dt[2], wv * dt[3]);
    }
}

}  // namespace

extern "C" int mmu_mamba_pre_small(const mmu_mamba_pre_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "mamba_pre_small: null params");
    MMU_CHECK(p->dim == 2 || p->dim == 6, "mamba_pre_small: inner width must be 2 or 6 (got %d)", p->dim);
    MMU_CHECK(p->batch > 0 && p->seqlen > 0 && p->seqlen % 4 == 0 && p->rows >= 1,
              "mamba_pre_small: seqlen must be a positive multiple of 4");
    MMU_CHECK(p->x && p->conv_weight && p->x_proj_weight && p->dt_proj_weight && p->conv_out && p->delta,
              "mamba_pre_small: x, conv_weight, x_proj_weight, dt_proj_weight, conv_out, delta are required");
    const long strides[] = {p->x_bs, p->x_ds, p->conv_bs, p->conv_ds, p->delta_bs, p->delta_ds};
    for (long sv : strides) MMU_CHECK(sv % 4 == 0, "mamba_pre_small: strides must be multiples of 4 elements");
    const void *ptrs[] = {p->x, p->conv_out, p->delta, p->x_dbl};
    for (const void *q : ptrs) MMU_CHECK(((uintptr_t)q & 15) == 0, "mamba_pre_small: tensors must be 16-byte aligned");
    dim3 grid((p->seqlen / 4 + 255) / 256, p->batch);
    hipStream_t st = (hipStream_t)stream;
    if (p->dim == 6)
        mamba_pre_small_kernel<6><<<grid, 256, 0, st>>>(p->x, p->x_bs, p->x_ds, p->conv_weight, p->conv_bias,
                                                        p->x_proj_weight, p->dt_proj_weight, p->conv_out, p->conv_bs,
                                                        p->conv_ds, p->x_dbl, p->delta, p->delta_bs, p->delta_ds,
                                                        p->batch, p->seqlen, p->rows);
    else
        mamba_pre_small_kernel<2><<<grid, 256, 0, st>>>(p->x, p->x_bs, p->x_ds, p->conv_weight, p->conv_bias,
                                                        p->x_proj_weight, p->dt_proj_weight, p->conv_out, p->conv_bs,
                                                        p->conv_ds, p->x_dbl, p->delta, p->delta_bs, p->delta_ds,
                                                        p->batch, p->seqlen, p->rows);
    MMU_HIP_LAUNCH_CHECK("mamba_pre_small");
    return 0;
}
