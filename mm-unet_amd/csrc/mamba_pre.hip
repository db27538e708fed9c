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

// ---- backward mirror: everything between the scan backward and the conv1d backward of a small block ----------
// selective_scan_interface.py:268-277: d dt = W_dt^T d delta; dW_dt = d delta . dt^T; dW_x = d x_dbl . conv^T;
// d conv += W_x^T d x_dbl -- two split-K products with their reductions, a K = 6 GEMM and a K = 33 addmm (6-8
// launches of 10-50 us, hipBLASLt at its worst: a [6, 2048] x [2048, 1] product takes 53 us).  Here one thread
// takes 4 tokens, streams the R = 1 + 2N rows of d x_dbl once (row 0 is formed in registers and never stored),
// updates d conv in place and keeps the R*D + D weight-gradient sums in registers; wave_sum4 batches + LDS give
// one partial per workgroup, a second kernel adds the workgroups in fixed order (deterministic, no atomics).
namespace {

// A workgroup owns 256 tokens (lane = 4 consecutive tokens) and its four waves SPLIT THE ROWS of x_dbl: wave 0 takes the
// dt row (built from ddelta) and rows 1..5, waves 1..3 nine rows each.  A row's weight gradients belong to one wave (no
// cross-wave reduction: wave_sum4_swap and a store), the partial d conv_out of waves 1..3 meet wave 0 through LDS.  The
// first form gave every thread all 33 rows: 3,000 instructions on one wave per SIMD, 15-18 us whatever the size.
template <int D, int R>
__global__ __launch_bounds__(256, 1) void mamba_post_small_kernel(const float *__restrict__ ddelta,
                                                                  const float *__restrict__ dt,
                                                                  const float *__restrict__ dxdbl,
                                                                  const float *__restrict__ conv,
                                                                  float *__restrict__ dconv,
                                                                  const float *__restrict__ wx,
                                                                  const float *__restrict__ wdt,
                                                                  float *__restrict__ part, long T) {
    static_assert(R == 33, "row split written for 33 rows");
    constexpr int NV = R * D + D, NV4 = (NV + 3) & ~3;
    constexpr int JW = 9;                                  // rows per wave at most
    __shared__ float4 xch[3][D][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jlo = w == 0 ? 0 : 6 + JW * (w - 1), nrows = w == 0 ? 6 : JW;
    const long ngroups = T >> 2;
    long gi = (long)blockIdx.x * 64 + lane;
    const float live = gi < ngroups ? 1.f : 0.f;           // lanes past the end re-read the last group with weight 0
    gi = gi < ngroups ? gi : ngroups - 1;
    const long col = gi * 4;

    float4 rows[JW];
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
        rows[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (jj < nrows && jlo + jj > 0) rows[jj] = *reinterpret_cast<const float4 *>(dxdbl + (long)(jlo + jj) * T + col);
    }
    float cv[D][4], pc[D][4];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float4 c4 = *reinterpret_cast<const float4 *>(conv + d * T + col);
        cv[d][0] = c4.x * live; cv[d][1] = c4.y * live; cv[d][2] = c4.z * live; cv[d][3] = c4.w * live;
        pc[d][0] = pc[d][1] = pc[d][2] = pc[d][3] = 0.f;
    }
    float vdt[D];
    if (w == 0) {   // the dt row = sum_d W_dt[d] ddelta[d];  dW_dt[d] = sum_t ddelta[d] dt
        const float4 t4 = *reinterpret_cast<const float4 *>(dt + col);
        const float dtv[4] = {t4.x, t4.y, t4.z, t4.w};
        float r0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float4 e4 = *reinterpret_cast<const float4 *>(ddelta + d * T + col);
            const float4 g4 = *reinterpret_cast<const float4 *>(dconv + d * T + col);
            const float e[4] = {e4.x, e4.y, e4.z, e4.w};
            pc[d][0] = g4.x; pc[d][1] = g4.y; pc[d][2] = g4.z; pc[d][3] = g4.w;
            const float wv = wdt[d];
            vdt[d] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r0[i] = fmaf(wv, e[i], r0[i]);
                vdt[d] = fmaf(e[i] * live, dtv[i], vdt[d]);
            }
        }
        rows[0] = make_float4(r0[0], r0[1], r0[2], r0[3]);
    }
    float v[JW * D];
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
        const float row[4] = {rows[jj].x, rows[jj].y, rows[jj].z, rows[jj].w};
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float wv = jj < nrows ? wx[(jlo + jj) * D + d] : 0.f;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pc[d][i] = fmaf(wv, row[i], pc[d][i]);
                acc = fmaf(row[i], cv[d][i], acc);
            }
            v[jj * D + d] = acc;
        }
    }
    if (w > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) xch[w - 1][d][lane] = make_float4(pc[d][0], pc[d][1], pc[d][2], pc[d][3]);
    }
    // this wave's rows of dW_x (and dW_dt from wave 0): sums over the 64 lanes land in lanes 12..15 of a group of four
    float *pb = part + (long)blockIdx.x * NV4;
#pragma unroll
    for (int i = 0; i < JW * D; i += 4) {
        const float r = wave_sum4_swap(v[i], i + 1 < JW * D ? v[i + 1] : 0.f, i + 2 < JW * D ? v[i + 2] : 0.f,
                                       i + 3 < JW * D ? v[i + 3] : 0.f);
        const int k = i + lane - 12;
        if (lane >= 12 && lane < 16 && k < nrows * D) pb[jlo * D + k] = r;
    }
    if (w == 0) {
#pragma unroll
        for (int i = 0; i < D; i += 4) {
            const float r = wave_sum4_swap(vdt[i], i + 1 < D ? vdt[i + 1] : 0.f, i + 2 < D ? vdt[i + 2] : 0.f,
                                           i + 3 < D ? vdt[i + 3] : 0.f);
            const int k = i + lane - 12;
            if (lane >= 12 && lane < 16 && k < D) pb[R * D + k] = r;
        }
    }
    __syncthreads();
    if (w == 0 && live != 0.f) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float4 a = xch[0][d][lane], b = xch[1][d][lane], c = xch[2][d][lane];
            *reinterpret_cast<float4 *>(dconv + d * T + col) =
                make_float4(pc[d][0] + a.x + b.x + c.x, pc[d][1] + a.y + b.y + c.y, pc[d][2] + a.z + b.z + c.z,
                            pc[d][3] + a.w + b.w + c.w);
        }
    }
}

// one wave per result: lanes stride over the workgroups' partials, fixed-order butterfly at the end
template <int D, int R>
__global__ __launch_bounds__(256) void mamba_post_small_sum_kernel(const float *__restrict__ part, float *__restrict__ dwx,
                                                                   float *__restrict__ dwdt, int nblk) {
    constexpr int NV = R * D + D, NV4 = (NV + 3) & ~3;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= NV) return;
    float s = 0.f;
    for (int k = lane; k < nblk; k += 64) s += part[(long)k * NV4 + i];
    s = wave_sum(s);
    if (lane == 0) {
        if (i < R * D) dwx[i] = s;
        else dwdt[i - R * D] = s;
    }
}

int post_blocks(long tokens) {   // 256 tokens per workgroup
    const long g = (tokens / 4 + 63) / 64;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" size_t mmu_mamba_post_small_workspace_floats(int dim, int rows, long tokens) {
    if (tokens <= 0 || dim <= 0 || rows <= 0) return 0;
    return (size_t)post_blocks(tokens) * ((rows * dim + dim + 3) & ~3);
}

extern "C" int mmu_mamba_post_small(const mmu_mamba_post_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "mamba_post_small: null params");
    MMU_CHECK((p->dim == 2 || p->dim == 6) && p->rows == 33,
              "mamba_post_small: inner width 2 or 6 and 33 rows (dt_rank 1, d_state 16) required (got %d, %d)", p->dim,
              p->rows);
    MMU_CHECK(p->tokens > 0 && p->tokens % 4 == 0 && p->tokens <= (1L << 30),
              "mamba_post_small: tokens must be a positive multiple of 4 (at most 2^30)");
    MMU_CHECK(p->ddelta && p->dt && p->dx_dbl && p->conv_out && p->dconv_out && p->x_proj_weight &&
                  p->dt_proj_weight && p->dx_proj_weight && p->ddt_proj_weight && p->workspace,
              "mamba_post_small: every pointer is required");
    const void *ptrs[] = {p->ddelta, p->dt, p->dx_dbl, p->conv_out, p->dconv_out};
    for (const void *q : ptrs) MMU_CHECK(((uintptr_t)q & 15) == 0, "mamba_post_small: tensors must be 16-byte aligned");
    const int nblk = post_blocks(p->tokens);
    hipStream_t st = (hipStream_t)stream;
    if (p->dim == 6) {
        mamba_post_small_kernel<6, 33><<<nblk, 256, 0, st>>>(p->ddelta, p->dt, p->dx_dbl, p->conv_out, p->dconv_out,
                                                             p->x_proj_weight, p->dt_proj_weight, p->workspace, p->tokens);
        MMU_HIP_LAUNCH_CHECK("mamba_post_small");
        const long job[8] = {3, (long)p->workspace, (long)p->dx_proj_weight, (long)p->ddt_proj_weight, 33 * 6, nblk,
                             (33 * 6 + 6 + 3) & ~3, 33 * 6 + 6};
        if (mmu_defer_job(job)) return 0;   // (deferred_reduce.hip: with the other weight-gradient sums of the pass)
        mamba_post_small_sum_kernel<6, 33><<<(33 * 6 + 6 + 3) / 4, 256, 0, st>>>(p->workspace, p->dx_proj_weight, p->ddt_proj_weight, nblk);
    } else {
        mamba_post_small_kernel<2, 33><<<nblk, 256, 0, st>>>(p->ddelta, p->dt, p->dx_dbl, p->conv_out, p->dconv_out,
                                                             p->x_proj_weight, p->dt_proj_weight, p->workspace, p->tokens);
        MMU_HIP_LAUNCH_CHECK("mamba_post_small");
        const long job[8] = {3, (long)p->workspace, (long)p->dx_proj_weight, (long)p->ddt_proj_weight, 33 * 2, nblk,
                             (33 * 2 + 2 + 3) & ~3, 33 * 2 + 2};
        if (mmu_defer_job(job)) return 0;
        mamba_post_small_sum_kernel<2, 33><<<(33 * 2 + 2 + 3) / 4, 256, 0, st>>>(p->workspace, p->dx_proj_weight, p->ddt_proj_weight, nblk);
    }
    MMU_HIP_LAUNCH_CHECK("mamba_post_small(sum)");
    return 0;
}
