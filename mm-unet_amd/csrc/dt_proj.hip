// dt_proj.hip -- Mamba's dt_proj on tokens-last operands: delta = W_dt (D x R) . dt (R x T) and its input gradient
// d dt = W_dt^T . d delta, R = dt_rank <= 8 (mamba_ssm/ops/selective_scan_interface.py:182 forward, :274 backward;
// mamba_simple.py:87 dt_proj).
//
// These are the two skinniest products of the model: K = 4 (forward) and N = 4 (backward) against up to 524,288 tokens
// and 128 channels.  A GEMM library pads them to its tile sizes and ran them at 3.6-4 TB/s (68 us for 276 MB); they are
// pure streaming: every thread owns four tokens, keeps the R dt values (forward) or R sums (backward) of its tokens in
// registers and walks the D rows, whose weights are wave-uniform (scalar loads).  float32; rows 16-byte aligned.
// Under bf16 autocast (io_t = bf16_t) the token rows are bfloat16 -- read and written natively, half the bytes of a
// memory-bound kernel --, the weights and all arithmetic stay float32, results round to nearest even at the store.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

// four consecutive tokens of a row: float32 (16 bytes) or bfloat16 (8 bytes; autocast) -- float32 in registers either way
template <typename io_t>
__device__ __forceinline__ float4 ld4(const io_t *row, long i) {
    if constexpr (sizeof(io_t) == 4) {
        return reinterpret_cast<const float4 *>(row)[i];
    } else {
        const uint2 v = reinterpret_cast<const uint2 *>(row)[i];
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xffff0000u));
    }
}
template <typename io_t>
__device__ __forceinline__ void st4(io_t *row, long i, float4 v) {
    if constexpr (sizeof(io_t) == 4) {
        reinterpret_cast<float4 *>(row)[i] = v;
    } else {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
        const bf2 a = {(__bf16)v.x, (__bf16)v.y}, b = {(__bf16)v.z, (__bf16)v.w};   // round to nearest even
        reinterpret_cast<uint2 *>(row)[i] = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    }
}

template <int R, typename io_t>
__global__ __launch_bounds__(256) void dt_proj_fwd_kernel(const io_t *__restrict__ dt, long dt_rs, const float *__restrict__ W,
                                                          long w_ld, io_t *__restrict__ out, long out_rs, int D, long T4,
                                                          int d_per_block) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= T4) return;
    float4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = ld4(dt + r * dt_rs, i);
    const int d0 = blockIdx.y * d_per_block, d1 = min(d0 + d_per_block, D);
    for (int d = d0; d < d1; ++d) {
        const float *w = W + (long)d * w_ld;   // uniform: scalar loads
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float c = w[r];
            o.x = fmaf(c, v[r].x, o.x); o.y = fmaf(c, v[r].y, o.y); o.z = fmaf(c, v[r].z, o.z); o.w = fmaf(c, v[r].w, o.w);
        }
        st4(out + (long)d * out_rs, i, o);
    }
}

template <int R, typename io_t>
__global__ __launch_bounds__(256) void dt_proj_bwd_kernel(const io_t *__restrict__ g, long g_rs, const float *__restrict__ W,
                                                          long w_ld, io_t *__restrict__ ddt, long ddt_rs, int D, long T4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= T4) return;
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    int d = 0;
    for (; d + 8 <= D; d += 8) {   // eight rows in flight per thread
        float4 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = ld4(g + (long)(d + j) * g_rs, i);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float *w = W + (long)(d + j) * w_ld;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float c = w[r];
                acc[r].x = fmaf(c, x[j].x, acc[r].x); acc[r].y = fmaf(c, x[j].y, acc[r].y);
                acc[r].z = fmaf(c, x[j].z, acc[r].z); acc[r].w = fmaf(c, x[j].w, acc[r].w);
            }
        }
    }
    for (; d < D; ++d) {
        const float4 x = ld4(g + (long)d * g_rs, i);
        const float *w = W + (long)d * w_ld;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float c = w[r];
            acc[r].x = fmaf(c, x.x, acc[r].x); acc[r].y = fmaf(c, x.y, acc[r].y);
            acc[r].z = fmaf(c, x.z, acc[r].z); acc[r].w = fmaf(c, x.w, acc[r].w);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) st4(ddt + r * ddt_rs, i, acc[r]);
}

// ---- x_proj on tokens-last operands: x_dbl (RW x T) = W_x (RW x D) . conv (D x T), and d conv += W_x^T . d x_dbl -------
// RW = dt_rank + 2 d_state (36 at d_state 16), D = d_inner (128): per token 4,608 multiply-adds against 656 bytes moved
// (forward) -- at the chip's 157 TFLOP/s of plain fp32 that is ~30 us of arithmetic under ~60-110 us of HBM time, so the
// product runs on the vector pipe in float32 (exact products, no operand split) as a streaming kernel: a thread owns
// four tokens, the weights sit transposed in LDS and are read as wave-wide broadcasts.  On the matrix-core GEMM
// (gemm_tokens_mfma.hip) the same products are barrier-bound: K is 8 or 3 chunks deep, and the accumulate form of the
// backward spends its time in the read-modify-write epilogue (97 / 250 us; the library: 97 / 157 us).
constexpr int XP_MAXR = 40;

template <int RW, typename io_t>
__global__ __launch_bounds__(256) void x_proj_fwd_kernel(const io_t *__restrict__ x, long x_rs, const float *__restrict__ W,
                                                         long w_ld, io_t *__restrict__ out, long out_rs, int D, long T4) {
    extern __shared__ __attribute__((aligned(16))) float sW[];   // [D][RWP]: W transposed, rows padded to a multiple of 4
    constexpr int RWP = (RW + 3) & ~3;
    for (int i = threadIdx.x; i < D * RWP; i += 256) {   // coalesced reads of W's rows, transposed into LDS
        const int r = i / D, d = i - r * D;
        sW[d * RWP + r] = r < RW ? W[(long)r * w_ld + d] : 0.f;
    }
    __syncthreads();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < T4; i += (long)gridDim.x * 256) {
    float4 acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int d0 = 0; d0 < D; d0 += 4) {   // four rows in flight
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ld4(x + (long)(d0 + j) * x_rs, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 *w4 = reinterpret_cast<const float4 *>(sW + (d0 + j) * RWP);
#pragma unroll
            for (int q = 0; q < RWP / 4; ++q) {
                const float4 w = w4[q];   // the same address in every lane: a broadcast
                const float c[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * q + k;
                    if (r < RW) {
                        acc[r].x = fmaf(c[k], v[j].x, acc[r].x); acc[r].y = fmaf(c[k], v[j].y, acc[r].y);
                        acc[r].z = fmaf(c[k], v[j].z, acc[r].z); acc[r].w = fmaf(c[k], v[j].w, acc[r].w);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) st4(out + (long)r * out_rs, i, acc[r]);
    }
}

// d conv[d][t] += sum_r W[r][d] * g[r][t]
template <int RW, typename io_t>
__global__ __launch_bounds__(256) void x_proj_bwd_kernel(const io_t *__restrict__ g, long g_rs, const float *__restrict__ W,
                                                         long w_ld, io_t *__restrict__ dx, long dx_rs, int D, long T4) {
    extern __shared__ __attribute__((aligned(16))) float sW[];
    constexpr int RWP = (RW + 3) & ~3;
    for (int i = threadIdx.x; i < D * RWP; i += 256) {   // coalesced reads of W's rows, transposed into LDS
        const int r = i / D, d = i - r * D;
        sW[d * RWP + r] = r < RW ? W[(long)r * w_ld + d] : 0.f;
    }
    __syncthreads();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < T4; i += (long)gridDim.x * 256) {
    float4 v[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) v[r] = ld4(g + (long)r * g_rs, i);
    for (int d0 = 0; d0 < D; d0 += 4) {
        float4 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = ld4(dx + (long)(d0 + j) * dx_rs, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 *w4 = reinterpret_cast<const float4 *>(sW + (d0 + j) * RWP);
#pragma unroll
            for (int q = 0; q < RWP / 4; ++q) {
                const float4 w = w4[q];
                const float c[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * q + k;
                    if (r < RW) {
                        o[j].x = fmaf(c[k], v[r].x, o[j].x); o[j].y = fmaf(c[k], v[r].y, o[j].y);
                        o[j].z = fmaf(c[k], v[r].z, o[j].z); o[j].w = fmaf(c[k], v[r].w, o[j].w);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) st4(dx + (long)(d0 + j) * dx_rs, i, o[j]);
    }
    }
}

int check_xp(const mmu_x_proj_params *p, const char *who) {
    MMU_CHECK(p != nullptr, "%s: null params", who);
    MMU_CHECK(p->rows >= 1 && p->rows <= XP_MAXR && p->dim > 0 && p->dim % 4 == 0 && p->dim <= 1024 && p->tokens > 0,
              "%s: rows must be 1..%d, dim a multiple of 4 up to 1024, tokens positive (got %d, %d, %ld)", who, XP_MAXR,
              p->rows, p->dim, (long)p->tokens);
    MMU_CHECK(p->x && p->weight && p->x_dbl, "%s: x, weight, x_dbl are required", who);
    MMU_CHECK(p->io_dtype == MMU_DTYPE_F32 || p->io_dtype == MMU_DTYPE_BF16, "%s: io_dtype must be float32 or bfloat16", who);
    const uintptr_t am = p->io_dtype == MMU_DTYPE_BF16 ? 7 : 15;
    MMU_CHECK(p->tokens % 4 == 0 && p->x_rs % 4 == 0 && p->x_dbl_rs % 4 == 0 && ((uintptr_t)p->x & am) == 0 &&
                  ((uintptr_t)p->x_dbl & am) == 0,
              "%s: tokens and the row strides must be multiples of 4, x and x_dbl %d-byte aligned", who, (int)am + 1);
    return 0;
}

int check(const mmu_dt_proj_params *p, const char *who) {
    MMU_CHECK(p != nullptr, "%s: null params", who);
    MMU_CHECK(p->rank >= 1 && p->rank <= 8 && p->dim > 0 && p->tokens > 0,
              "%s: rank must be 1..8, dim and tokens positive (got %d, %d, %ld)", who, p->rank, p->dim, (long)p->tokens);
    MMU_CHECK(p->dt && p->weight && p->delta, "%s: dt, weight, delta are required", who);
    MMU_CHECK(p->io_dtype == MMU_DTYPE_F32 || p->io_dtype == MMU_DTYPE_BF16, "%s: io_dtype must be float32 or bfloat16", who);
    const uintptr_t am = p->io_dtype == MMU_DTYPE_BF16 ? 7 : 15;
    MMU_CHECK(p->tokens % 4 == 0 && p->dt_rs % 4 == 0 && p->delta_rs % 4 == 0 && ((uintptr_t)p->dt & am) == 0 &&
                  ((uintptr_t)p->delta & am) == 0,
              "%s: tokens and the row strides must be multiples of 4, dt and delta %d-byte aligned", who, (int)am + 1);
    MMU_CHECK(p->tokens / 4 / 256 < (1L << 31), "%s: too many tokens", who);
    return 0;
}

}  // namespace

#define DT_DISPATCH(R_, ...)                                 \
    switch (R_) {                                            \
        case 1: { constexpr int R = 1; __VA_ARGS__ } break;  \
        case 2: { constexpr int R = 2; __VA_ARGS__ } break;  \
        case 3: { constexpr int R = 3; __VA_ARGS__ } break;  \
        case 4: { constexpr int R = 4; __VA_ARGS__ } break;  \
        case 5: { constexpr int R = 5; __VA_ARGS__ } break;  \
        case 6: { constexpr int R = 6; __VA_ARGS__ } break;  \
        case 7: { constexpr int R = 7; __VA_ARGS__ } break;  \
        default: { constexpr int R = 8; __VA_ARGS__ } break; \
    }

extern "C" int mmu_dt_proj_fwd(const mmu_dt_proj_params *p, void *stream) {
    if (int r = check(p, "dt_proj_fwd")) return r;
    const long T4 = p->tokens / 4;
    const unsigned bx = (unsigned)((T4 + 255) / 256);
    // few tokens: cut the channels over the y axis so that the launch still covers the chip
    int ny = 1;
    while (ny < 8 && (long)bx * ny < 2 * mmu_cu_count() && p->dim / (ny * 2) >= 8) ny *= 2;
    const int dpb = (p->dim + ny - 1) / ny;
    dim3 grid(bx, (unsigned)((p->dim + dpb - 1) / dpb));
    if (p->io_dtype == MMU_DTYPE_BF16) {
        DT_DISPATCH(p->rank, dt_proj_fwd_kernel<R, bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(
            (const bf16_t *)p->dt, p->dt_rs, p->weight, p->w_ld, (bf16_t *)p->delta, p->delta_rs, p->dim, T4, dpb););
    } else {
        DT_DISPATCH(p->rank, dt_proj_fwd_kernel<R, float><<<grid, 256, 0, (hipStream_t)stream>>>(
            (const float *)p->dt, p->dt_rs, p->weight, p->w_ld, (float *)p->delta, p->delta_rs, p->dim, T4, dpb););
    }
    MMU_HIP_LAUNCH_CHECK("dt_proj_fwd");
    return 0;
}

extern "C" int mmu_dt_proj_bwd(const mmu_dt_proj_params *p, void *stream) {
    if (int r = check(p, "dt_proj_bwd")) return r;
    const long T4 = p->tokens / 4;
    const unsigned bx = (unsigned)((T4 + 255) / 256);
    // here `delta` is the incoming gradient d delta (read) and `dt` receives d dt (written)
    if (p->io_dtype == MMU_DTYPE_BF16) {
        DT_DISPATCH(p->rank, dt_proj_bwd_kernel<R, bf16_t><<<bx, 256, 0, (hipStream_t)stream>>>(
            (const bf16_t *)p->delta, p->delta_rs, p->weight, p->w_ld, (bf16_t *)const_cast<void *>(p->dt), p->dt_rs, p->dim, T4););
    } else {
        DT_DISPATCH(p->rank, dt_proj_bwd_kernel<R, float><<<bx, 256, 0, (hipStream_t)stream>>>(
            (const float *)p->delta, p->delta_rs, p->weight, p->w_ld, (float *)const_cast<void *>(p->dt), p->dt_rs, p->dim, T4););
    }
    MMU_HIP_LAUNCH_CHECK("dt_proj_bwd");
    return 0;
}

#define XP_DISPATCH(R_, ...)                                   \
    switch (R_) {                                              \
        case 33: { constexpr int RW = 33; __VA_ARGS__ } break; \
        case 34: { constexpr int RW = 34; __VA_ARGS__ } break; \
        case 36: { constexpr int RW = 36; __VA_ARGS__ } break; \
        case 40: { constexpr int RW = 40; __VA_ARGS__ } break; \
        default: return mmu_fail("x_proj: rows = dt_rank + 2 * d_state must be 33, 34, 36 or 40 (got %d)", R_); \
    }

// x_dbl (rows x tokens) = weight (rows x dim) . x (dim x tokens)            (selective_scan_interface.py:181)
extern "C" int mmu_x_proj_fwd(const mmu_x_proj_params *p, void *stream) {
    if (int r = check_xp(p, "x_proj_fwd")) return r;
    const long T4 = p->tokens / 4;
    const long nb = (T4 + 255) / 256, cap = 8L * mmu_cu_count();   // a block stages the weight once and walks its token groups
    const unsigned bx = (unsigned)(nb < cap ? nb : cap);
    const size_t lds = sizeof(float) * (size_t)p->dim * ((p->rows + 3) & ~3);
    MMU_CHECK(lds <= 64 * 1024, "x_proj_fwd: the weight does not fit 64 KiB of LDS");
    if (p->io_dtype == MMU_DTYPE_BF16) {
        XP_DISPATCH(p->rows, x_proj_fwd_kernel<RW, bf16_t><<<bx, 256, lds, (hipStream_t)stream>>>(
            (const bf16_t *)p->x, p->x_rs, p->weight, p->w_ld, (bf16_t *)p->x_dbl, p->x_dbl_rs, p->dim, T4););
    } else {
        XP_DISPATCH(p->rows, x_proj_fwd_kernel<RW, float><<<bx, 256, lds, (hipStream_t)stream>>>(
            (const float *)p->x, p->x_rs, p->weight, p->w_ld, (float *)p->x_dbl, p->x_dbl_rs, p->dim, T4););
    }
    MMU_HIP_LAUNCH_CHECK("x_proj_fwd");
    return 0;
}

// x (dim x tokens) += weight^T (dim x rows) . x_dbl (rows x tokens): `x` holds d conv and is updated in place, `x_dbl`
// holds d x_dbl                                                               (selective_scan_interface.py:277)
extern "C" int mmu_x_proj_bwd(const mmu_x_proj_params *p, void *stream) {
    if (int r = check_xp(p, "x_proj_bwd")) return r;
    const long T4 = p->tokens / 4;
    const long nb = (T4 + 255) / 256, cap = 8L * mmu_cu_count();
    const unsigned bx = (unsigned)(nb < cap ? nb : cap);
    const size_t lds = sizeof(float) * (size_t)p->dim * ((p->rows + 3) & ~3);
    MMU_CHECK(lds <= 64 * 1024, "x_proj_bwd: the weight does not fit 64 KiB of LDS");
    if (p->io_dtype == MMU_DTYPE_BF16) {
        XP_DISPATCH(p->rows, x_proj_bwd_kernel<RW, bf16_t><<<bx, 256, lds, (hipStream_t)stream>>>(
            (const bf16_t *)p->x_dbl, p->x_dbl_rs, p->weight, p->w_ld, (bf16_t *)const_cast<void *>(p->x), p->x_rs, p->dim, T4););
    } else {
        XP_DISPATCH(p->rows, x_proj_bwd_kernel<RW, float><<<bx, 256, lds, (hipStream_t)stream>>>(
            (const float *)p->x_dbl, p->x_dbl_rs, p->weight, p->w_ld, (float *)const_cast<void *>(p->x), p->x_rs, p->dim, T4););
    }
    MMU_HIP_LAUNCH_CHECK("x_proj_bwd");
    return 0;
}
