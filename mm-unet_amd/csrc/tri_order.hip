// tri_order.hip -- the token re-orderings of the tri-directional ("v3") Mamba block, fused (f2, first step).
//
// requirements/mamba_simple.py:212-270 runs three scans over the same xz: as is, token-reversed
// (`flip([-1])`) and slice-interleaved (token i of slice s -> position i*nslices + s, :245-247), then adds the
// three results after undoing the re-orderings (:263-270).  As tensor ops that is flip + permute-copy on
// the way in, flip + permute-copy + two adds on the way out, and the mirror image in the backward pass:
// 6 ms of a 71 ms training step, ten passes over [B, d_inner, L] tensors where four suffice.
//   split   : x            -> x_flip, x_slice                 (read 1, write 2)
//   combine : a, b_f, c_s  -> a + unflip(b_f) + unslice(c_s)  (read 3, write 1)
// Each is the other's adjoint.  Thread = one position i of every slice: all global accesses are either
// 4-byte accesses contiguous across the lanes of a wave or one contiguous nslices-float vector per lane.
// Tensors are [rows][L] with dense rows (any of the [C][B][L] / [B][C][L] layouts), float32.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

template <int NS>
__global__ __launch_bounds__(256) void tri_split_kernel(const float *__restrict__ x, float *__restrict__ xf,
                                                        float *__restrict__ xs, int L, int ns_rt) {
    const int ns = NS > 0 ? NS : ns_rt;
    const int Ls = L / ns;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ls) return;
    const long base = (long)blockIdx.y * L;
    if constexpr (NS == 4) {
        float v[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = s * Ls + i;
            v[s] = x[base + t];
            xf[base + L - 1 - t] = v[s];
        }
        *reinterpret_cast<float4 *>(xs + base + (long)i * 4) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int s = 0; s < ns; ++s) {
            const int t = s * Ls + i;
            const float v = x[base + t];
            xf[base + L - 1 - t] = v;
            xs[base + (long)i * ns + s] = v;
        }
    }
}

template <int NS>
__global__ __launch_bounds__(256) void tri_combine_kernel(const float *__restrict__ a, const float *__restrict__ bf,
                                                          const float *__restrict__ cs, float *__restrict__ out, int L,
                                                          int ns_rt) {
    const int ns = NS > 0 ? NS : ns_rt;
    const int Ls = L / ns;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ls) return;
    const long base = (long)blockIdx.y * L;
    if constexpr (NS == 4) {
        const float4 c = *reinterpret_cast<const float4 *>(cs + base + (long)i * 4);
        const float cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = s * Ls + i;
            out[base + t] = a[base + t] + bf[base + L - 1 - t] + cv[s];
        }
    } else {
        for (int s = 0; s < ns; ++s) {
            const int t = s * Ls + i;
            out[base + t] = a[base + t] + bf[base + L - 1 - t] + cs[base + (long)i * ns + s];
        }
    }
}

// nslices in 5..64: the slice interleave is a transpose of the [nslices][L/nslices] view of a row.  A workgroup
// takes 64 positions i of every slice: reads are contiguous along i per slice, the interleaved side is one
// contiguous run of 64*nslices floats, the transpose happens in a padded LDS tile (conflict-free both ways).
template <typename io_t>
__global__ __launch_bounds__(256) void tri_split_tiled_kernel(const io_t *__restrict__ x, io_t *__restrict__ xf,
                                                              io_t *__restrict__ xs, int L, int ns) {
    __shared__ float tile[64][65];
    const int Ls = L / ns;
    const int i0 = blockIdx.x * 64;
    const long base = (long)blockIdx.y * L;
    const int tx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int ni = Ls - i0 < 64 ? Ls - i0 : 64;
    for (int sl = ry; sl < ns; sl += 4) {
        if (tx < ni) {
            const int t = sl * Ls + i0 + tx;
            const io_t raw = x[base + t];
            xf[base + L - 1 - t] = raw;
            tile[sl][tx] = to_f32(raw);
        }
    }
    __syncthreads();
    io_t *dst = xs + base + (long)i0 * ns;
    for (int j = threadIdx.x; j < ni * ns; j += 256) dst[j] = from_f32<io_t>(tile[j % ns][j / ns]);
}

template <typename io_t>
__global__ __launch_bounds__(256) void tri_combine_tiled_kernel(const io_t *__restrict__ a, const io_t *__restrict__ bf,
                                                                const io_t *__restrict__ cs, io_t *__restrict__ out,
                                                                int L, int ns) {
    __shared__ float tile[64][65];
    const int Ls = L / ns;
    const int i0 = blockIdx.x * 64;
    const long base = (long)blockIdx.y * L;
    const int tx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int ni = Ls - i0 < 64 ? Ls - i0 : 64;
    const io_t *src = cs + base + (long)i0 * ns;
    for (int j = threadIdx.x; j < ni * ns; j += 256) tile[j % ns][j / ns] = to_f32(src[j]);
    __syncthreads();
    for (int sl = ry; sl < ns; sl += 4) {
        if (tx < ni) {
            const int t = sl * Ls + i0 + tx;
            out[base + t] = from_f32<io_t>(to_f32(a[base + t]) + to_f32(bf[base + L - 1 - t]) + tile[sl][tx]);
        }
    }
}

int check(const mmu_tri_params *p, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->rows > 0 && p->seqlen > 0 && p->nslices > 0, "%s: empty tensor", name);
    MMU_CHECK(p->seqlen % p->nslices == 0, "%s: seqlen %d must be divisible by nslices %d", name, p->seqlen, p->nslices);
    MMU_CHECK(p->rows < 65536, "%s: more than 65535 rows", name);
    MMU_CHECK(p->dtype == MMU_DTYPE_F32 || (p->dtype == MMU_DTYPE_BF16 && p->nslices > 4 && p->nslices <= 64),
              "%s: float32, or bfloat16 with 5..64 slices (got dtype %d, %d slices)", name, p->dtype, p->nslices);
    return 0;
}

}  // namespace

extern "C" int mmu_tri_split(const mmu_tri_params *p, void *stream) {
    if (int r = check(p, "tri_split")) return r;
    MMU_CHECK(p->a && p->flip && p->slice, "tri_split: a (input), flip and slice (outputs) are required");
    const int Ls = p->seqlen / p->nslices;
    dim3 grid((Ls + 255) / 256, p->rows);
    hipStream_t st = (hipStream_t)stream;
    const float *a = (const float *)p->a;
    float *fl = (float *)p->flip, *sl = (float *)p->slice;
    if (p->dtype == MMU_DTYPE_BF16)
        tri_split_tiled_kernel<bf16_t><<<dim3((Ls + 63) / 64, p->rows), 256, 0, st>>>(
            (const bf16_t *)p->a, (bf16_t *)p->flip, (bf16_t *)p->slice, p->seqlen, p->nslices);
    else if (p->nslices == 4 && ((uintptr_t)p->slice & 15) == 0 && p->seqlen % 4 == 0)
        tri_split_kernel<4><<<grid, 256, 0, st>>>(a, fl, sl, p->seqlen, 4);
    else if (p->nslices > 4 && p->nslices <= 64)
        tri_split_tiled_kernel<float><<<dim3((Ls + 63) / 64, p->rows), 256, 0, st>>>(a, fl, sl, p->seqlen, p->nslices);
    else
        tri_split_kernel<0><<<grid, 256, 0, st>>>(a, fl, sl, p->seqlen, p->nslices);
    MMU_HIP_LAUNCH_CHECK("tri_split");
    return 0;
}

extern "C" int mmu_tri_combine(const mmu_tri_params *p, void *stream) {
    if (int r = check(p, "tri_combine")) return r;
    MMU_CHECK(p->a && p->flip && p->slice && p->out, "tri_combine: a, flip, slice (inputs) and out are required");
    const int Ls = p->seqlen / p->nslices;
    dim3 grid((Ls + 255) / 256, p->rows);
    hipStream_t st = (hipStream_t)stream;
    const float *a = (const float *)p->a, *fl = (const float *)p->flip, *sl = (const float *)p->slice;
    float *out = (float *)p->out;
    if (p->dtype == MMU_DTYPE_BF16)
        tri_combine_tiled_kernel<bf16_t><<<dim3((Ls + 63) / 64, p->rows), 256, 0, st>>>(
            (const bf16_t *)p->a, (const bf16_t *)p->flip, (const bf16_t *)p->slice, (bf16_t *)p->out, p->seqlen, p->nslices);
    else if (p->nslices == 4 && ((uintptr_t)p->slice & 15) == 0 && p->seqlen % 4 == 0)
        tri_combine_kernel<4><<<grid, 256, 0, st>>>(a, fl, sl, out, p->seqlen, 4);
    else if (p->nslices > 4 && p->nslices <= 64)
        tri_combine_tiled_kernel<float><<<dim3((Ls + 63) / 64, p->rows), 256, 0, st>>>(a, fl, sl, out, p->seqlen, p->nslices);
    else
        tri_combine_kernel<0><<<grid, 256, 0, st>>>(a, fl, sl, out, p->seqlen, p->nslices);
    MMU_HIP_LAUNCH_CHECK("tri_combine");
    return 0;
}
