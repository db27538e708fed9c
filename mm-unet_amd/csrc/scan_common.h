// scan_common.h -- pieces shared by the selective-scan translation units (selective_scan.hip: chunk-parallel
// kernels; selective_scan_stream.hip: the streaming forward).  Internal to libmmunet_hip.so.
#pragma once
#include "mmu_common.h"

// kernel arguments of every scan kernel (one struct, passed by value)
struct ScanArgs {
    int batch, dim, seqlen, dstate, ngroups, n_chunks, softplus;
    int vec_io;   // u/delta/z/out/... K-groups naturally aligned
    int vec_bc;   // B/C rows K-aligned
    int vec_dbc;  // dB/dC rows K-aligned
    const void *u, *delta, *z, *B, *C, *dout;
    const float *A, *D, *delta_bias;
    void *out, *out_z, *du, *ddelta, *dz;
    float *x;        // [B][D][nc][N][2]  (P, H)
    float *gx;       // [B][D][nc][N][2]  (Q, Gamma)   (bwd)
    float *dB, *dC;  // [B][G][N][L] fp32
    float *part;     // [B][nc][D][N+2]   (bwd partials of dA, dD, dbias)
    long u_bs, u_ds, delta_bs, delta_ds, z_bs, z_ds, out_bs, out_ds, out_z_bs, out_z_ds;
    long dout_bs, dout_ds, du_bs, du_ds, ddelta_bs, ddelta_ds, dz_bs, dz_ds;
    long A_ds, A_ns, B_bs, B_gs, B_ns, C_bs, C_gs, C_ns;
    long dB_bs, dB_gs, dB_ns, dC_bs, dC_gs, dC_ns;
    // backward apply with the channels of a group cut into d_splits ranges (grid.z = ngroups * d_splits): range s
    // writes its dB / dC sums at dB + s * dBC_ss, dC + s * dBC_ss; sum_splits_kernel adds the ranges afterwards
    int d_splits;
    long dBC_ss;
};

namespace {

// ---- state pairs on packed math ----------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f exp2_2(v2f x) { return v2f{fast_exp2(x.x), fast_exp2(x.y)}; }
// x * s[HI] on both halves in one v_pk_mul_f32 (op_sel broadcast): per-token scalars stay packed two to
// a register pair instead of being splatted (16 fewer VGPRs in the forward apply kernel).
// NEVER feed it a fresh v_exp / v_log / v_rcp result: gfx950 needs a wait state between a transcendental
// op and a VALU consumer, and the compiler's hazard recogniser does not look inside inline asm (seen as
// stale odd-state sums in chunk_reduce8).  Exp results go to compiler-generated v_pk_fma only.
template <int HI>
__device__ __forceinline__ v2f mul_bcast(v2f s, v2f x) {
    v2f r;
    if constexpr (HI)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(s), "v"(x));
    else
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(s), "v"(x));
    return r;
}

// x * s[HI] + c in one v_pk_fma_f32 (same rule: no fresh transcendental results as inputs)
template <int HI>
__device__ __forceinline__ v2f fma_bcast(v2f s, v2f x, v2f c) {
    v2f r;
    if constexpr (HI)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(s), "v"(x), "v"(c));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(s), "v"(x), "v"(c));
    return r;
}

// quarter q of lane l of pair `pr` of an [pair][4][64] float4 tile (see stage_pair8 in selective_scan.hip):
//     (row 2p [8l+2q], row 2p+1 [8l+2q], row 2p [8l+2q+1], row 2p+1 [8l+2q+1])
__device__ __forceinline__ void pair8_read(const float *tile, int pr, int lane, v2f (&v)[8]) {
    const float4 *src = reinterpret_cast<const float4 *>(tile) + pr * 256 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 f = src[q * 64];
        v[2 * q] = v2f{f.x, f.y};
        v[2 * q + 1] = v2f{f.z, f.w};
    }
}

// ---- host helpers --------------------------------------------------------------------------------------
template <typename F>
int set_lds(F kernel, size_t bytes) {
    if (bytes > 160 * 1024) return mmu_fail("selective_scan: needs %zu B of LDS (> 160 KiB)", bytes);
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return mmu_fail("hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    return 0;
}

// runtime bool -> template bool
#define MMU_BOOL(cond, NAME, ...)        \
    do {                                 \
        if (cond) {                      \
            constexpr bool NAME = true;  \
            __VA_ARGS__                  \
        } else {                         \
            constexpr bool NAME = false; \
            __VA_ARGS__                  \
        }                                \
    } while (0)

}  // namespace

// streaming forward (selective_scan_stream.hip): returns 1 if it took the call, 0 if the shape is not its,
// < 0 on error
int mmu_scan_fwd_stream(const ScanArgs &a, int dtype, hipStream_t st);

// backward apply on 512-token tiles, one state pair per wave (selective_scan_bwd_w8.hip): same return convention;
// its dA / dD / dbias partials have their own layout and reduction
int mmu_scan_bwd_apply_w8(const ScanArgs &a, int dtype, hipStream_t st);
int mmu_scan_bwd_reduce_w8(const float *part8, float *part2, int batch, int dim, int seqlen, float *dA, float *dD,
                           float *dbias, const float *Asc, hipStream_t st);
