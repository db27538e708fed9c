// mmu_common.h -- shared device/host helpers for libmmunet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define MMU_WAVE 64
#define MMU_LOG2E 1.4426950408889634f
#define MMU_LN2 0.6931471805599453f

// ---------------------------------------------------------------------------
// host-side error reporting (thread-local, read through mmu_last_error())
// ---------------------------------------------------------------------------
extern thread_local char g_mmu_err[512];
int mmu_fail(const char *fmt, ...);
// deferred_reduce.hip: records a final weight-gradient reduction instead of launching it (true) when a deferred scope is open
bool mmu_defer_job(const long (&row)[8]);

#define MMU_CHECK(cond, ...)                       \
    do {                                           \
        if (!(cond)) return mmu_fail(__VA_ARGS__); \
    } while (0)

#define MMU_HIP_LAUNCH_CHECK(what)                                                        \
    do {                                                                                  \
        hipError_t e__ = hipGetLastError();                                               \
        if (e__ != hipSuccess) return mmu_fail("%s: %s", what, hipGetErrorString(e__));   \
    } while (0)

// Per-DEVICE one-time setup: one process may drive several GPUs -- a function attribute set while device 0 was
// current is not set on device 1, and the CU count is a property of the device, not of the process.
inline int mmu_cu_count() {
    static int n[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    dev &= 63;
    if (n[dev] == 0) {
        hipDeviceProp_t prop;
        n[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n[dev];
}
template <typename F>
inline hipError_t mmu_set_lds_once(F kernel, int bytes, unsigned long long &done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if ((done_mask >> (dev & 63)) & 1ull) return hipSuccess;
    e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done_mask |= 1ull << (dev & 63);
    return e;
}

// Zero-fill of an accumulation target as a KERNEL, not hipMemsetAsync: under stream capture a memset becomes a memset
// node, and on this ROCm the replays of a graph did not reliably clear the buffer before the kernel that accumulates
// into it (seen as garbage / NaN gradients from the second replay on, at the small sizes where the atomic paths
// run; tools/dbg/replay_diff.py).  A kernel node keeps the stream order.
__global__ __launch_bounds__(256) static void mmu_zero_kernel(float *__restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}
inline hipError_t mmu_zero_async(float *p, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    mmu_zero_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(p, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// I/O element types
// ---------------------------------------------------------------------------
struct bf16_t {
    uint16_t bits;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v.bits) << 16); }

template <typename T>
__device__ __forceinline__ T from_f32(float f);
template <>
__device__ __forceinline__ float from_f32<float>(float f) { return f; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float f) {
    // plain cast -> v_cvt_pk_bf16_f32 (RNE, keeps NaN a NaN)
    __bf16 h = (__bf16)f;
    bf16_t r;
    r.bits = __builtin_bit_cast(uint16_t, h);
    return r;
}

template <typename T, int K>
struct alignas(sizeof(T) * K) Pack {
    T e[K];
};

// Loads K consecutive items at p (items beyond nvalid read as 0).  `vec` = the
// host verified that every such K-group is naturally aligned.
// FULL = compile-time promise that every K-group is in range and aligned (no tail, no branches).
template <typename T, int K, bool FULL = false>
__device__ __forceinline__ void load_k(const T *__restrict__ p, int nvalid, bool vec, float (&o)[K]) {
    if (FULL || (vec && nvalid >= K)) {
        Pack<T, K> v = *reinterpret_cast<const Pack<T, K> *>(p);
#pragma unroll
        for (int i = 0; i < K; ++i) o[i] = to_f32(v.e[i]);
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) o[i] = (i < nvalid) ? to_f32(p[i]) : 0.f;
    }
}

template <typename T, int K, bool FULL = false>
__device__ __forceinline__ void store_k(T *__restrict__ p, int nvalid, bool vec, const float (&v)[K]) {
    if (FULL || (vec && nvalid >= K)) {
        Pack<T, K> o;
#pragma unroll
        for (int i = 0; i < K; ++i) o.e[i] = from_f32<T>(v[i]);
        *reinterpret_cast<Pack<T, K> *>(p) = o;
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (i < nvalid) p[i] = from_f32<T>(v[i]);
    }
}

// ---- buffer-resource I/O -------------------------------------------------------------------------
// A stream of rows read by a wave: 128-bit resource descriptor in SGPRs (base = start of the tile in the
// batch item), row offset in an SGPR (soffset), the lane's byte offset in ONE VGPR shared by every stream.
// Plain pointers made the compiler keep a per-lane 64-bit address per stream (14 VGPRs, spilled, and every
// scratch reload is a vmcnt wait in the middle of the prefetch).  Offsets are 32-bit: the host checks that
// a batch item spans < 2 GiB before taking a kernel that uses these.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ float buf_load1(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <typename T>
__device__ __forceinline__ void buf_load8(rsrc_t r, unsigned voff, unsigned soff, float (&o)[8]);
template <>
__device__ __forceinline__ void buf_load8<float>(rsrc_t r, unsigned voff, unsigned soff, float (&o)[8]) {
    const v4u a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    const v4u b = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16, soff, 0);
    o[0] = __uint_as_float(a.x); o[1] = __uint_as_float(a.y); o[2] = __uint_as_float(a.z); o[3] = __uint_as_float(a.w);
    o[4] = __uint_as_float(b.x); o[5] = __uint_as_float(b.y); o[6] = __uint_as_float(b.z); o[7] = __uint_as_float(b.w);
}
template <>
__device__ __forceinline__ void buf_load8<bf16_t>(rsrc_t r, unsigned voff, unsigned soff, float (&o)[8]) {
    const v4u a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = __uint_as_float(w[i] << 16);
        o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
// Stores of MORE than 64 bits never carry a register in soffset: the store reads its data registers after it
// has issued, a VALU write of them within two wait states corrupts the data, and LLVM's hazard recogniser pads
// that case only when soffset is not a register (the GCN3 rule "an SGPR offset needs no wait states") -- on
// gfx950 the exemption does not hold (seen in the streaming scan: buffer_store_dwordx4 v[2:5] ... s0 offen
// followed directly by v_mul_f32 v2 corrupted element 0 of some lanes).  The scalar part goes into the VGPR
// offset (one v_add), and the compiler pads.
template <typename T>
__device__ __forceinline__ void buf_store8(rsrc_t r, unsigned voff, unsigned soff, const float (&v)[8]);
template <>
__device__ __forceinline__ void buf_store8<float>(rsrc_t r, unsigned voff, unsigned soff, const float (&v)[8]) {
    __builtin_amdgcn_raw_buffer_store_b128(v4u{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, r, voff + soff, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(v4u{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, r, voff + soff + 16, 0, 0);
}
template <>
__device__ __forceinline__ void buf_store8<bf16_t>(rsrc_t r, unsigned voff, unsigned soff, const float (&v)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        w[i] = (unsigned)from_f32<bf16_t>(v[2 * i]).bits | ((unsigned)from_f32<bf16_t>(v[2 * i + 1]).bits << 16);
    __builtin_amdgcn_raw_buffer_store_b128(v4u{w[0], w[1], w[2], w[3]}, r, voff + soff, 0, 0);
}

// 4-token forms (one 16-B / 8-B access per lane)
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void buf_load4(rsrc_t r, unsigned voff, unsigned soff, float (&o)[4]);
template <>
__device__ __forceinline__ void buf_load4<float>(rsrc_t r, unsigned voff, unsigned soff, float (&o)[4]) {
    const v4u a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    o[0] = __uint_as_float(a.x); o[1] = __uint_as_float(a.y); o[2] = __uint_as_float(a.z); o[3] = __uint_as_float(a.w);
}
template <>
__device__ __forceinline__ void buf_load4<bf16_t>(rsrc_t r, unsigned voff, unsigned soff, float (&o)[4]) {
    const v2u a = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    o[0] = __uint_as_float(a.x << 16); o[1] = __uint_as_float(a.x & 0xffff0000u);
    o[2] = __uint_as_float(a.y << 16); o[3] = __uint_as_float(a.y & 0xffff0000u);
}
template <typename T>
__device__ __forceinline__ void buf_store4(rsrc_t r, unsigned voff, unsigned soff, const float (&v)[4]);
template <>
__device__ __forceinline__ void buf_store4<float>(rsrc_t r, unsigned voff, unsigned soff, const float (&v)[4]) {
    __builtin_amdgcn_raw_buffer_store_b128(v4u{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, r, voff + soff, 0, 0);
}
template <>
__device__ __forceinline__ void buf_store4<bf16_t>(rsrc_t r, unsigned voff, unsigned soff, const float (&v)[4]) {
    const unsigned w0 = (unsigned)from_f32<bf16_t>(v[0]).bits | ((unsigned)from_f32<bf16_t>(v[1]).bits << 16);
    const unsigned w1 = (unsigned)from_f32<bf16_t>(v[2]).bits | ((unsigned)from_f32<bf16_t>(v[3]).bits << 16);
    __builtin_amdgcn_raw_buffer_store_b64(v2u{w0, w1}, r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store1(rsrc_t r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for
// the next channel's prefetch and the previous channel's stores at every exchange.
#define MMU_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * MMU_LOG2E); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * MMU_LN2; }  // v_log_f32
// softplus with the reference kernel's threshold (selective_scan_fwd_kernel.cuh:153-156), on the
// hardware exp2/log2 units: libm expf/log1pf cost ~60 instructions each, which per token was as
// much work as the 16-state recurrence itself.  log1p(e) for tiny e via its series (1 + e would
// round to 1).
__device__ __forceinline__ float softplus_thr(float x) {
    const float e = fast_exp(fminf(x, 20.f));
    float lg = fast_log(1.f + e);
    asm("" : "+v"(lg));  // keep both sides branch-free: an exec-mask branch per token costs more than the log
    const float sp = e < 1e-4f ? e * fmaf(-0.5f, e, 1.f) : lg;
    return x <= 20.f ? sp : x;
}
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + fast_exp(-x)); }

// ---------------------------------------------------------------------------
// wave64 cross-lane primitives (DPP: row_shr 1/2/4/8, row_bcast15, row_bcast31)
// ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float old, float src) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                           ROW_MASK, 0xf, false));
}

#define MMU_DPP_ROW_SHR(n) (0x110 + (n))
#define MMU_DPP_ROW_SHL(n) (0x100 + (n))
#define MMU_DPP_WAVE_SHR1 0x138
#define MMU_DPP_ROW_BCAST15 0x142
#define MMU_DPP_ROW_BCAST31 0x143

// Inclusive scan over the 64 lanes of affine pairs f(h) = P*h + S, composition
// "earlier lane first": (P0,S0) then (P1,S1) -> (P1*P0, P1*S0 + S1)
// (the reference's SSMScanOp, selective_scan_common.h:110-115).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void scan_step_dpp(float &P, float &S) {
    const float Pp = dpp_mov<CTRL, ROW_MASK>(1.0f, P);
    const float Sp = dpp_mov<CTRL, ROW_MASK>(0.0f, S);
    S = fmaf(P, Sp, S);
    P = P * Pp;
}

__device__ __forceinline__ void wave_scan_affine_dpp(float &P, float &S) {
    scan_step_dpp<MMU_DPP_ROW_SHR(1), 0xf>(P, S);
    scan_step_dpp<MMU_DPP_ROW_SHR(2), 0xf>(P, S);
    scan_step_dpp<MMU_DPP_ROW_SHR(4), 0xf>(P, S);
    scan_step_dpp<MMU_DPP_ROW_SHR(8), 0xf>(P, S);
    scan_step_dpp<MMU_DPP_ROW_BCAST15, 0xa>(P, S);
    scan_step_dpp<MMU_DPP_ROW_BCAST31, 0xc>(P, S);
}

__device__ __forceinline__ void wave_scan_affine_shfl(float &P, float &S) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float Pp = __shfl_up(P, off, 64);
        const float Sp = __shfl_up(S, off, 64);
        if (lane >= off) {
            S = fmaf(P, Sp, S);
            P = P * Pp;
        }
    }
}

// Hand-scheduled form of wave_scan_affine_dpp: each scan step is ONE v_fmac_f32_dpp
// (S += dpp(S) * P) and ONE v_mul_f32_dpp (P *= dpp(P)); lanes whose DPP source does not exist
// are not written (bound_ctrl off), rows masked out by row_mask are not written -- which is
// exactly the identity behaviour the scan needs, so no `old` registers and no selects.
// hipcc lowers the intrinsic form to v_mov_b32_dpp + s_nop + separate fmac/mul (7 instructions
// per step).  Two independent scans are interleaved so that every DPP read is >= 2 VALU
// instructions after the write of the register it reads (the VALU->DPP hazard), no s_nop needed
// inside.  Requires EXEC = all 64 lanes.
#define MMU_SCAN2_STEP(ctrl, mask)                                                         \
    "v_fmac_f32_dpp %1, %1, %0 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"               \
    "v_fmac_f32_dpp %3, %3, %2 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"               \
    "v_mul_f32_dpp %0, %0, %0 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                \
    "v_mul_f32_dpp %2, %2, %2 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"

__device__ __forceinline__ void wave_scan_affine_x2(float &P0, float &S0, float &P1, float &S1) {
    asm volatile("s_nop 1\n\t"
                 MMU_SCAN2_STEP("row_shr:1", "0xf")
                 MMU_SCAN2_STEP("row_shr:2", "0xf")
                 MMU_SCAN2_STEP("row_shr:4", "0xf")
                 MMU_SCAN2_STEP("row_shr:8", "0xf")
                 MMU_SCAN2_STEP("row_bcast:15", "0xa")
                 MMU_SCAN2_STEP("row_bcast:31", "0xc")
                 "s_nop 1"
                 : "+v"(P0), "+v"(S0), "+v"(P1), "+v"(S1));
}

// The same inside each 16-lane DPP row only (four steps): prefix (row_shr) and suffix (row_shl) forms.  A row is one
// 128-token chunk at 8 tokens per lane; the chunk carries come from memory (selective_scan_bwd_w8.hip).
__device__ __forceinline__ void row_scan_affine_x2(float &P0, float &S0, float &P1, float &S1) {
    asm volatile("s_nop 1\n\t"
                 MMU_SCAN2_STEP("row_shr:1", "0xf")
                 MMU_SCAN2_STEP("row_shr:2", "0xf")
                 MMU_SCAN2_STEP("row_shr:4", "0xf")
                 MMU_SCAN2_STEP("row_shr:8", "0xf")
                 "s_nop 1"
                 : "+v"(P0), "+v"(S0), "+v"(P1), "+v"(S1));
}
__device__ __forceinline__ void row_rscan_affine_x2(float &P0, float &S0, float &P1, float &S1) {
    asm volatile("s_nop 1\n\t"
                 MMU_SCAN2_STEP("row_shl:1", "0xf")
                 MMU_SCAN2_STEP("row_shl:2", "0xf")
                 MMU_SCAN2_STEP("row_shl:4", "0xf")
                 MMU_SCAN2_STEP("row_shl:8", "0xf")
                 "s_nop 1"
                 : "+v"(P0), "+v"(S0), "+v"(P1), "+v"(S1));
}

#define MMU_SCAN1_STEP(ctrl, mask)                                                         \
    "v_fmac_f32_dpp %1, %1, %0 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"               \
    "v_mul_f32_dpp %0, %0, %0 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                \
    "s_nop 1\n\t"

__device__ __forceinline__ void wave_scan_affine(float &P, float &S) {
    asm volatile("s_nop 1\n\t"
                 MMU_SCAN1_STEP("row_shr:1", "0xf")
                 MMU_SCAN1_STEP("row_shr:2", "0xf")
                 MMU_SCAN1_STEP("row_shr:4", "0xf")
                 MMU_SCAN1_STEP("row_shr:8", "0xf")
                 MMU_SCAN1_STEP("row_bcast:15", "0xa")
                 MMU_SCAN1_STEP("row_bcast:31", "0xc")
                 : "+v"(P), "+v"(S));
}

// value of lane (l-1); lane 0 gets `fill`
__device__ __forceinline__ float wave_shift_up1(float v, float fill) {
    return dpp_mov<MMU_DPP_WAVE_SHR1, 0xf>(fill, v);
}

// lane l <-> lane 63-l
__device__ __forceinline__ float wave_reverse(float v) {
    const int lane = threadIdx.x & 63;
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((63 - lane) << 2, __builtin_bit_cast(int, v)));
}

// inclusive prefix sum over lanes
__device__ __forceinline__ float wave_scan_add(float v) {
    v += dpp_mov<MMU_DPP_ROW_SHR(1), 0xf>(0.f, v);
    v += dpp_mov<MMU_DPP_ROW_SHR(2), 0xf>(0.f, v);
    v += dpp_mov<MMU_DPP_ROW_SHR(4), 0xf>(0.f, v);
    v += dpp_mov<MMU_DPP_ROW_SHR(8), 0xf>(0.f, v);
    v += dpp_mov<MMU_DPP_ROW_BCAST15, 0xa>(0.f, v);
    v += dpp_mov<MMU_DPP_ROW_BCAST31, 0xc>(0.f, v);
    return v;
}

// inclusive prefix / suffix sums within each 16-lane DPP row (4 steps, no cross-row traffic)
__device__ __forceinline__ float row_scan_add_up(float v) {  // lane l gets sum over lanes <= l of its row
    v += dpp_mov<MMU_DPP_ROW_SHR(1), 0xf>(0.f, v);
    v += dpp_mov<MMU_DPP_ROW_SHR(2), 0xf>(0.f, v);
    v += dpp_mov<MMU_DPP_ROW_SHR(4), 0xf>(0.f, v);
    v += dpp_mov<MMU_DPP_ROW_SHR(8), 0xf>(0.f, v);
    return v;
}
__device__ __forceinline__ float row_scan_add_down(float v) {  // lane l gets sum over lanes >= l of its row
    v += dpp_mov<MMU_DPP_ROW_SHL(1), 0xf>(0.f, v);
    v += dpp_mov<MMU_DPP_ROW_SHL(2), 0xf>(0.f, v);
    v += dpp_mov<MMU_DPP_ROW_SHL(4), 0xf>(0.f, v);
    v += dpp_mov<MMU_DPP_ROW_SHL(8), 0xf>(0.f, v);
    return v;
}

__device__ __forceinline__ float wave_bcast_last(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sums FOUR values over the wave at once: on return lanes 12..15 of every 16-lane row hold the wave totals of
// v0..v3 (lane & 3 selects which).  Two quad butterflies fold the four registers into one (each lane ends up
// with the quad sum of value lane & 3), two row_shr steps finish the rows, two ds_bpermute steps join the
// four rows: 17 instructions instead of four 6-step scans + readlanes (9 % of the backward apply kernel).
__device__ __forceinline__ float wave_sum4(float v0, float v1, float v2, float v3) {
    const int lane = threadIdx.x & 63;
    const bool b0 = lane & 1, b1 = lane & 2;
    // quad_perm [1,0,3,2] = 0xB1 (lane ^ 1), [2,3,0,1] = 0x4E (lane ^ 2)
    const float x01 = b0 ? v1 : v0, y01 = b0 ? v0 : v1;
    const float x23 = b0 ? v3 : v2, y23 = b0 ? v2 : v3;
    const float r01 = x01 + dpp_mov<0xB1, 0xf>(0.f, y01);
    const float r23 = x23 + dpp_mov<0xB1, 0xf>(0.f, y23);
    const float x = b1 ? r23 : r01, y = b1 ? r01 : r23;
    float r = x + dpp_mov<0x4E, 0xf>(0.f, y);
    r += dpp_mov<MMU_DPP_ROW_SHR(4), 0xf>(0.f, r);
    r += dpp_mov<MMU_DPP_ROW_SHR(8), 0xf>(0.f, r);
    r += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, __builtin_bit_cast(int, r)));
    r += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, r)));
    return r;
}

// wave_sum4 with the four rows joined by the gfx950 row-swap instructions instead of two ds_bpermute round trips
// (v_permlane16_swap: odd rows of the first operand <-> even rows of the second; v_permlane32_swap: upper half of
// the first <-> lower half of the second; with both operands = r the sum of the two results is r + its partner
// row / half in every lane).  Pure VALU: no LDS latency in the dependency chain.
__device__ __forceinline__ float wave_sum4_swap(float v0, float v1, float v2, float v3) {
    const int lane = threadIdx.x & 63;
    const bool b0 = lane & 1, b1 = lane & 2;
    const float x01 = b0 ? v1 : v0, y01 = b0 ? v0 : v1;
    const float x23 = b0 ? v3 : v2, y23 = b0 ? v2 : v3;
    const float r01 = x01 + dpp_mov<0xB1, 0xf>(0.f, y01);
    const float r23 = x23 + dpp_mov<0xB1, 0xf>(0.f, y23);
    const float x = b1 ? r23 : r01, y = b1 ? r01 : r23;
    float r = x + dpp_mov<0x4E, 0xf>(0.f, y);
    r += dpp_mov<MMU_DPP_ROW_SHR(4), 0xf>(0.f, r);
    r += dpp_mov<MMU_DPP_ROW_SHR(8), 0xf>(0.f, r);
    float a = r, b = r;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    r = a + b;
    a = r;
    b = r;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// sum over the 64 lanes, result valid in every lane
__device__ __forceinline__ float wave_sum(float v) { return wave_bcast_last(wave_scan_add(v)); }
