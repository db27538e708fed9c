// valu_rates.hip -- VALU issue-rate microbenchmark for gfx950 (MI355X): cycles per wave-instruction on one SIMD
// for plain / packed / transcendental / DPP ops at 1, 2, 4 waves per SIMD.  Standalone: hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const float m = 0.999f, c = 1e-4f;
    const v2f m2 = {m, m}, c2 = {c, c};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {  // 16 independent-ish plain fma (8 chains x 2)
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n") ""
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            asm volatile(REP8("v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n") ""
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if constexpr (KIND == 1) {  // 32 packed fma, 8 chains
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n") ""
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
        } else if constexpr (KIND == 2) {  // 32 exp, 4 chains
            asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n") ""
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if constexpr (KIND == 3) {  // 32 dpp mul
            asm volatile(REP8("v_mul_f32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mul_f32_dpp %2, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n") ""
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
        } else if constexpr (KIND == 4) {  // scan-like mix: per 8: 2 exp + 5 pk + 1 plain   (x4 = 32 instr)
            asm volatile("s_nop 0\n" REP8("v_exp_f32 %4, %5\n v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_exp_f32 %6, %7\n") ""
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m2), "v"(c2));
        } else if constexpr (KIND == 5) {  // 32 dependent packed fma (ONE chain)
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %0, %0, %8, %9\n") ""
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
        } else if constexpr (KIND == 6) {  // 32 dependent plain fma (ONE chain)
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n") ""
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if constexpr (KIND == 7) {  // exp feeding a packed fma (dependent): 8 x (2 exp + 1 pk using them + 1 indep pk) = 32 instr
            asm volatile("s_nop 0\n" REP8("v_exp_f32 %4, %6\n v_exp_f32 %5, %7\n v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n") ""
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m2), "v"(c2));
        } else if constexpr (KIND == 8) {  // 16 plain + 16 ds_read_b32 interleaved (LDS issue beside VALU)
            __shared__ float sm[1024];
            float r;
            asm volatile(REP8("v_fma_f32 %0, %0, %9, %10\n ds_read_b32 %8, %11\n v_fma_f32 %1, %1, %9, %10\n ds_read_b32 %8, %11\n") "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(r) : "v"(m), "v"(c), "v"((unsigned)((threadIdx.x & 255) * 4)));
            a7 += r * 0.f + sm[0] * 0.f;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char *name, int ninstr) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4 * 2); hipMalloc(&cyc, 256 * 16 * 8 * 2);
    const int iters = 2000;
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps;  // 4*wps waves per CU = wps per SIMD
        k<KIND><<<256, threads>>>(out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<KIND><<<256, threads>>>(out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * 4 * wps);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; unsigned long long mx = 0;
        for (auto v : h) { s += v; mx = v > mx ? v : mx; }
        const double avg = s / h.size();
        // s_memtime ticks at 100 MHz? report both raw ticks and wall-derived cycles
        printf("%-28s waves/SIMD %d: memtime ticks/instr/wave %.3f  -> per SIMD-instr %.3f | wall %.1f us -> %.2f ns per SIMD-instr\n", name, wps,
               avg / (iters * (double)ninstr), avg / (iters * (double)ninstr) / wps, ms * 1e3, ms * 1e6 / (iters * (double)ninstr * wps));
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32 x8 chains", 32);
    run<1>("v_pk_fma_f32 x8 chains", 32);
    run<2>("v_exp_f32 x4 chains", 32);
    run<3>("v_mul_f32_dpp x4 chains", 32);
    run<4>("mix 2exp+2pk (x8)", 32);
    run<5>("v_pk_fma_f32 dependent", 32);
    run<6>("v_fma_f32 dependent", 32);
    run<7>("2exp -> pk dependent", 32);
    run<8>("v_fma + ds_read_b32", 32);
    return 0;
}
