// probe: B operand of v_mfma_f32_32x32x16_bf16 built from a [k][n] LDS image with ds_read_b64_tr_b16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe(const unsigned short *Bin /*[16][32]*/, float *C /*[32][32]*/) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16 * 32];
    const int l = threadIdx.x;
    for (int i = l; i < 16 * 32; i += 64) lds[i] = Bin[i];
    __syncthreads();
    const int h = l >> 5, grp = (l >> 4) & 1, li = l & 15, q = li >> 2, p = li & 3;
    // lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of the group's 4 x 16 block
    const int col0 = 16 * grp;
    auto addr = [&](int krow0) {
        return (__attribute__((address_space(3))) s4 *)(lds + (krow0 + q) * 32 + col0 + 4 * p);
    };
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(addr(8 * h));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(addr(8 * h + 4));
    typedef __attribute__((ext_vector_type(8))) short s8;
    s8 bs = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    bf16x8 b = __builtin_bit_cast(bf16x8, bs);
    // A[m][k] = (m == k): lane (r = l & 31, h) holds k = 8h + j
    s8 as;
    for (int j = 0; j < 8; ++j) as[j] = ((l & 31) == 8 * h + j) ? (short)0x3f80 : (short)0;   // bf16 1.0
    bf16x8 a = __builtin_bit_cast(bf16x8, as);
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), col = l & 31;
        C[row * 32 + col] = acc[e];
    }
}
int main() {
    std::vector<unsigned short> B(16 * 32);
    std::vector<float> Bf(16 * 32);
    for (int k = 0; k < 16; ++k)
        for (int n = 0; n < 32; ++n) {
            float v = (float)((k * 7 + n * 3) % 200);
            Bf[k * 32 + n] = v;
            unsigned u;
            memcpy(&u, &v, 4);
            B[k * 32 + n] = (unsigned short)(u >> 16);
        }
    unsigned short *dB;
    float *dC;
    hipMalloc(&dB, B.size() * 2);
    hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dB, dC);
    std::vector<float> C(32 * 32);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            const float want = m < 16 ? Bf[m * 32 + n] : 0.f;
            if (C[m * 32 + n] != want) {
                if (bad < 8) printf("mismatch C[%d][%d] = %g want %g\n", m, n, C[m * 32 + n], want);
                ++bad;
            }
        }
    printf("tr probe: %d mismatches\n", bad);
    return bad != 0;
}
