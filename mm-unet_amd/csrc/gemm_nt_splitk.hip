// gemm_nt_splitk.hip -- C (M x N) = sum over tokens of A[m][t] * B[n][t] on the fp32 matrix cores, split over the
// token axis, for gfx950.
//
// Where it sits: the WEIGHT GRADIENTS of every projection whose reduction runs over the tokens -- in_proj / out_proj /
// x_proj / dt_proj of the Mamba blocks (requirements/mamba/mamba_ssm/ops/selective_scan_interface.py:272-277,394;
// mamba_simple.py:201-205,270) and MMConv's K x 1 DSC convolution (src/UM_Net/MMUNet.py:262): dW = G . X^T with
// T = batch * L up to 524,288 tokens and M * N <= 49,152.  As one library GEMM these run at 0.5-1 TB/s of operands
// (no split-K in hipBLASLt for them); the framework's earlier form -- slabs of the token axis as a batched GEMM plus an
// ordered sum -- reaches 3.5-4.5 TB/s and needs both operands as 2-D (rows, tokens) matrices, i.e. a transposing copy
// whenever one of them is batch-major (profiles/r02_library_gemm_wgrad_shapes.txt).
//
// Both operands are contiguous along the contraction ("NT"): no transposes anywhere.  A token t = b * L + l of operand
// X lives at X + row * x_rs + b * x_bs + l, so channel-major [C][B][L] and batch-major [B][C][L] storage are both
// addressed in place.
//   * workgroup = 256 threads = 4 waves, output tile 128 x 64: wave w owns rows 32w..32w+31 and both 32-column tiles
//     (2 x 16 accumulator VGPRs), v_mfma_f32_32x32x2_f32 -- float32 in, float32 accumulate: exact products, no split;
//   * a slab of the token axis per workgroup, walked in chunks of 32 tokens: 16-byte global loads (a row's 32 tokens
//     are 128 contiguous bytes) one chunk ahead -> LDS [row][36] (the pad makes the 16-byte operand reads of 32 rows
//     conflict-free), double-buffered, one LDS-only barrier per chunk;
//   * operand reads: lane (row r, half h) takes tokens 8q + 4h .. + 3 of its row with ONE ds_read_b128 and feeds four
//     MFMAs (the k-index of an MFMA step is arbitrary as long as A and B agree);
//   * rows beyond M / N read a clamped row (outputs are independent: what they compute is never stored);
//   * partials[slab][M][N] -> reduce kernel adds the slabs in a fixed order: deterministic, no atomics.
// HBM-bound by design: 24 KB of operands per 32 MFMAs (2,048 cycles) per workgroup.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ unsigned nt_pack_bf16(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
// x ~= hi + lo with both in bf16: hi = RNE(x), lo = RNE(x - hi)
__device__ __forceinline__ void nt_split2(float a, float b, unsigned &hi, unsigned &lo) {
    hi = nt_pack_bf16(a, b);
    const float ah = __builtin_bit_cast(float, hi << 16), bh = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = nt_pack_bf16(a - ah, b - bh);
}

constexpr int NT_TM = 128, NT_TN = 64, NT_TK = 32, NT_LD = 36;   // LDS row stride in floats
constexpr int NT_A_FLOATS = NT_TM * NT_LD, NT_B_FLOATS = NT_TN * NT_LD;
constexpr int NT_STAGE = NT_A_FLOATS + NT_B_FLOATS;             // 6,912 floats = 27 KB per buffer

struct NtArgs {
    const void *a, *b;      // float32 (bfloat16 for gemm_nt_wide_kernel<true>)
    float *part;            // [slabs][M][N]
    long a_rs, a_bs, b_rs, b_bs;
    int M, N, L, batch;     // tokens = batch * L, L % 32 == 0
    int slab_chunks;        // chunks of 32 tokens per slab
    int n_chunks;           // total chunks
};

// SPLIT = false: float32 operands in LDS, v_mfma_f32_32x32x2_f32 (exact products; 2,048 MFMA cycles per chunk and wave --
//   the kernel is then bound by the fp32 matrix pipe at ~4.3 TB/s of operands).
// SPLIT = true: each operand element is split into bf16 hi + lo when it is staged (LDS row = [32 hi | 32 lo | pad], the
//   same 144 bytes) and a product costs three v_mfma_f32_32x32x16_bf16 (lo*hi + hi*lo + hi*hi, the 2^-16-relative
//   lo*lo term dropped, as in gemm_tokens_mfma.hip): 384 MFMA cycles per chunk and wave -- HBM-bound.
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_nt_splitk_kernel(NtArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * NT_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * NT_TM, n0 = blockIdx.y * NT_TN, slab = blockIdx.z;
    const int c_lo = slab * p.slab_chunks;
    const int c_hi = min(c_lo + p.slab_chunks, p.n_chunks);

    // ---- staging: thread -> (row, 16-byte column group) of the A tile (4 per thread) and of the B tile (2 per thread)
    const int colg = tid & 7;                 // 8 groups of 4 tokens
    const int row_a = tid >> 3;               // 0..31, + 32 j
    const float *ap[4];
    const float *bp[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = min(m0 + row_a + 32 * j, p.M - 1);   // (never dereferenced beyond M)
        ap[j] = (const float *)p.a + (long)r * p.a_rs + 4 * colg;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = min(n0 + row_a + 32 * j, p.N - 1);
        bp[j] = (const float *)p.b + (long)r * p.b_rs + 4 * colg;
    }
    const int cpb = p.L / NT_TK;              // chunks per batch item
    // rows beyond M / N are never loaded: their LDS rows stay zero-fed from registers
    bool va[4], vb[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) va[j] = m0 + row_a + 32 * j < p.M;
#pragma unroll
    for (int j = 0; j < 2; ++j) vb[j] = n0 + row_a + 32 * j < p.N;
    float4 ra[2][4], rb[2][2];                // two chunks in flight in registers (2 waves per SIMD: 256 VGPRs each)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[s][j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 2; ++j) rb[s][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto load = [&](int c, float4 (&xa)[4], float4 (&xb)[2]) {
        const int bi = c / cpb, l0 = (c - bi * cpb) * NT_TK;
        const long oa = (long)bi * p.a_bs + l0, ob = (long)bi * p.b_bs + l0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (va[j]) xa[j] = *reinterpret_cast<const float4 *>(ap[j] + oa);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (vb[j]) xb[j] = *reinterpret_cast<const float4 *>(bp[j] + ob);
    };
    auto put = [&](float *row, const float4 &v) {
        if constexpr (SPLIT) {
            uint2 hi, lo;
            nt_split2(v.x, v.y, hi.x, lo.x);
            nt_split2(v.z, v.w, hi.y, lo.y);
            *reinterpret_cast<uint2 *>(row + 2 * colg) = hi;
            *reinterpret_cast<uint2 *>(row + 16 + 2 * colg) = lo;
        } else {
            *reinterpret_cast<float4 *>(row + 4 * colg) = v;
        }
    };
    auto stage = [&](float *buf, const float4 (&xa)[4], const float4 (&xb)[2]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) put(buf + (row_a + 32 * j) * NT_LD, xa[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) put(buf + NT_A_FLOATS + (row_a + 32 * j) * NT_LD, xb[j]);
    };

    f32x16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;

    // operand reads: lane (r = lane % 32, h = lane / 32) -> tokens 8 q + 4 h .. + 3 of its row
    const int a_off = (32 * w + (lane & 31)) * NT_LD + 4 * (lane >> 5);
    const int b_off = NT_A_FLOATS + (lane & 31) * NT_LD + 4 * (lane >> 5);
    auto products = [&](const float *cur) {
        if constexpr (SPLIT) {
            // lane (r, h) holds tokens 16 s + 8 h .. + 7 of its row: 16 bytes of the hi half, 16 of the lo half
#pragma unroll
            for (int s = 0; s < NT_TK / 16; ++s) {
                const float *ar = cur + a_off + 8 * s, *br = cur + b_off + 8 * s;
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(ar), al = *reinterpret_cast<const bf16x8 *>(ar + 16);
                const bf16x8 b0h = *reinterpret_cast<const bf16x8 *>(br), b0l = *reinterpret_cast<const bf16x8 *>(br + 16);
                const bf16x8 b1h = *reinterpret_cast<const bf16x8 *>(br + 32 * NT_LD);
                const bf16x8 b1l = *reinterpret_cast<const bf16x8 *>(br + 32 * NT_LD + 16);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0h, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b1h, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0l, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1l, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0h, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1h, acc[1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < NT_TK / 8; ++q) {
                const float4 av = *reinterpret_cast<const float4 *>(cur + a_off + 8 * q);
                const float4 b0 = *reinterpret_cast<const float4 *>(cur + b_off + 8 * q);
                const float4 b1 = *reinterpret_cast<const float4 *>(cur + b_off + 32 * NT_LD + 8 * q);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc[1], 0, 0, 0);
            }
        }
    };
    // chunk c is consumed from LDS buffer (c - c_lo) & 1; while it is, chunk c + 1 goes registers -> the other buffer
    // and chunk c + 3 is requested into the registers that just emptied (chunk c + 2's request is one step old):
    // a request has two chunks of MFMA work (~1.7 us) to land.  Register set of chunk c: (c - c_lo) & 1.
    auto step = [&](int c, const float *cur, float *nxt, float4 (&xa)[4], float4 (&xb)[2]) {
        if (c + 1 < c_hi) {
            stage(nxt, xa, xb);
            if (c + 3 < c_hi) load(c + 3, xa, xb);
        }
        products(cur);
        MMU_LDS_BARRIER();
    };
    if (c_lo < c_hi) {
        load(c_lo, ra[0], rb[0]);
        stage(lds, ra[0], rb[0]);
        if (c_lo + 1 < c_hi) load(c_lo + 1, ra[1], rb[1]);
        if (c_lo + 2 < c_hi) load(c_lo + 2, ra[0], rb[0]);
        MMU_LDS_BARRIER();
        int c = c_lo;
        for (; c + 1 < c_hi; c += 2) {
            step(c, lds, lds + NT_STAGE, ra[1], rb[1]);
            step(c + 1, lds + NT_STAGE, lds, ra[0], rb[0]);
        }
        if (c < c_hi) step(c, lds, lds + NT_STAGE, ra[1], rb[1]);
    }
    // ---- this slab's partial tile: D[i][j], j = lane % 32, i = 8 (e / 4) + 4 (lane / 32) + e % 4
    float *op = p.part + ((long)slab * p.M) * p.N;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = n0 + 32 * n + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + 32 * w + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
            if (row < p.M && col < p.N) op[(long)row * p.N + col] = acc[n][e];
        }
    }
}

// ---- the wide form: 128 tokens per step, one workgroup of 8 waves per CU ---------------------------------------------
// Rows of an operand are typically a power of two apart (B * L tokens): the 192 row segments a workgroup asks for per
// step then lie in ONE HBM channel / bank, each in a different DRAM row.  With 128-byte segments (32 tokens) the kernel
// above reaches 3.5-3.7 TB/s on such operands against 5.0-5.6 TB/s when the rows are padded apart
// (tools/dbg/nt_layout_probe.py).  Here a wave-wide load covers 512 contiguous bytes of each of two rows, a quarter of
// the row activations per byte.  192 rows x 128 tokens as hi | lo bf16 = 99 KB of LDS, single-buffered: the next step's
// 96 KB travel in registers (12 x 16 bytes per thread) while the matrix cores work through the current one -- at 768
// MFMA cycles per wave and step the kernel does not need LDS double buffering to stay HBM-bound.
constexpr int NW_TK = 128, NW_LD = 132;                 // LDS row: [128 hi bf16 | 128 lo bf16 | 16 B pad] = 132 dwords
constexpr int NW_ROWS = NT_TM + NT_TN;                  // 192
constexpr int NW_LDS_BYTES = NW_ROWS * NW_LD * 4;       // 101,376
constexpr int NW_LOADS = NW_ROWS / 16;                  // 12 row passes of 16 rows (512 threads = 16 rows x 32 pieces)

// XB: bfloat16 operands (autocast) -- a row's 128 tokens are 256 bytes, a thread's piece 8 bytes written straight into
// the hi image; both operands are exact bf16 values: ONE MFMA per product.
template <bool XB>
__global__ __launch_bounds__(512, 1) void gemm_nt_wide_kernel(NtArgs p) {
    using el_t = typename std::conditional<XB, unsigned short, float>::type;
    using ld_t = typename std::conditional<XB, uint2, float4>::type;
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * NT_TM, n0 = blockIdx.y * NT_TN, slab = blockIdx.z;
    const int c_lo = slab * p.slab_chunks;                       // in steps of 128 tokens
    const int c_hi = min(c_lo + p.slab_chunks, p.n_chunks);
    const int piece = tid & 31, row_p = tid >> 5;                // 16-byte piece of the 512-byte row segment; row of the pass
    const el_t *src[NW_LOADS];
    bool valid[NW_LOADS];
#pragma unroll
    for (int j = 0; j < NW_LOADS; ++j) {
        const int r = row_p + 16 * j;                            // tile row: 0..127 = A, 128..191 = B
        if (r < NT_TM) {
            valid[j] = m0 + r < p.M;
            src[j] = (const el_t *)p.a + (long)min(m0 + r, p.M - 1) * p.a_rs + 4 * piece;
        } else {
            valid[j] = n0 + r - NT_TM < p.N;
            src[j] = (const el_t *)p.b + (long)min(n0 + r - NT_TM, p.N - 1) * p.b_rs + 4 * piece;
        }
    }
    const int cpb = p.L / NW_TK;
    ld_t regs[NW_LOADS];
#pragma unroll
    for (int j = 0; j < NW_LOADS; ++j) regs[j] = ld_t{};
    auto load = [&](int c) {
        const int bi = c / cpb, l0 = (c - bi * cpb) * NW_TK;
        const long oa = (long)bi * p.a_bs + l0, ob = (long)bi * p.b_bs + l0;
#pragma unroll
        for (int j = 0; j < NW_LOADS; ++j)
            if (valid[j]) regs[j] = *reinterpret_cast<const ld_t *>(src[j] + (j < NT_TM / 16 ? oa : ob));
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < NW_LOADS; ++j) {
            float *row = wlds + (row_p + 16 * j) * NW_LD;
            if constexpr (XB) {
                *reinterpret_cast<uint2 *>(row + 2 * piece) = regs[j];
            } else {
                uint2 hi, lo;
                nt_split2(regs[j].x, regs[j].y, hi.x, lo.x);
                nt_split2(regs[j].z, regs[j].w, hi.y, lo.y);
                *reinterpret_cast<uint2 *>(row + 2 * piece) = hi;
                *reinterpret_cast<uint2 *>(row + 64 + 2 * piece) = lo;
            }
        }
    };
    // wave w: rows 32 (w & 3) .. + 31 of A against columns 32 (w >> 2) .. + 31 of B
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int a_off = (32 * (w & 3) + (lane & 31)) * NW_LD + 4 * (lane >> 5);
    const int b_off = (NT_TM + 32 * (w >> 2) + (lane & 31)) * NW_LD + 4 * (lane >> 5);
    if (c_lo < c_hi) {
        load(c_lo);
        for (int c = c_lo; c < c_hi; ++c) {
            stage();                                   // (waits for the loads of step c)
            if (c + 1 < c_hi) load(c + 1);
            MMU_LDS_BARRIER();
#pragma unroll
            for (int s = 0; s < NW_TK / 16; ++s) {     // lane (r, h): tokens 16 s + 8 h .. + 7 of its row
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(wlds + a_off + 8 * s);
                const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(wlds + b_off + 8 * s);
                if constexpr (!XB) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8 *>(wlds + a_off + 64 + 8 * s);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(wlds + b_off + 64 + 8 * s);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
            MMU_LDS_BARRIER();                         // everyone is done reading before the next step is staged
        }
    }
    float *op = p.part + ((long)slab * p.M) * p.N;
    const int col = n0 + 32 * (w >> 2) + (lane & 31);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + 32 * (w & 3) + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
        if (row < p.M && col < p.N) op[(long)row * p.N + col] = acc[e];
    }
}

// C[i] = sum over slabs of part[s][i] in a fixed order: 16 slab groups per output in parallel (each a strided serial
// sum), then the 16 group sums in order.  transpose_out: part is (N x M) row-major and C is (M x N).
__global__ __launch_bounds__(1024) void gemm_nt_reduce_kernel(const float *__restrict__ part, float *__restrict__ c,
                                                              long n, int slabs, int rows, int cols, int transpose_out) {
    __shared__ float sums[16][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + o;
    float s = 0.f;
    if (i < n) {
        int k = g;
        for (; k + 48 < slabs; k += 64) {
            const float v0 = part[(long)k * n + i], v1 = part[(long)(k + 16) * n + i];
            const float v2 = part[(long)(k + 32) * n + i], v3 = part[(long)(k + 48) * n + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < slabs; k += 16) s += part[(long)k * n + i];
    }
    sums[g][o] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = sums[0][o];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += sums[k][o];
        if (transpose_out) {
            const long r = i / cols, q = i - r * cols;      // part element (r, q) of the (rows x cols) swapped product
            c[q * rows + r] = t;
        } else {
            c[i] = t;
        }
    }
}

// slabs: all workgroups co-resident (2 per CU), at least 8 chunks (256 tokens) per slab
// the wide form: 128-token steps, one workgroup per CU, at least 2 steps per slab
bool nt_wide(int seqlen, int exact) {
    static const int on = []() { const char *e = getenv("MMU_GEMM_NT_WIDE"); return e ? atoi(e) : 1; }();
    return on && !exact && seqlen % NW_TK == 0;
}
int nt_plan(int M, int N, long tokens, int &slab_chunks, bool wide = false) {
    const long n_chunks = tokens / (wide ? NW_TK : NT_TK);
    const long tiles = (long)((M + NT_TM - 1) / NT_TM) * ((N + NT_TN - 1) / NT_TN);
    long want = ((wide ? 1L : 2L) * mmu_cu_count()) / tiles;
    if (want < 1) want = 1;
    long sc = (n_chunks + want - 1) / want;
    if (sc < (wide ? 2 : 8)) sc = wide ? 2 : 8;
    // An ODD number of chunks per slab.  Rows of an operand are typically a power of two apart (B * L tokens), so the
    // 192 row segments a workgroup requests per chunk fall into one HBM channel; with power-of-two slabs every
    // workgroup would also START in the same channel and walk the channels in lockstep (measured: 3.5 instead of
    // 5.3 TB/s).  Odd slabs spread the workgroups over all 128-byte positions.
    static const int odd = []() { const char *e = getenv("MMU_GEMM_NT_ODD"); return e ? atoi(e) : 1; }();
    if (odd) sc |= 1;
    slab_chunks = (int)sc;
    return (int)((n_chunks + sc - 1) / sc);
}

// rows of operands a workgroup column/row of tiles loads per chunk, summed over the grid: the orientation that loads
// fewer wins (the 128-row side should hold the larger operand)
long nt_rows_loaded(int M, int N) {
    return (long)((M + NT_TM - 1) / NT_TM) * ((N + NT_TN - 1) / NT_TN) * (NT_TM + NT_TN);
}

}  // namespace

extern "C" size_t mmu_gemm_nt_splitk_workspace_floats(int m, int n, int batch, int seqlen) {
    if (m <= 0 || n <= 0 || batch <= 0 || seqlen <= 0) return 0;
    int sc = 0;
    const bool swap = nt_rows_loaded(n, m) < nt_rows_loaded(m, n);
    const int M = swap ? n : m, N = swap ? m : n;
    int slabs = nt_plan(M, N, (long)batch * seqlen, sc, false);      // enough for either form
    if (seqlen % NW_TK == 0) slabs = max(slabs, nt_plan(M, N, (long)batch * seqlen, sc, true));
    return (size_t)slabs * m * n;
}

extern "C" int mmu_gemm_nt_splitk(const mmu_gemm_nt_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "gemm_nt_splitk: null params");
    MMU_CHECK(p->m > 0 && p->n > 0 && p->batch > 0 && p->seqlen > 0, "gemm_nt_splitk: empty problem");
    MMU_CHECK(p->seqlen % NT_TK == 0, "gemm_nt_splitk: seqlen must be a multiple of %d (got %d)", NT_TK, p->seqlen);
    MMU_CHECK(p->a && p->b && p->c && p->workspace, "gemm_nt_splitk: a, b, c, workspace are required");
    MMU_CHECK(p->ab_dtype == MMU_DTYPE_F32 || p->ab_dtype == MMU_DTYPE_BF16, "gemm_nt_splitk: ab_dtype must be float32 or bfloat16");
    const bool xb = p->ab_dtype == MMU_DTYPE_BF16;
    const uintptr_t amask = xb ? 7 : 15;
    MMU_CHECK(((uintptr_t)p->a & amask) == 0 && ((uintptr_t)p->b & amask) == 0 && p->a_rs % 4 == 0 && p->a_bs % 4 == 0 &&
                  p->b_rs % 4 == 0 && p->b_bs % 4 == 0,
              "gemm_nt_splitk: operands must be %d-byte aligned with row / batch strides that are multiples of 4", (int)amask + 1);
    MMU_CHECK(!xb || (p->seqlen % NW_TK == 0 && !p->exact_products && !p->narrow_steps),
              "gemm_nt_splitk: bfloat16 operands need seqlen %% %d == 0 (128-token steps) and no exact / narrow request", NW_TK);
    hipStream_t st = (hipStream_t)stream;
    const bool swap = nt_rows_loaded(p->n, p->m) < nt_rows_loaded(p->m, p->n);   // compute C^T, transpose in the reduce
    NtArgs a;
    a.part = p->workspace;
    if (swap) {
        a.a = p->b; a.b = p->a; a.a_rs = p->b_rs; a.a_bs = p->b_bs; a.b_rs = p->a_rs; a.b_bs = p->a_bs;
        a.M = p->n; a.N = p->m;
    } else {
        a.a = p->a; a.b = p->b; a.a_rs = p->a_rs; a.a_bs = p->a_bs; a.b_rs = p->b_rs; a.b_bs = p->b_bs;
        a.M = p->m; a.N = p->n;
    }
    a.L = p->seqlen; a.batch = p->batch;
    const long tokens = (long)p->batch * p->seqlen;
    static const bool exact_env = []() { const char *e = getenv("MMU_GEMM_NT_EXACT"); return e && e[0] == '1'; }();
    const bool exact = exact_env || p->exact_products;
    const bool wide = xb || (!p->narrow_steps && nt_wide(p->seqlen, exact));
    a.n_chunks = (int)(tokens / (wide ? NW_TK : NT_TK));
    const int slabs = nt_plan(a.M, a.N, tokens, a.slab_chunks, wide);
    dim3 grid((a.M + NT_TM - 1) / NT_TM, (a.N + NT_TN - 1) / NT_TN, slabs);
    if (wide) {
        static unsigned long long attr_mask = 0, attr_mask_xb = 0;  // per device
        if (hipError_t e = xb ? mmu_set_lds_once(gemm_nt_wide_kernel<true>, NW_LDS_BYTES, attr_mask_xb)
                              : mmu_set_lds_once(gemm_nt_wide_kernel<false>, NW_LDS_BYTES, attr_mask);
            e != hipSuccess)
            return mmu_fail("gemm_nt_splitk: LDS attribute: %s", hipGetErrorString(e));
        if (xb)
            gemm_nt_wide_kernel<true><<<grid, 512, NW_LDS_BYTES, st>>>(a);
        else
            gemm_nt_wide_kernel<false><<<grid, 512, NW_LDS_BYTES, st>>>(a);
    } else if (exact) {
        gemm_nt_splitk_kernel<false><<<grid, 256, 0, st>>>(a);
    } else {
        gemm_nt_splitk_kernel<true><<<grid, 256, 0, st>>>(a);
    }
    MMU_HIP_LAUNCH_CHECK("gemm_nt_splitk");
    const long n = (long)p->m * p->n;
    const long job[8] = {0, (long)p->workspace, (long)p->c, 0, n, slabs, ((long)a.M << 32) | (long)a.N, swap ? 1 : 0};
    if (mmu_defer_job(job)) return 0;   // (deferred_reduce.hip: the slab sums of a whole backward pass in one launch)
    gemm_nt_reduce_kernel<<<(unsigned)((n + 63) / 64), 1024, 0, st>>>(p->workspace, p->c, n, slabs, a.M, a.N, swap ? 1 : 0);
    MMU_HIP_LAUNCH_CHECK("gemm_nt_splitk(reduce)");
    return 0;
}
