// gemm_tokens_mfma.hip -- out[b] = W (M x K) . X[b] (K x T tokens-last) on the bf16 matrix cores with float32
// accuracy (the hi/lo split of conv3x3_mfma.hip: x*w ~= xh*wh + xh*wl + xl*wh, three MFMAs per product).
//
// Where it sits: MMConv's K x 1 DSC convolution on the tokens-last sampler output (src/UM_Net/MMUNet.py:262; 39.5 %
// of the model's conv FLOPs) -- forward W2 (Cout x 3 Cin) . samples, input gradient W2^T . G -- which hipBLASLt runs
// as skinny fp32 GEMMs at ~36-90 TFLOP/s (the f32-input MFMA peaks at 157).
//
// Workgroup = 512 threads, a 64-row x 512-token output tile; K in chunks of 16.  The eight waves are SPECIALISED:
//   waves 4-7 (producers): dword loads of the X chunk coalesced along tokens (8 rows per lane) -> hi/lo split in
//       registers -> one 16-byte LDS write per lane into [k half][token][8 k] images (conflict-free both ways),
//       plus the chunk's prepared weights; their loads run one chunk ahead of their LDS writes;
//   waves 0-3 (consumers): 128 tokens each, 2 x 4 MFMA tiles (128 accumulators), 12 ds_read_b128 + 24 MFMAs per
//       chunk, nothing else.
// One wave of each kind per SIMD: the matrix pipe and the VALU / VMEM / LDS-write work co-issue (a single wave
// doing both serialises them -- the MFMA:staging ratio here is 1:1, not 9:1 as in the 3x3 conv).  LDS is
// double-buffered (2 x 36 KB), one LDS-only barrier per chunk, persistent workgroups walk the tiles so the
// producers' stream runs straight across tile boundaries.
//
// Measured (MI355X, 3.2 GFLOP each): the large-token DSC shapes are HBM-bound, not MFMA-bound -- 64 x 192 x 131,072
// tokens moves 133 MB: 27 us = 5.0 TB/s here, 40-46 us in hipBLASLt; its input gradient (192 x 64) 32 us vs 47 us.
// Deep-K problems with few tokens (512 x 1536 x 2,048: 64 tiles x 96 chunks) leave most CUs idle and would need
// split-K: the Python side keeps those on hipBLASLt (mfma_gemm.MIN_TILES).  Ablation at that shape: consumers
// alone 0.52 us per chunk (768 cycles of MFMA + LDS latency + barrier), producers alone 0.81 us, together 1.16 us:
// the two kinds of wave still contend for VALU issue and the LDS port.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TT = 512, CK = 16;
constexpr int XIMG = 2 * TT * 16;          // one bf16 image of the X chunk: [half][token][8 k] = 16,384 B
constexpr int WIMG = 2 * 64 * 16;          // one bf16 image of the W chunk: [half][64 rows][8 k] = 2,048 B
constexpr int STAGE = 2 * XIMG + 2 * WIMG;  // 36,864 B
constexpr int LDS_BYTES = 2 * STAGE;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo) {
    hi = pack_bf16(a, b);
    const float ah = __builtin_bit_cast(float, hi << 16), bh = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pack_bf16(a - ah, b - bh);
}

// W [M][K] with leading dimension ldw (trans = 0) or W [K][M] read transposed (trans = 1)
//   -> [M/64][K/16][hi|lo][half][64 rows][8 k] bf16
// M, K: the weight's own dimensions; the image covers them rounded up to 64 rows / 16 columns, zero beyond.
__global__ __launch_bounds__(256) void gemm_tokens_prep_kernel(const float *__restrict__ w, long ldw,
                                                               unsigned short *__restrict__ out, int M, int K, int trans) {
    const int Mp = (M + 63) & ~63, Kp = (K + CK - 1) & ~(CK - 1);
    const long n = (long)Mp * Kp;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int k8 = (int)(idx & 7), row = (int)((idx >> 3) & 63), half = (int)((idx >> 9) & 1);
    long r = idx >> 10;
    const int nch = Kp / CK;
    const int ch = (int)(r % nch), mt = (int)(r / nch);
    const int m = mt * 64 + row, k = ch * CK + half * 8 + k8;
    const float v = (m < M && k < K) ? (trans ? w[(long)k * ldw + m] : w[(long)m * ldw + k]) : 0.f;
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 l = (__bf16)r1;
    const long base = ((long)(mt * nch + ch) * 2) * (2 * 64 * 8) + ((long)half * 64 + row) * 8 + k8;
    out[base] = __builtin_bit_cast(unsigned short, h);
    out[base + 2 * 64 * 8] = __builtin_bit_cast(unsigned short, l);
    // third part (bits 17-24 of the mantissa), behind the whole two-part image: read by the 32-token kernel only
    out[2 * n + (long)(mt * nch + ch) * (2 * 64 * 8) + ((long)half * 64 + row) * 8 + k8] =
        __builtin_bit_cast(unsigned short, (__bf16)(r1 - (float)l));
}

// The same images for MANY weights in one launch (mmu_gemm_tokens_prepare_batch): blockIdx.y picks a row of the device
// table {weight, leading dimension, image, rows, inner, transposed} (6 x int64).  MM_Net prepares the DSC weights of all
// its MMConv blocks -- both orientations -- once per forward pass instead of one 4.8 us launch in front of each product.
__global__ __launch_bounds__(256) void gemm_tokens_prep_batch_kernel(const long *__restrict__ table) {
    const long *row = table + 6 * (long)blockIdx.y;
    const float *w = reinterpret_cast<const float *>(row[0]);
    const long ldw = row[1];
    unsigned short *out = reinterpret_cast<unsigned short *>(row[2]);
    const int M = (int)row[3], K = (int)row[4], trans = (int)row[5];
    const int Mp = (M + 63) & ~63, Kp = (K + CK - 1) & ~(CK - 1);
    const long n = (long)Mp * Kp;
    const int nch = Kp / CK;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int k8 = (int)(idx & 7), rw = (int)((idx >> 3) & 63), half = (int)((idx >> 9) & 1);
        const long r = idx >> 10;
        const int ch = (int)(r % nch), mt = (int)(r / nch);
        const int m = mt * 64 + rw, k = ch * CK + half * 8 + k8;
        const float v = (m < M && k < K) ? (trans ? w[(long)k * ldw + m] : w[(long)m * ldw + k]) : 0.f;
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 l = (__bf16)r1;
        const long base = ((long)(mt * nch + ch) * 2) * (2 * 64 * 8) + ((long)half * 64 + rw) * 8 + k8;
        out[base] = __builtin_bit_cast(unsigned short, h);
        out[base + 2 * 64 * 8] = __builtin_bit_cast(unsigned short, l);
        out[2 * n + (long)(mt * nch + ch) * (2 * 64 * 8) + ((long)half * 64 + rw) * 8 + k8] =
            __builtin_bit_cast(unsigned short, (__bf16)(r1 - (float)l));
    }
}

struct GemmArgs {
    const void *x;            // float, or bf16_t for the XB instantiation of the 512-token kernel
    const unsigned short *wp;
    void *out;                // same type as x
    long x_rs, x_bs, o_rs, o_bs;
    int M, K, T, B, tiles_t, n_mt, total_tiles;   // M, K: padded to 64 / 16
    int Mv, Kv, acc;                              // the matrix' own rows / inner size; acc: out += W . X
};

// XB: X and out are bfloat16 (activations under autocast): a bf16 value IS its own hi part, so the producers pack
// the loaded halves straight into the [token][8 k] image (v_perm_b32, no split), there is no lo image of X and a product
// is two MFMAs (W lo x X, W hi x X); the result is rounded to bf16 where it is stored.
template <bool XB>
__global__ __launch_bounds__(512, 2) void gemm_tokens_mfma_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nch = p.K / CK;
    const int G = gridDim.x;
    const int ntl = (p.total_tiles - (int)blockIdx.x + G - 1) / G;
    const int niter = ntl * nch;
    auto decode = [&](int tj, int &b, int &mt, int &t0) {
        int t = (int)blockIdx.x + tj * G;
        const int tt = t % p.tiles_t;
        t /= p.tiles_t;
        mt = t % p.n_mt;
        b = t / p.n_mt;
        t0 = tt * TT;
    };

    const int niter_pad = (niter + 2) / 3 * 3;   // both kinds of waves run this many barriers (see the producers)

    if (wv >= 4) {
        // ================= producers: 256 threads; thread = (k half, group of 4 tokens) of the chunk: 8 x 16-byte
        // buffer loads (scalar row offset + one per-lane token offset: no address arithmetic), hi/lo split, 8 LDS
        // writes.  THREE chunks stay in flight (register sets c % 3): one chunk takes the consumers ~0.4 us, a
        // memory round trip under load 1-2 us.  The loop is unrolled by three so the sets are static registers, and
        // padded to a multiple of three so that nothing that issues loads sits in a branch (past the end the
        // stream re-reads the last chunk into the buffer nobody reads).  Columns of tokens >= T read token 0's
        // data: GEMM columns are independent and those outputs are never stored.
        const int ptid = tid - 256;
        const int half = (wv - 4) >> 1, grp = ptid & 127;   // wave-uniform: the row offset of the loads is scalar
        v4u px[3][8], wr[3];   // (XB: only .x / .y of a px entry are loaded -- four bf16 tokens)
        int l_tj = 0, l_ch = 0, l_b, l_mt, l_t0;
        decode(0, l_b, l_mt, l_t0);
        constexpr unsigned ES = XB ? 2u : 4u;   // bytes per X element
        auto load = [&](v4u(&dst)[8], v4u &wdst) {
            const rsrc_t rs = make_rsrc((const char *)p.x + (long)l_b * p.x_bs * ES);
            const int tok = l_t0 + 4 * grp;
            const unsigned voff = (unsigned)(tok < p.T ? tok : 0) * ES;
            const unsigned row0 = (unsigned)(l_ch * CK + half * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {   // rows past the matrix (inner padded to 16): a valid row again, its weights are zero
                const unsigned rj = row0 + j < (unsigned)p.Kv ? row0 + j : (unsigned)p.Kv - 1u;
                if constexpr (XB) {
                    typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
                    const v2u_ t2 = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, rj * (unsigned)p.x_rs * ES, 0);
                    dst[j].x = t2.x;
                    dst[j].y = t2.y;
                } else {
                    dst[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, rj * (unsigned)p.x_rs * ES, 0);
                }
            }
            wdst = reinterpret_cast<const v4u *>(p.wp + ((long)l_mt * nch + l_ch) * (2 * 2 * 64 * 8))[ptid];
            if (l_ch + 1 < nch) {
                ++l_ch;
            } else if (l_tj + 1 < ntl) {
                l_ch = 0;
                ++l_tj;
                decode(l_tj, l_b, l_mt, l_t0);
            }
        };
        auto stage = [&](const v4u(&src)[8], const v4u &wsrc, unsigned char *buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned hw[4], lw[4];
                const int off = (half * TT + 4 * grp + i) * 16;
                if constexpr (XB) {
                    // token i of rows 2j, 2j+1: the low (i even) or high (i odd) halves of word i >> 1, packed (k even | k odd << 16)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        hw[j] = __builtin_amdgcn_perm(src[2 * j + 1][i >> 1], src[2 * j][i >> 1], (i & 1) ? 0x07060302u : 0x05040100u);
                    const v4u h = {hw[0], hw[1], hw[2], hw[3]};
                    *reinterpret_cast<v4u *>(buf + off) = h;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        split2(__uint_as_float(src[2 * j][i]), __uint_as_float(src[2 * j + 1][i]), hw[j], lw[j]);
                    const v4u h = {hw[0], hw[1], hw[2], hw[3]}, l = {lw[0], lw[1], lw[2], lw[3]};
                    *reinterpret_cast<v4u *>(buf + off) = h;
                    *reinterpret_cast<v4u *>(buf + XIMG + off) = l;
                }
            }
            *reinterpret_cast<v4u *>(buf + 2 * XIMG + ptid * 16) = wsrc;   // 256 x 16 B = hi and lo images of W
        };
        load(px[0], wr[0]);
        load(px[1], wr[1]);
        load(px[2], wr[2]);
        stage(px[0], wr[0], lds);
        load(px[0], wr[0]);
        MMU_LDS_BARRIER();
        // iteration it: stage chunk it+1 (set (it+1) % 3) into buffer (it+1) & 1, then refill that set with chunk it+4
        for (int it = 0; it < niter_pad; it += 3) {
            stage(px[1], wr[1], lds + ((it + 1) & 1) * STAGE);
            load(px[1], wr[1]);
            MMU_LDS_BARRIER();
            stage(px[2], wr[2], lds + ((it + 2) & 1) * STAGE);
            load(px[2], wr[2]);
            MMU_LDS_BARRIER();
            stage(px[0], wr[0], lds + ((it + 3) & 1) * STAGE);
            load(px[0], wr[0]);
            MMU_LDS_BARRIER();
        }
        return;
    }

    // ===================== consumers: wave cw owns tokens [128 cw, 128 cw + 128) of the tile, 64 rows
    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
    const int a_lane = (lane >> 5) * (64 * 16) + (lane & 31) * 16;
    const int b_lane = (lane >> 5) * (TT * 16) + (wv * 128 + (lane & 31)) * 16;
    MMU_LDS_BARRIER();
    int c_tj = 0, c_ch = 0;
    for (int it = 0; it < niter_pad; ++it) {
        if (it >= niter) {   // padding iterations (the producers' loop runs in threes): barrier only
            MMU_LDS_BARRIER();
            continue;
        }
        const unsigned char *cur = lds + (it & 1) * STAGE;
        bf16x8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            ah[m] = *reinterpret_cast<const bf16x8 *>(cur + 2 * XIMG + m * (32 * 16) + a_lane);
            al[m] = *reinterpret_cast<const bf16x8 *>(cur + 2 * XIMG + WIMG + m * (32 * 16) + a_lane);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            bh[n] = *reinterpret_cast<const bf16x8 *>(cur + n * (32 * 16) + b_lane);
            if constexpr (!XB) bl[n] = *reinterpret_cast<const bf16x8 *>(cur + XIMG + n * (32 * 16) + b_lane);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
                if constexpr (!XB) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
            }
        if (++c_ch == nch) {
            int b, mt, t0;
            decode(c_tj, b, mt, t0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int tok = t0 + wv * 128 + n * 32 + (lane & 31);
                    if (tok < p.T) {
                        const int r0 = mt * 64 + m * 32 + 4 * (lane >> 5);
                        if constexpr (XB) {
                            bf16_t *ob = (bf16_t *)p.out + (long)b * p.o_bs + (long)r0 * p.o_rs + tok;
                            float old[16];
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int ro = (e & 3) + 8 * (e >> 2);
                                old[e] = (p.acc && r0 + ro < p.Mv) ? to_f32(ob[(long)ro * p.o_rs]) : 0.f;
                            }
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int ro = (e & 3) + 8 * (e >> 2);
                                if (r0 + ro < p.Mv) ob[(long)ro * p.o_rs] = from_f32<bf16_t>(old[e] + acc[m][n][e]);
                            }
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
                            continue;
                        }
                        float *op = (float *)p.out + (long)b * p.o_bs + (long)r0 * p.o_rs + tok;
                        if (r0 + 28 < p.Mv && !p.acc) {   // all 16 rows of this lane inside the matrix, plain store
#pragma unroll
                            for (int e = 0; e < 16; ++e) op[((e & 3) + 8 * (e >> 2)) * p.o_rs] = acc[m][n][e];
                        } else {
                            // all 16 loads of the old values first, then the stores: interleaved, every load would wait
                            // behind the store in front of it (the compiler cannot tell the rows apart)
                            float old[16];
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int ro = (e & 3) + 8 * (e >> 2);
                                old[e] = (p.acc && r0 + ro < p.Mv) ? op[(long)ro * p.o_rs] : 0.f;
                            }
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int ro = (e & 3) + 8 * (e >> 2);
                                if (r0 + ro < p.Mv) op[(long)ro * p.o_rs] = old[e] + acc[m][n][e];
                            }
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
                }
            c_ch = 0;
            ++c_tj;
        }
        MMU_LDS_BARRIER();
    }
}

// ---- few tokens, deep inner dimension -------------------------------------------------------------------------
// The DSC products and projections of the maps <= 64 x 64 (8 x (512 x 1536) . (1536 x 256), 8 x (256 x 768) . (768 x
// 1,024), 8 x (128 x 384) . (384 x 4,096), the 64 <-> 128 / 256 projections at 4,096 tokens, ...) have 64-128 tiles of
// 64 rows x 512 tokens: the kernel above leaves most CUs idle on them (and they were library GEMMs, ~75 launches of
// 24-30 us per step + a zero-fill each).  Here a workgroup takes 64 rows x 32 TOKENS and its four waves split the
// inner dimension (chunk c to wave c % 4): 512-2,048 workgroups.  No LDS staging at all -- a lane's B fragment of
// v_mfma_f32_32x32x16_bf16 is 8 consecutive k for ONE token, i.e. 8 dword loads at the row stride, 32 lanes on 32
// consecutive tokens (one 128-byte line per row); its A fragment is 16 bytes of the prepared weight image.  Loads run
// one chunk ahead in registers.  The four partial tiles meet in LDS and are added in wave order (deterministic).
// Work items are numbered so that the row tiles of one token block run on the same XCD (workgroup ids go round-robin over
// the 8 XCDs): X is fetched into one L2 instead of n_mt of them.
// NB = 32-token blocks per wave (tile = 64 rows x 32 NB tokens): the A fragments (three 16-byte loads per row tile and
// chunk -- three quarters of the kernel's L1 traffic at NB = 1) serve NB token blocks.
template <int NB>
__global__ __launch_bounds__(256, NB == 1 ? 4 : 2) void gemm_tokens_small_kernel(GemmArgs p, int per_xcd) {
    __shared__ float red[4 * 32 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if ((int)blockIdx.x >> 3 >= per_xcd || item >= p.total_tiles) return;
    int t = item;
    const int mt = t % p.n_mt;
    t /= p.n_mt;
    const int tt = t % p.tiles_t;
    const int b = t / p.tiles_t;
    const int nch = p.K / CK;
    const int half = lane >> 5;
    const int tok0 = tt * (32 * NB) + (lane & 31);
    // X through a buffer resource: the row offset of a load is SCALAR ((chunk * 16 + j) rows), the lane's part (its k half
    // and its token) one VGPR that never changes -- as plain pointers every load cost seven vector instructions of 64-bit
    // address arithmetic, two of them quarter-rate multiplies (more issue time than the chunk's twelve MFMAs)
    const rsrc_t xrs = make_rsrc((const float *)p.x + (long)b * p.x_bs);
    unsigned xoff[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
        xoff[n] = ((unsigned)(half * 8) * (unsigned)p.x_rs + (unsigned)(tok0 + 32 * n < p.T ? tok0 + 32 * n : 0)) * 4u;
    const v4u *wimg = reinterpret_cast<const v4u *>(p.wp) + (long)mt * nch * 256 + half * 64 + (lane & 31);
    const v4u *wimg3 = reinterpret_cast<const v4u *>(p.wp) + (long)(p.M / 64) * nch * 256 + (long)mt * nch * 128 + half * 64 + (lane & 31);
    f32x16 acc[NB][2];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][m][e] = 0.f;
    auto loadx = [&](int ch, float(&v)[NB][8]) {
        if (ch * CK + CK <= p.Kv) {   // every row of the chunk inside the matrix
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned soff = (unsigned)(ch * CK + j) * (unsigned)p.x_rs * 4u;
#pragma unroll
                for (int n = 0; n < NB; ++n) v[n][j] = buf_load1(xrs, xoff[n], soff);
            }
        } else {                      // the last chunk of an inner size that is no multiple of 16: rows past the matrix
            const int r0 = ch * CK + half * 8;   // read a valid row again (their weights are zero)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = r0 + j < p.Kv ? r0 + j : p.Kv - 1;
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    v[n][j] = buf_load1(xrs, xoff[n] - (unsigned)(half * 8) * (unsigned)p.x_rs * 4u + (unsigned)r * (unsigned)p.x_rs * 4u, 0);
            }
        }
    };
    auto loadw = [&](int ch, v4u(&w)[6]) {
        const v4u *q = wimg + (long)ch * 256, *q3 = wimg3 + (long)ch * 128;
        w[0] = q[0]; w[1] = q[32]; w[2] = q[128]; w[3] = q[160];   // hi rows 0-31, 32-63; mid rows 0-31, 32-63
        w[4] = q3[0]; w[5] = q3[32];                                // lo
    };
    // FLOAT32-GRADE products: both operands as three bf16 parts (8 + 8 + 8 mantissa bits), the six products down to
    // 2^-16 of the leading one (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi).  These are the deep products of the small
    // maps: as library GEMMs they were exact float32, and with the two-part split of the 512-token kernel the
    // d_state-64 model's logits moved from 5.2e-4 to 1.2e-3 off the oracle (tools/dbg/config5_fwd_err.py).
    // Two chunks of this wave in flight (register sets 0 / 1, the loop unrolled by two so that they are static).
    float xv[2][NB][8];
    v4u wr[2][6];
    auto compute = [&](const float(&xs)[NB][8], const v4u(&ws)[6]) {
        bf16x8 bh[NB], bm[NB], bl[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            unsigned hw[4], mw[4], lw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = xs[n][2 * q], c = xs[n][2 * q + 1];
                hw[q] = pack_bf16(a, c);
                const float ra = a - __builtin_bit_cast(float, hw[q] << 16), rc = c - __builtin_bit_cast(float, hw[q] & 0xffff0000u);
                mw[q] = pack_bf16(ra, rc);
                lw[q] = pack_bf16(ra - __builtin_bit_cast(float, mw[q] << 16), rc - __builtin_bit_cast(float, mw[q] & 0xffff0000u));
            }
            const v4u hq = {hw[0], hw[1], hw[2], hw[3]}, mq = {mw[0], mw[1], mw[2], mw[3]}, lq = {lw[0], lw[1], lw[2], lw[3]};
            bh[n] = __builtin_bit_cast(bf16x8, hq);
            bm[n] = __builtin_bit_cast(bf16x8, mq);
            bl[n] = __builtin_bit_cast(bf16x8, lq);
        }
        // the six terms, smallest first; within a term the 2 NB accumulators are independent chains
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, ws[m]), am = __builtin_bit_cast(bf16x8, ws[2 + m]),
                                 al = __builtin_bit_cast(bf16x8, ws[4 + m]);
                    const bf16x8 a = term == 0 ? al : (term == 2 || term == 3) ? am : ah;
                    const bf16x8 bb = (term == 0 || term == 3 || term == 5) ? bh[n] : (term == 1) ? bl[n] : bm[n];
                    acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc[n][m], 0, 0, 0);
                }
    };
#pragma unroll
    for (int s = 0; s < 2; ++s)
        if (wv + 4 * s < nch) {
            loadx(wv + 4 * s, xv[s]);
            loadw(wv + 4 * s, wr[s]);
        }
    for (int ch = wv; ch < nch; ch += 8) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = ch + 4 * s;
            if (c < nch) {
                compute(xv[s], wr[s]);
                if (c + 8 < nch) {
                    loadx(c + 8, xv[s]);
                    loadw(c + 8, wr[s]);
                }
            }
        }
    }
    // the four waves' partial tiles, one token block at a time through the 32 KB of LDS, added in wave order
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        if (n) __syncthreads();
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(wv * 32 + m * 16 + e) * 64 + lane] = acc[n][m][e];
        __syncthreads();
        const int tok = tok0 + 32 * n;
        if (tok < p.T) {
            float *ob = (float *)p.out + (long)b * p.o_bs + tok;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int v = 8 * wv + i;
                const float s = ((red[v * 64 + lane] + red[(32 + v) * 64 + lane]) + red[(64 + v) * 64 + lane]) + red[(96 + v) * 64 + lane];
                const int e = v & 15;
                const int row = mt * 64 + (v >> 4) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (row < p.Mv) {
                    float *o = ob + (long)row * p.o_rs;
                    *o = p.acc ? *o + s : s;
                }
            }
        }
    }
}

}  // namespace

// 64-row x 512-token tiles below which the 32-token kernel takes the product (MMUNET_GEMM_TOKENS_SMALL_BELOW; 0: never)
static long gemm_tokens_small_below() {
    static long v = -1;
    if (v < 0) {
        const char *e = getenv("MMUNET_GEMM_TOKENS_SMALL_BELOW");
        v = e ? atol(e) : 520;   // swept on the training step: 0 / 192 / 320 / 520 / 1100 / 2100 = 34.36 / 32.30 / 32.18 / 32.05 / 32.12 / 32.18 ms
    }
    return v;
}

static int gemm_tokens_small_nb() {   // MMUNET_GEMM_TOKENS_SMALL_NB=1: always one token block per wave (A/B)
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MMUNET_GEMM_TOKENS_SMALL_NB");
        v = e ? atoi(e) : 0;
    }
    return v;
}

extern "C" size_t mmu_gemm_tokens_workspace_bytes(int rows, int inner) {
    if (rows <= 0 || inner <= 0) return 0;
    return (size_t)((rows + 63) & ~63) * ((inner + CK - 1) & ~(CK - 1)) * 3 * sizeof(unsigned short);   // hi | lo images, then the third parts
}

extern "C" int mmu_gemm_tokens_mfma(const mmu_gemm_tokens_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "gemm_tokens_mfma: null params");
    MMU_CHECK(p->rows > 0 && p->inner > 0, "gemm_tokens_mfma: rows and inner must be positive (got %d, %d)", p->rows, p->inner);
    const int Mp = (p->rows + 63) & ~63, Kp = (p->inner + CK - 1) & ~(CK - 1);
    MMU_CHECK(p->tokens > 0 && p->batch > 0, "gemm_tokens_mfma: empty problem");
    const bool xb = p->x_dtype == MMU_DTYPE_BF16;
    MMU_CHECK(p->x_dtype == p->out_dtype && (p->x_dtype == MMU_DTYPE_F32 || xb),
              "gemm_tokens_mfma: x and out both float32 or both bfloat16 (got dtypes %d, %d)", p->x_dtype, p->out_dtype);
    MMU_CHECK(p->tokens % 4 == 0 && p->x_rs % 4 == 0 && p->x_bs % 4 == 0 && ((uintptr_t)p->x & (xb ? 7 : 15)) == 0,
              "gemm_tokens_mfma: tokens, x_rs, x_bs must be multiples of 4 and x 16-byte (bfloat16: 8-byte) aligned");
    MMU_CHECK((long)p->inner * p->x_rs * 4 < (1L << 31), "gemm_tokens_mfma: x rows span more than 2 GB");
    MMU_CHECK(p->x && p->out && p->workspace, "gemm_tokens_mfma: x, out, workspace are required");
    MMU_CHECK(((uintptr_t)p->workspace & 15) == 0, "gemm_tokens_mfma: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long nw = (long)Mp * Kp;
    if (p->weight) {   // NULL: the workspace already holds this weight's image (mmu_gemm_tokens_prepare_batch)
        gemm_tokens_prep_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, st>>>(p->weight, p->w_ld, (unsigned short *)p->workspace,
                                                                             p->rows, p->inner, p->transposed_weight ? 1 : 0);
        MMU_HIP_LAUNCH_CHECK("gemm_tokens_mfma(prep)");
    }
    GemmArgs a;
    a.x = p->x; a.wp = (const unsigned short *)p->workspace; a.out = p->out;
    a.x_rs = p->x_rs; a.x_bs = p->x_bs; a.o_rs = p->out_rs; a.o_bs = p->out_bs;
    a.M = Mp; a.K = Kp; a.T = p->tokens; a.B = p->batch;
    a.Mv = p->rows; a.Kv = p->inner; a.acc = p->accumulate ? 1 : 0;
    a.tiles_t = (p->tokens + TT - 1) / TT;
    a.n_mt = Mp / 64;
    if (!xb && (long)a.tiles_t * a.n_mt * p->batch < gemm_tokens_small_below()) {   // too few 512-token tiles to fill the chip
        // two token blocks per wave when that still leaves two workgroups per CU
        const long items64 = (long)((p->tokens + 63) / 64) * a.n_mt * p->batch;
        const int nb = items64 >= 2L * mmu_cu_count() && gemm_tokens_small_nb() != 1 ? 2 : 1;
        a.tiles_t = (p->tokens + 32 * nb - 1) / (32 * nb);
        const long items = (long)a.tiles_t * a.n_mt * p->batch;
        MMU_CHECK(items < (1L << 28), "gemm_tokens_mfma: too many tiles");
        a.total_tiles = (int)items;
        const int per_xcd = (int)((items + 7) / 8);
        if (nb == 2)
            gemm_tokens_small_kernel<2><<<per_xcd * 8, 256, 0, st>>>(a, per_xcd);
        else
            gemm_tokens_small_kernel<1><<<per_xcd * 8, 256, 0, st>>>(a, per_xcd);
        MMU_HIP_LAUNCH_CHECK("gemm_tokens_mfma(small)");
        return 0;
    }
    static unsigned long long attr_mask = 0, attr_mask_b = 0;  // per device
    if (hipError_t e = xb ? mmu_set_lds_once(gemm_tokens_mfma_kernel<true>, LDS_BYTES, attr_mask_b)
                          : mmu_set_lds_once(gemm_tokens_mfma_kernel<false>, LDS_BYTES, attr_mask);
        e != hipSuccess)
        return mmu_fail("gemm_tokens_mfma: LDS attribute: %s", hipGetErrorString(e));
    const long total = (long)a.tiles_t * a.n_mt * p->batch;
    MMU_CHECK(total < (1L << 30), "gemm_tokens_mfma: too many tiles");
    a.total_tiles = (int)total;
    const int n_cu = mmu_cu_count();
    const int grid = total < n_cu ? (int)total : n_cu;   // persistent: one workgroup per CU (register-bound: 2 waves/SIMD)
    if (xb)
        gemm_tokens_mfma_kernel<true><<<grid, 512, LDS_BYTES, st>>>(a);
    else
        gemm_tokens_mfma_kernel<false><<<grid, 512, LDS_BYTES, st>>>(a);
    MMU_HIP_LAUNCH_CHECK("gemm_tokens_mfma");
    return 0;
}

extern "C" int mmu_gemm_tokens_prepare_batch(const int64_t *table, int n_items, int64_t max_elements, void *stream) {
    MMU_CHECK(table != nullptr && n_items > 0 && n_items <= 65535 && max_elements > 0,
              "gemm_tokens_prepare_batch: need a device table, 1..65535 items and the largest rows * inner");
    static_assert(sizeof(long) == sizeof(int64_t), "the table is read as long");
    const long blocks = (max_elements + 255) / 256;
    dim3 grid((unsigned)(blocks < 64 ? blocks : 64), (unsigned)n_items);
    gemm_tokens_prep_batch_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const long *>(table));
    MMU_HIP_LAUNCH_CHECK("gemm_tokens_prepare_batch");
    return 0;
}
