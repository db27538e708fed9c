// stem7_mfma.hip -- MM_Net's stem, nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
// (src/UM_Net/MMUNet.py:492), forward and weight gradient on the bf16 matrix cores with FLOAT32-GRADE products.
//
// The stem feeds every other layer: with the two-part hi/lo split of conv3x3_mfma.hip (2^-16 products) it cost half of
// the forward parity margin (HISTORY.md), so it stayed in MIOpen (203 us forward, 177 + 49 us weight gradient at
// 8 x 3 x 512 x 512) -- the last library kernel of the float32 step.  Here both operands are split into THREE bf16
// parts (8 + 8 + 8 mantissa bits) and a product is six v_mfma_f32_32x32x16_bf16 (hi*hi, hi*mid, mid*hi, mid*mid,
// hi*lo, lo*hi: everything down to 2^-16 of the leading term; float32 accumulation) -- 416 TFLOP/s of float32-grade peak
// against the 157 of v_mfma_f32_32x32x2_f32.
//
// GEMM view.  The 147 = 3 x 7 x 7 taps are laid out as 22 groups of 8: group g = (channel c, kernel row ky), g < 21, holds
// kx = 0..6 and one zero tap; group 21 is all zero.  K (forward) / N (weight gradient) = 176.
//   forward : out[co][pixel] = sum_k W[co][k] * patch[k][pixel]     M = 64 co, N = pixels, K = 176 (11 chunks of 16)
//   wgrad   : dW[co][k] = sum_pixel dout[co][pixel] * patch[k][pixel]   M = 64 co, N = 176 -> 6 blocks of 32, K = pixels
// A workgroup (4 waves) takes an 8-row x 32-column tile of OUTPUT pixels; its input patch -- 3 channels x 21 rows x 69
// columns -- is split once into the three bf16 parts while it is staged in LDS ([part][c][row][72] bf16, 27 KB), and the
// im2col operand is pure addressing: a lane's 8 consecutive k of the forward are 8 consecutive patch columns (four
// ds_read_b32 per part), the weight gradient's 8 consecutive pixels are 8 patch columns two apart (ds_read_u16).
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int OTH = 8, OTW = 32;                  // output tile
constexpr int PR = 2 * OTH + 5, PC = 72;          // patch rows (21), padded columns (69 used)
constexpr int PART = 3 * PR * PC;                 // elements of one bf16 part: 4,536
constexpr int NCH = 11, NG = 21;                  // chunks of 16 k; real groups

__device__ __forceinline__ unsigned pk(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split3(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    h = pk(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    m = pk(ra, rb);
    l = pk(ra - __builtin_bit_cast(float, m << 16), rb - __builtin_bit_cast(float, m & 0xffff0000u));
}

// W [64][3][7][7] f32 -> [chunk 11][part 3][half 2][64 co][8 k] bf16   (k = kx; group = 2 chunk + half = c * 7 + ky)
__global__ __launch_bounds__(256) void stem7_prep_kernel(const float *__restrict__ w, unsigned short *__restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= NCH * 2 * 64 * 8) return;
    const int k = idx & 7, co = (idx >> 3) & 63, half = (idx >> 9) & 1, ch = idx >> 10;
    const int g = 2 * ch + half;
    const float v = (g < NG && k < 7) ? w[(co * 3 + g / 7) * 49 + (g % 7) * 7 + k] : 0.f;
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    const long base = (((long)ch * 3) * 2 + half) * 512 + co * 8 + k;
    out[base] = __builtin_bit_cast(unsigned short, h);
    out[base + 1024] = __builtin_bit_cast(unsigned short, m);
    out[base + 2048] = __builtin_bit_cast(unsigned short, l);
}

struct StemArgs {
    const float *x, *dout;
    const unsigned short *wp;
    float *out, *ws;
    int B, H, W, Ho, Wo, tiles_x, tiles_y, total_tiles;
};

// the input patch of output tile (b, y0, x0) -> LDS, three bf16 parts; pixels outside the image are 0
__device__ __forceinline__ void stage_patch(const StemArgs &p, int b, int y0, int x0, unsigned short *lds, int tid) {
    const long HW = (long)p.H * p.W;
    const float *xb = p.x + (long)b * 3 * HW;
    float v[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const int e = tid + 256 * i;
        const int c = e / (PR * PC), rem = e - c * (PR * PC);
        const int r = rem / PC, cc = rem - r * PC;
        const int gy = 2 * y0 - 3 + r, gx = 2 * x0 - 3 + cc;
        const bool inb = e < PART && cc < 69 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        v[i] = inb ? xb[c * HW + (long)gy * p.W + gx] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const int e = tid + 256 * i;
        if (e < PART) {
            const __bf16 h = (__bf16)v[i];
            const float r1 = v[i] - (float)h;
            const __bf16 m = (__bf16)r1;
            const __bf16 l = (__bf16)(r1 - (float)m);
            lds[e] = __builtin_bit_cast(unsigned short, h);
            lds[PART + e] = __builtin_bit_cast(unsigned short, m);
            lds[2 * PART + e] = __builtin_bit_cast(unsigned short, l);
        }
    }
}

// element offset of (c, ky) of group g inside a part: (c * 21 + ky) * 72; groups past the last real one alias group 20
__device__ __forceinline__ int group_off(int g) {
    const int gg = g < NG ? g : NG - 1;
    const int c = (gg * 37) >> 8;   // gg / 7 for gg <= 20
    return (c * PR + (gg - 7 * c)) * PC;
}

__device__ __forceinline__ void mfma6(f32x16 &acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);   // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

__global__ __launch_bounds__(256) void stem7_fwd_kernel(StemArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[3 * PART];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t = blockIdx.x;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y, b = t / p.tiles_y;
    const int y0 = ty * OTH, x0 = tx * OTW;
    stage_patch(p, b, y0, x0, lds, tid);
    const int half = lane >> 5, px = lane & 31;
    const v4u *wimg = reinterpret_cast<const v4u *>(p.wp) + half * 64 + px;   // + ((ch * 3 + part) * 2) * 64 + 32 m
    f32x16 acc[2][2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][m][e] = 0.f;
    v4u wr[2][6];
    auto loadw = [&](int ch, v4u(&w)[6]) {
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int m = 0; m < 2; ++m) w[part * 2 + m] = wimg[((ch * 3 + part) * 2) * 64 + 32 * m];
    };
    loadw(0, wr[0]);
    __syncthreads();
    const unsigned *lds32 = reinterpret_cast<const unsigned *>(lds);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch + 1 < NCH) loadw(ch + 1, wr[(ch + 1) & 1]);
        const int goff = group_off(2 * ch + half);
        bf16x8 bf[2][3];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int e0 = goff + (2 * (2 * wv + n)) * PC + 2 * px;   // even: a dword boundary
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                const unsigned *q = lds32 + ((part * PART + e0) >> 1);
                const v4u f = {q[0], q[1], q[2], q[3]};
                bf[n][part] = __builtin_bit_cast(bf16x8, f);
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const v4u(&w)[6] = wr[ch & 1];
            const bf16x8 af[3] = {__builtin_bit_cast(bf16x8, w[m]), __builtin_bit_cast(bf16x8, w[2 + m]),
                                  __builtin_bit_cast(bf16x8, w[4 + m])};
#pragma unroll
            for (int n = 0; n < 2; ++n) mfma6(acc[n][m], af, bf[n]);
        }
    }
    const long HWo = (long)p.Ho * p.Wo;
    const int ox = x0 + px;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int oy = y0 + 2 * wv + n;
        if (oy < p.Ho && ox < p.Wo) {
            float *op = p.out + ((long)b * 64 + 4 * half) * HWo + (long)oy * p.Wo + ox;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 16; ++e) op[(m * 32 + (e & 3) + 8 * (e >> 2)) * HWo] = acc[n][m][e];
        }
    }
}

// Weight gradient.  Persistent workgroups walk the output tiles and keep dW[64][192] in registers: wave w holds the
// rows 32 (w & 1) .. + 31 of the column blocks 3 (w >> 1) .. + 2 (48 accumulator registers).  A tile = 16 k-steps of 16
// pixels (half a tile row): the dout fragment is 8 consecutive pixels of one channel from global memory (split in
// registers), the patch fragment 8 pixels of one tap.  Partials per workgroup -> workspace, added in a fixed order.
__global__ __launch_bounds__(256) void stem7_wgrad_kernel(StemArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[3 * PART];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int mrow = (wv & 1) * 32 + l31;              // dout channel of this lane's A fragment
    const int nb0 = 3 * (wv >> 1);
    int boff[3];                                       // element offset of this lane's tap (group, kx) in a part
#pragma unroll
    for (int j = 0; j < 3; ++j) boff[j] = group_off((nb0 + j) * 4 + (l31 >> 3)) + (l31 & 7);
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const long HWo = (long)p.Ho * p.Wo;
    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y, b = t / p.tiles_y;
        const int y0 = ty * OTH, x0 = tx * OTW;
        __syncthreads();                               // the previous tile's fragments have been read
        stage_patch(p, b, y0, x0, lds, tid);
        __syncthreads();
        const float *dr = p.dout + ((long)b * 64 + mrow) * HWo;
        // dout fragment of k-step q: 8 consecutive pixels of this lane's channel; the loads run one k-step ahead
        auto loadg = [&](int q, float(&g)[8]) {
            const int oy = y0 + (q >> 1), ox = x0 + (q & 1) * 16 + half * 8;
            if (oy < p.Ho && ox + 8 <= p.Wo) {         // (Wo % 8 == 0: checked by the host)
                const float4 g0 = *reinterpret_cast<const float4 *>(dr + (long)oy * p.Wo + ox);
                const float4 g1 = *reinterpret_cast<const float4 *>(dr + (long)oy * p.Wo + ox + 4);
                g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = 0.f;
            }
        };
        float gq[2][8];
        loadg(0, gq[0]);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int oyl = q >> 1, oxl = (q & 1) * 16 + half * 8;
            if (q + 1 < 16) loadg(q + 1, gq[(q + 1) & 1]);
            const float(&g)[8] = gq[q & 1];
            unsigned hw[4], mw[4], lw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split3(g[2 * j], g[2 * j + 1], hw[j], mw[j], lw[j]);
            const v4u hq = {hw[0], hw[1], hw[2], hw[3]}, mq = {mw[0], mw[1], mw[2], mw[3]}, lq = {lw[0], lw[1], lw[2], lw[3]};
            const bf16x8 af[3] = {__builtin_bit_cast(bf16x8, hq), __builtin_bit_cast(bf16x8, mq), __builtin_bit_cast(bf16x8, lq)};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int e0 = boff[j] + 2 * oyl * PC + 2 * oxl;
                bf16x8 bf[3];
#pragma unroll
                for (int part = 0; part < 3; ++part) {
                    const unsigned short *s = lds + part * PART + e0;
                    unsigned d[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[i] = (unsigned)s[4 * i] | ((unsigned)s[4 * i + 2] << 16);
                    const v4u f = {d[0], d[1], d[2], d[3]};
                    bf[part] = __builtin_bit_cast(bf16x8, f);
                }
                mfma6(acc[j], af, bf);
            }
        }
    }
    float *wsb = p.ws + (long)blockIdx.x * (64 * 192) + (long)((wv & 1) * 32 + 4 * half) * 192 + l31;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) wsb[((e & 3) + 8 * (e >> 2)) * 192 + (nb0 + j) * 32] = acc[j][e];
}

// dW[co][c][ky][kx] = sum over the workgroups' partials [wg][co][group * 8 + kx]: a workgroup takes 32 results, its eight
// groups of 32 threads each add every eighth partial in workgroup order, the eight sums are added in group order
__global__ __launch_bounds__(256) void stem7_wgrad_sum_kernel(const float *__restrict__ ws, float *__restrict__ dw, int nwg) {
    __shared__ float part[8][32];
    const int l = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + l;
    float a = 0.f;
    if (idx < 64 * 147) {
        const int co = idx / 147, r = idx - co * 147;
        const int g = r / 7, kx = r - 7 * g;
        const float *s = ws + (long)co * 192 + g * 8 + kx;
        for (int i = grp; i < nwg; i += 8) a += s[(long)i * (64 * 192)];
    }
    part[grp][l] = a;
    __syncthreads();
    if (grp == 0 && idx < 64 * 147) {
        float t = part[0][l];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += part[k][l];
        dw[idx] = t;
    }
}

int stem_check(const mmu_stem7_params *p, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->batch > 0 && p->height > 0 && p->width > 0, "%s: empty tensor", name);
    MMU_CHECK(p->height % 2 == 0 && p->width % 16 == 0, "%s: height must be even and width a multiple of 16 (got %d x %d)",
              name, p->height, p->width);
    MMU_CHECK(p->input && p->workspace, "%s: input and workspace are required", name);
    MMU_CHECK(((uintptr_t)p->workspace & 15) == 0, "%s: workspace must be 16-byte aligned", name);
    return 0;
}

void stem_fill(const mmu_stem7_params *p, StemArgs &a) {
    a.x = p->input; a.dout = p->dout; a.out = p->out;
    a.B = p->batch; a.H = p->height; a.W = p->width; a.Ho = p->height / 2; a.Wo = p->width / 2;
    a.tiles_x = (a.Wo + OTW - 1) / OTW;
    a.tiles_y = (a.Ho + OTH - 1) / OTH;
    a.total_tiles = a.tiles_x * a.tiles_y * p->batch;
}

inline int stem_wgrad_blocks(int total_tiles) {
    const int n = 2 * mmu_cu_count();
    return total_tiles < n ? total_tiles : n;
}

}  // namespace

// forward: the bf16 image of the weight (67,584 B); weight gradient: the workgroups' partial sums
extern "C" size_t mmu_stem7_workspace_bytes(int batch, int height, int width, int backward) {
    if (!backward) return (size_t)NCH * 3 * 2 * 64 * 8 * sizeof(unsigned short);
    const int tiles = ((width / 2 + OTW - 1) / OTW) * ((height / 2 + OTH - 1) / OTH) * batch;
    return (size_t)stem_wgrad_blocks(tiles) * 64 * 192 * sizeof(float);
}

extern "C" int mmu_stem7_fwd(const mmu_stem7_params *p, void *stream) {
    if (int r = stem_check(p, "stem7_fwd")) return r;
    MMU_CHECK(p->weight && p->out, "stem7_fwd: weight and out are required");
    hipStream_t st = (hipStream_t)stream;
    stem7_prep_kernel<<<(NCH * 2 * 64 * 8 + 255) / 256, 256, 0, st>>>(p->weight, (unsigned short *)p->workspace);
    MMU_HIP_LAUNCH_CHECK("stem7_fwd(prep)");
    StemArgs a = {};
    stem_fill(p, a);
    a.wp = (const unsigned short *)p->workspace;
    stem7_fwd_kernel<<<a.total_tiles, 256, 0, st>>>(a);
    MMU_HIP_LAUNCH_CHECK("stem7_fwd");
    return 0;
}

extern "C" int mmu_stem7_wgrad(const mmu_stem7_params *p, void *stream) {
    if (int r = stem_check(p, "stem7_wgrad")) return r;
    MMU_CHECK(p->dout && p->dweight, "stem7_wgrad: dout and dweight are required");
    MMU_CHECK(((uintptr_t)p->dout & 15) == 0, "stem7_wgrad: dout must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    StemArgs a = {};
    stem_fill(p, a);
    a.ws = (float *)p->workspace;
    const int nwg = stem_wgrad_blocks(a.total_tiles);
    stem7_wgrad_kernel<<<nwg, 256, 0, st>>>(a);
    MMU_HIP_LAUNCH_CHECK("stem7_wgrad");
    stem7_wgrad_sum_kernel<<<(64 * 147 + 31) / 32, 256, 0, st>>>(a.ws, p->dweight, nwg);
    MMU_HIP_LAUNCH_CHECK("stem7_wgrad(sum)");
    return 0;
}
