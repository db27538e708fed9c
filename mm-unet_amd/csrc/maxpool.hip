// maxpool.hip -- backward of nn.MaxPool2d(kernel_size=3, stride=2, padding=1) as a gather.
//
// (round 3: also the forward, with one-byte arg-max codes, and a backward that reads them -- mmu_maxpool3s2_fwd / _bwd_codes.)
// Where it sits: MM_Net's stem (src/UM_Net/MMUNet.py:493,537 `self.maxpool`, applied to the 64 x 256 x 256 stem map).  The
// forward stays ATen's (0.09 ms; it returns the arg-max indices); ATen's backward zero-fills d input and scatters the
// output gradients with float atomics (windows overlap): 18 + 221 us for [8, 64, 128, 128] -> [8, 64, 256, 256].
// Here every input pixel looks at the (at most four) windows that contain it and adds the gradients of those whose
// recorded arg-max is this pixel: no atomics, no zero fill, bit-reproducible; 134 MB written + 100 MB read.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float *__restrict__ g, const long *__restrict__ idx,
                                                             float *__restrict__ dx, int H, int W, int OH, int OW,
                                                             long planes) {
    const int wq = (W + 3) / 4;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= planes * H * wq) return;
    const int q = (int)(t % wq);
    const long r = t / wq;
    const int y = (int)(r % H);
    const long plane = r / H;
    const float *gp = g + plane * OH * OW;
    const long *ip = idx + plane * OH * OW;
    // windows (oy, ox) cover rows 2 oy - 1 .. 2 oy + 1: input row y lies in oy = y / 2 and, for odd y, (y + 1) / 2
    const int oy0 = y >> 1, oy1 = (y + 1) >> 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = q * 4 + j;
        if (x >= W) break;
        const long me = (long)y * W + x;
        const int ox0 = x >> 1, ox1 = (x + 1) >> 1;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = a == 0 ? oy0 : oy1;
            if ((a == 1 && oy1 == oy0) || oy >= OH) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ox = b == 0 ? ox0 : ox1;
                if ((b == 1 && ox1 == ox0) || ox >= OW) continue;
                const long o = (long)oy * OW + ox;
                if (ip[o] == me) acc[j] += gp[o];
            }
        }
    }
    float *dst = dx + (plane * H + y) * W + q * 4;
    if ((W & 3) == 0) {
        *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
        for (int j = 0; j < 4 && q * 4 + j < W; ++j) dst[j] = acc[j];
    }
}

// Forward with the arg-max kept as ONE byte per output (position 3 * dy + dx inside its window; ATen: an int64 plane
// index, 8x the bytes, and 97 us for [8, 64, 256, 256]).  Same choice as ATen's kernel among equal values: the first in
// row-major window order (strict >), NaN wins.  thread = one output element.
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                             unsigned char *__restrict__ code, int H, int W, int OH, int OW,
                                                             long planes) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long per = (long)OH * OW;
    if (t >= planes * per) return;
    const long plane = t / per;
    const int o = (int)(t - plane * per);
    const int oy = o / OW, ox = o - oy * OW;
    const float *xp = x + plane * H * W;
    float best = -INFINITY;
    int bc = 0;
    bool first = true;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int y = 2 * oy - 1 + dy;
        if (y < 0 || y >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = 2 * ox - 1 + dx;
            if (xx < 0 || xx >= W) continue;
            const float v = xp[(long)y * W + xx];
            if (first || v > best || v != v) {   // (the first valid element stands in for ATen's -inf start)
                best = v;
                bc = dy * 3 + dx;
                first = false;
            }
        }
    }
    out[t] = best;
    code[t] = (unsigned char)bc;
}

// Backward from the byte codes: every input pixel adds the gradients of the (at most four) windows whose code points at it.
__global__ __launch_bounds__(256) void maxpool3s2_bwd_code_kernel(const float *__restrict__ g,
                                                                  const unsigned char *__restrict__ code,
                                                                  float *dx, const float *addend, int H, int W, int OH,
                                                                  int OW, unsigned total) {
    const int wq = (W + 3) / 4;
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int q = (int)(t % (unsigned)wq);
    const unsigned r = t / (unsigned)wq;
    const int y = (int)(r % (unsigned)H);
    const long plane = r / (unsigned)H;
    const float *gp = g + plane * OH * OW;
    const unsigned char *cp = code + plane * OH * OW;
    const int oy0 = y >> 1, oy1 = (y + 1) >> 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = q * 4 + j;
        if (x >= W) break;
        const int ox0 = x >> 1, ox1 = (x + 1) >> 1;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = a == 0 ? oy0 : oy1;
            if ((a == 1 && oy1 == oy0) || oy >= OH) continue;
            const int dy = y - (2 * oy - 1);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ox = b == 0 ? ox0 : ox1;
                if ((b == 1 && ox1 == ox0) || ox >= OW) continue;
                const int o = oy * OW + ox;
                if (cp[o] == dy * 3 + (x - (2 * ox - 1))) acc[j] += gp[o];
            }
        }
    }
    float *dst = dx + (plane * H + y) * W + q * 4;
    if (addend) {
        const float *ad = addend + (plane * H + y) * W + q * 4;
        for (int j = 0; j < 4 && q * 4 + j < W; ++j) acc[j] += ad[j];
    }
    if ((W & 3) == 0) {
        *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
        for (int j = 0; j < 4 && q * 4 + j < W; ++j) dst[j] = acc[j];
    }
}

// The same two for W % 8 == 0 and even H (the stem: 256 x 256), vector loads and stores.
// Forward: thread = four neighbouring outputs of one row; per input row two float4 (columns 8q .. 8q+7) and the column left
// of them.
// io_t = float or bf16_t (activations under autocast: read and written natively; comparisons and gradient sums in float32)
template <typename io_t>
__device__ __forceinline__ void ld8(const io_t *p, float (&o)[8]) {
    if constexpr (sizeof(io_t) == 4) {
        const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else {
        const uint4 v = *reinterpret_cast<const uint4 *>(p);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(w[i] << 16);
            o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
}
template <typename io_t>
__device__ __forceinline__ void ld4(const io_t *p, float (&o)[4]) {
    if constexpr (sizeof(io_t) == 4) {
        const float4 a = *reinterpret_cast<const float4 *>(p);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
    } else {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    }
}
__device__ __forceinline__ unsigned pk2(float a, float b) {
    return (unsigned)from_f32<bf16_t>(a).bits | ((unsigned)from_f32<bf16_t>(b).bits << 16);
}
template <typename io_t>
__device__ __forceinline__ void st4(io_t *p, const float (&v)[4]) {
    if constexpr (sizeof(io_t) == 4)
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else
        *reinterpret_cast<uint2 *>(p) = make_uint2(pk2(v[0], v[1]), pk2(v[2], v[3]));
}
template <typename io_t>
__device__ __forceinline__ void st8(io_t *p, const float (&v)[8]) {
    if constexpr (sizeof(io_t) == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        *reinterpret_cast<uint4 *>(p) = make_uint4(pk2(v[0], v[1]), pk2(v[2], v[3]), pk2(v[4], v[5]), pk2(v[6], v[7]));
    }
}

template <typename io_t>
__global__ __launch_bounds__(256) void maxpool3s2_fwd_v4_kernel(const io_t *__restrict__ x, io_t *__restrict__ out,
                                                                unsigned char *__restrict__ code, int H, int W, int OH,
                                                                unsigned total) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int OW = W >> 1, oq = OW >> 2;
    const int q = (int)(t % (unsigned)oq);
    const unsigned r = t / (unsigned)oq;
    const int oy = (int)(r % (unsigned)OH);
    const long plane = r / (unsigned)OH;
    const io_t *xp = x + plane * H * W + 8 * q;
    float v[3][9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int y = min(max(2 * oy - 1 + dy, 0), H - 1);       // (clamped: the row above the image is masked below)
        const io_t *row = xp + (long)y * W;
        float r8[8];
        ld8(row, r8);
        v[dy][0] = to_f32(row[q > 0 ? -1 : 0]);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[dy][1 + i] = r8[i];
    }
    float best[4];
    unsigned codes = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float bv = 0.f;
        int bc = 0;
        bool first = true;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            if (dy == 0 && oy == 0) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                if (j == 0 && dx == 0 && q == 0) continue;
                const float val = v[dy][2 * j + dx];
                if (first || val > bv || val != val) {
                    bv = val;
                    bc = dy * 3 + dx;
                    first = false;
                }
            }
        }
        best[j] = bv;
        codes |= (unsigned)bc << (8 * j);
    }
    const long o = ((plane * OH + oy) * OW) + 4 * q;
    st4(out + o, best);
    *reinterpret_cast<unsigned *>(code + o) = codes;
}

// Backward: thread = input rows 2r, 2r+1, columns 8q .. 8q+7; it needs gradient rows r, r+1 at columns 4q .. 4q+4.
template <typename io_t>
__global__ __launch_bounds__(256) void maxpool3s2_bwd_code_v8_kernel(const io_t *__restrict__ g,
                                                                     const unsigned char *__restrict__ code,
                                                                     io_t *dx, const io_t *addend, int H, int W,
                                                                     unsigned total) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int OH = H >> 1, OW = W >> 1, wq = W >> 3;
    const int q = (int)(t % (unsigned)wq);
    const unsigned rr = t / (unsigned)wq;
    const int r = (int)(rr % (unsigned)OH);
    const long plane = rr / (unsigned)OH;
    const bool row1 = r + 1 < OH, col4 = 4 * q + 4 < OW;
    float gv[2][5];
    int cv[2][5];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const long o = (plane * OH + min(r + a, OH - 1)) * OW + 4 * q;
        float g4[4];
        ld4(g + o, g4);
        const unsigned c4 = *reinterpret_cast<const unsigned *>(code + o);
        const float g1 = to_f32(g[o + (col4 ? 4 : 0)]);
        const int c1 = code[o + (col4 ? 4 : 0)];
        const bool ok = a == 0 || row1;
        gv[a][0] = g4[0]; gv[a][1] = g4[1]; gv[a][2] = g4[2]; gv[a][3] = g4[3]; gv[a][4] = g1;
#pragma unroll
        for (int i = 0; i < 4; ++i) cv[a][i] = ok ? (int)((c4 >> (8 * i)) & 255u) : 255;
        cv[a][4] = ok && col4 ? c1 : 255;
    }
    float top[8], bot[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int o0 = i >> 1, o1 = (i + 1) >> 1;        // windows left / right of the pixel (the same one for even i)
        const int dx0 = (i & 1) ? 2 : 1;                 // position of the pixel inside window o0; inside o1 (odd i): 0
        // row 2r: window row r, dy = 1
        float s = cv[0][o0] == 3 + dx0 ? gv[0][o0] : 0.f;
        if (i & 1) s += cv[0][o1] == 3 ? gv[0][o1] : 0.f;
        top[i] = s;
        // row 2r+1: window row r with dy = 2, then window row r + 1 with dy = 0
        s = cv[0][o0] == 6 + dx0 ? gv[0][o0] : 0.f;
        if (i & 1) s += cv[0][o1] == 6 ? gv[0][o1] : 0.f;
        s += cv[1][o0] == dx0 ? gv[1][o0] : 0.f;
        if (i & 1) s += cv[1][o1] == 0 ? gv[1][o1] : 0.f;
        bot[i] = s;
    }
    io_t *dst = dx + (plane * H + 2 * r) * W + 8 * q;
    if (addend) {
        const io_t *ad = addend + (plane * H + 2 * r) * W + 8 * q;
        float a8[8], b8[8];
        ld8(ad, a8);
        ld8(ad + W, b8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            top[i] += a8[i];
            bot[i] += b8[i];
        }
    }
    st8(dst, top);
    st8(dst + W, bot);
}

inline bool maxpool_vector_ok(const mmu_maxpool_params *p) {
    static const bool off = getenv("MMU_MAXPOOL_VEC") && atoi(getenv("MMU_MAXPOOL_VEC")) == 0;
    return !off && p->width % 8 == 0 && p->height % 2 == 0;
}

}  // namespace

extern "C" int mmu_maxpool3s2_fwd(const mmu_maxpool_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "maxpool3s2_fwd: null params");
    MMU_CHECK(p->planes > 0 && p->height > 0 && p->width > 0, "maxpool3s2_fwd: empty tensor");
    MMU_CHECK(p->out_height == (p->height - 1) / 2 + 1 && p->out_width == (p->width - 1) / 2 + 1,
              "maxpool3s2_fwd: output size must be that of kernel 3 / stride 2 / padding 1");
    MMU_CHECK(p->input && p->out && p->codes, "maxpool3s2_fwd: input, out, codes are required");
    const long total = (long)p->planes * p->out_height * p->out_width;
    MMU_CHECK(total < (1L << 40), "maxpool3s2_fwd: tensor too large");
    const bool xb = p->io_dtype == MMU_DTYPE_BF16;
    MMU_CHECK(xb || p->io_dtype == MMU_DTYPE_F32, "maxpool3s2_fwd: io_dtype must be float32 or bfloat16 (got %d)", p->io_dtype);
    const bool vec = maxpool_vector_ok(p) && total / 4 < (1L << 32) && ((uintptr_t)p->input & 15) == 0 &&
                     ((uintptr_t)p->out & (xb ? 7 : 15)) == 0 && ((uintptr_t)p->codes & 3) == 0;
    MMU_CHECK(!xb || vec, "maxpool3s2_fwd: bfloat16 maps need width %% 8 == 0, even height, 16-byte aligned input");
    if (vec) {
        const unsigned n = (unsigned)(total / 4);
        if (xb)
            maxpool3s2_fwd_v4_kernel<bf16_t><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(
                (const bf16_t *)p->input, (bf16_t *)p->out, p->codes, p->height, p->width, p->out_height, n);
        else
            maxpool3s2_fwd_v4_kernel<float><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(
                (const float *)p->input, (float *)p->out, p->codes, p->height, p->width, p->out_height, n);
        MMU_HIP_LAUNCH_CHECK("maxpool3s2_fwd");
        return 0;
    }
    maxpool3s2_fwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (const float *)p->input, (float *)p->out, p->codes, p->height, p->width, p->out_height, p->out_width, p->planes);
    MMU_HIP_LAUNCH_CHECK("maxpool3s2_fwd");
    return 0;
}

extern "C" int mmu_maxpool3s2_bwd_codes(const mmu_maxpool_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "maxpool3s2_bwd_codes: null params");
    MMU_CHECK(p->planes > 0 && p->height > 0 && p->width > 0, "maxpool3s2_bwd_codes: empty tensor");
    MMU_CHECK(p->out_height == (p->height - 1) / 2 + 1 && p->out_width == (p->width - 1) / 2 + 1,
              "maxpool3s2_bwd_codes: output size must be that of kernel 3 / stride 2 / padding 1");
    MMU_CHECK(p->dout && p->codes && p->dinput, "maxpool3s2_bwd_codes: dout, codes, dinput are required");
    MMU_CHECK(((uintptr_t)p->dinput & 15) == 0, "maxpool3s2_bwd_codes: dinput must be 16-byte aligned");
    const long total = (long)p->planes * p->height * ((p->width + 3) / 4);
    MMU_CHECK(total < (1L << 32), "maxpool3s2_bwd_codes: tensor too large");
    const bool xb = p->io_dtype == MMU_DTYPE_BF16;
    MMU_CHECK(xb || p->io_dtype == MMU_DTYPE_F32, "maxpool3s2_bwd_codes: io_dtype must be float32 or bfloat16 (got %d)", p->io_dtype);
    const bool vec = maxpool_vector_ok(p) && ((uintptr_t)p->dout & (xb ? 7 : 15)) == 0 && ((uintptr_t)p->codes & 3) == 0 &&
                     ((uintptr_t)p->dinput_addend & 15) == 0;
    MMU_CHECK(!xb || vec, "maxpool3s2_bwd_codes: bfloat16 maps need width %% 8 == 0, even height, aligned gradients");
    if (vec) {
        const unsigned n = (unsigned)((long)p->planes * (p->height / 2) * (p->width / 8));
        if (xb)
            maxpool3s2_bwd_code_v8_kernel<bf16_t><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(
                (const bf16_t *)p->dout, p->codes, (bf16_t *)p->dinput, (const bf16_t *)p->dinput_addend, p->height, p->width, n);
        else
            maxpool3s2_bwd_code_v8_kernel<float><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(
                (const float *)p->dout, p->codes, (float *)p->dinput, (const float *)p->dinput_addend, p->height, p->width, n);
        MMU_HIP_LAUNCH_CHECK("maxpool3s2_bwd_codes");
        return 0;
    }
    maxpool3s2_bwd_code_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (const float *)p->dout, p->codes, (float *)p->dinput, (const float *)p->dinput_addend, p->height, p->width,
        p->out_height, p->out_width, (unsigned)total);
    MMU_HIP_LAUNCH_CHECK("maxpool3s2_bwd_codes");
    return 0;
}

extern "C" int mmu_maxpool3s2_bwd(const mmu_maxpool_params *p, void *stream) {
    MMU_CHECK(p != nullptr, "maxpool3s2_bwd: null params");
    MMU_CHECK(p->planes > 0 && p->height > 0 && p->width > 0, "maxpool3s2_bwd: empty tensor");
    MMU_CHECK(p->out_height == (p->height - 1) / 2 + 1 && p->out_width == (p->width - 1) / 2 + 1,
              "maxpool3s2_bwd: output size must be that of kernel 3 / stride 2 / padding 1 (got %d x %d for %d x %d)",
              p->out_height, p->out_width, p->height, p->width);
    MMU_CHECK(p->dout && p->indices && p->dinput, "maxpool3s2_bwd: dout, indices, dinput are required");
    MMU_CHECK(((uintptr_t)p->dinput & 15) == 0, "maxpool3s2_bwd: dinput must be 16-byte aligned");
    const long total = (long)p->planes * p->height * ((p->width + 3) / 4);
    MMU_CHECK(total < (1L << 40), "maxpool3s2_bwd: tensor too large");
    MMU_CHECK(p->io_dtype == MMU_DTYPE_F32, "maxpool3s2_bwd: float32 only (bfloat16: mmu_maxpool3s2_bwd_codes)");
    maxpool3s2_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (const float *)p->dout, (const long *)p->indices, (float *)p->dinput, p->height, p->width, p->out_height, p->out_width,
        p->planes);
    MMU_HIP_LAUNCH_CHECK("maxpool3s2_bwd");
    return 0;
}
