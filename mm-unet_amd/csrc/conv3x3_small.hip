// conv3x3_small.hip -- 3x3 / stride 1 / pad 1 convolution with FEW output channels (CO <= 8) for gfx950:
// MMConv's offset_conv (Cin -> 2K = 6, src/UM_Net/MMUNet.py:46,250), 44 of them per MM-UNet forward.
//
// With 6 output channels there is nothing for a matrix core to chew on (an MFMA tile would be >80 % padding)
// and MIOpen's Winograd / implicit-GEMM kernels take 60 us forward and 140 us backward for
// [8,64,128,128] (5.3 ms per training step over the 44 layers).  These are direct kernels whose weights are
// wave-uniform scalars (the weight tensor is passed transposed, [Cin][3][3][CO], so a channel's CO*9
// weights are one contiguous scalar load) and whose activations stream through once:
//   fwd : thread = 4 consecutive pixels of a row; per input channel 9 loads (3 rows x (16 B + 2 halo)),
//         4*9*CO FMAs.
//   bwd data : thread = 2 pixels; loads the CO x 3 x 4 neighbourhood of dout ONCE, then per input channel
//         2*9*CO FMAs against scalar weights and one 8-byte store.
//   bwd weight/bias : wave = (input channel, 2048-pixel chunk); lane = pixel (coalesced), CO*9 register
//         accumulators, wave reduction at the end, one float atomic per (wave, weight).
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

// Every border case is handled by clamping the address and multiplying by a 0/1 mask: a conditional load
// compiles to an exec-mask branch with its own s_waitcnt, which serialises the nine neighbour loads (the
// first version of the weight-gradient kernel ran 8x slower than its instruction count for that reason).

// fwd.  grid (ceil(B*H*ceil(W/4) / 256), splits): split s handles input channels [s*cps, (s+1)*cps) and
// writes its partial sums to part[s] ([B][CO][H][W]); with one split `part` is the output itself.
// NATIVE: wt is the weight as the module holds it, [CO][Cin][3][3] (no transposed copy); else [Cin][3][3][CO]
template <int CO, typename in_t, bool NATIVE>
__global__ __launch_bounds__(256) void conv3x3s_fwd_kernel(const in_t *__restrict__ x, const float *__restrict__ wt,
                                                           const float *__restrict__ bias, float *__restrict__ part,
                                                           int B, int Cin, int H, int W, int cps) {
    const int wq = (W + 3) / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * H * wq) return;
    const int q = (int)(idx % wq);
    const long r = idx / wq;
    const int h = (int)(r % H), b = (int)(r / H);
    const int w0 = q * 4;
    const long HW = (long)H * W;
    const int c_lo = blockIdx.y * cps, c_hi = c_lo + cps < Cin ? c_lo + cps : Cin;
    float acc[CO][4];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const float bv = (bias && blockIdx.y == 0) ? bias[co] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[co][j] = bv;
    }
    // clamped neighbourhood: rows h-1..h+1, columns w0-1..w0+4
    int roff[3], coff[6];
    float rm[3], cm[6];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int hh = h + dy - 1;
        rm[dy] = (hh >= 0 && hh < H) ? 1.f : 0.f;
        roff[dy] = (hh < 0 ? 0 : (hh > H - 1 ? H - 1 : hh)) * W;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int ww = w0 + j - 1;
        cm[j] = (ww >= 0 && ww < W) ? 1.f : 0.f;
        coff[j] = ww < 0 ? 0 : (ww > W - 1 ? W - 1 : ww);
    }
    const bool vec = (W & 3) == 0;  // then columns w0..w0+3 are in range and 16-byte aligned
    const in_t *xp = x + ((long)b * Cin + c_lo) * HW;
    for (int ci = c_lo; ci < c_hi; ++ci, xp += HW) {
        float v[3][6];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const in_t *rp = xp + roff[dy];
            if (vec) {
                float c[4];
                load_k<in_t, 4, true>(rp + w0, 4, true, c);
                v[dy][1] = c[0] * rm[dy]; v[dy][2] = c[1] * rm[dy]; v[dy][3] = c[2] * rm[dy]; v[dy][4] = c[3] * rm[dy];
                v[dy][0] = to_f32(rp[coff[0]]) * (rm[dy] * cm[0]);
                v[dy][5] = to_f32(rp[coff[5]]) * (rm[dy] * cm[5]);
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) v[dy][j] = to_f32(rp[coff[j]]) * (rm[dy] * cm[j]);
            }
        }
        const float *wc = NATIVE ? wt + (long)ci * 9 : wt + (long)ci * 9 * CO;  // wave-uniform: scalar loads
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float wv = NATIVE ? wc[(long)co * Cin * 9 + dy * 3 + dx] : wc[(dy * 3 + dx) * CO + co];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[co][j] = fmaf(wv, v[dy][j + dx], acc[co][j]);
                }
    }
    float *op = part + ((long)blockIdx.y * B + b) * CO * HW + (long)h * W + w0;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        if (vec) {
            *reinterpret_cast<float4 *>(op + co * HW) = make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (w0 + j < W) op[co * HW + j] = acc[co][j];
        }
    }
}

// out = sum over splits of part[s]  (fixed order: reproducible)
__global__ __launch_bounds__(256) void conv3x3s_sum_splits_kernel(const float *__restrict__ part, float *__restrict__ out,
                                                                  long n, int splits) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = part[i];
    for (int k = 1; k < splits; ++k) s += part[(long)k * n + i];
    out[i] = s;
}

// The same for many splits of a small map (16 x 16 maps of 512 channels: 64 splits of 12,288 outputs -- one thread per
// output walked 64 dependent round trips, 25-38 us): 64 outputs x 4 split groups per workgroup, each group a strided
// serial sum with four loads in flight, the four group sums added in order.
__global__ __launch_bounds__(256) void conv3x3s_sum_splits_wide_kernel(const float *__restrict__ part,
                                                                       float *__restrict__ out, long n, int splits) {
    __shared__ float sums[4][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + o;
    float s = 0.f;
    if (i < n) {
        int k = g;
        for (; k + 12 < splits; k += 16) {
            const float v0 = part[(long)k * n + i], v1 = part[(long)(k + 4) * n + i];
            const float v2 = part[(long)(k + 8) * n + i], v3 = part[(long)(k + 12) * n + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < splits; k += 4) s += part[(long)k * n + i];
    }
    sums[g][o] = s;
    __syncthreads();
    if (g == 0 && i < n) out[i] = (sums[0][o] + sums[1][o]) + (sums[2][o] + sums[3][o]);
}

// dx[b,ci,y,x] = sum_co sum_{ky,kx} W[co][ci][ky][kx] * g[b,co,y-ky+1,x-kx+1]
// grid (ceil(B*H*ceil(W/2) / 256), channel slices): every slice re-reads the (small) dout neighbourhood
template <int CO, typename in_t, bool NATIVE>
__global__ __launch_bounds__(256) void conv3x3s_bwd_data_kernel(const float *__restrict__ g, const float *__restrict__ wt,
                                                                in_t *__restrict__ dx, int B, int Cin, int H, int W,
                                                                int cps, const in_t *__restrict__ addend) {
    // addend (optional, same shape as dx): dx = conv gradient + addend -- the gradient the OTHER consumer of the
    // convolution's input already produced (MMConv: the sampler), so that autograd has nothing left to add
    const int wq = (W + 1) / 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * H * wq) return;
    const int q = (int)(idx % wq);
    const long r = idx / wq;
    const int h = (int)(r % H), b = (int)(r / H);
    const int w0 = q * 2;
    const long HW = (long)H * W;
    const int c_lo = blockIdx.y * cps, c_hi = c_lo + cps < Cin ? c_lo + cps : Cin;
    // gv[co][dy][j]: g at row h+dy-1, col w0+j-1 (0 outside the image)
    float gv[CO][3][4];
    const float *gp = g + (long)b * CO * HW;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int hh = h + dy - 1;
        const float rmask = (hh >= 0 && hh < H) ? 1.f : 0.f;
        const int ro = (hh < 0 ? 0 : (hh > H - 1 ? H - 1 : hh)) * W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ww = w0 + j - 1;
            const float m = (ww >= 0 && ww < W) ? rmask : 0.f;
            const int off = ro + (ww < 0 ? 0 : (ww > W - 1 ? W - 1 : ww));
#pragma unroll
            for (int co = 0; co < CO; ++co) gv[co][dy][j] = gp[co * HW + off] * m;
        }
    }
    in_t *dp = dx + ((long)b * Cin + c_lo) * HW + (long)h * W + w0;
    const in_t *ap = addend ? addend + ((long)b * Cin + c_lo) * HW + (long)h * W + w0 : nullptr;
    const bool two = (W & 1) == 0;
    for (int ci = c_lo; ci < c_hi; ++ci, dp += HW) {
        const float *wc = NATIVE ? wt + (long)ci * 9 : wt + (long)ci * 9 * CO;
        float a0 = 0.f, a1 = 0.f;
        if (ap) {
            const in_t *aq = ap + (long)(ci - c_lo) * HW;
            if (two) {
                float t2[2];
                load_k<in_t, 2, true>(aq, 2, true, t2);
                a0 = t2[0];
                a1 = t2[1];
            } else {
                a0 = to_f32(aq[0]);
                a1 = w0 + 1 < W ? to_f32(aq[1]) : 0.f;
            }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float wv = NATIVE ? wc[(long)co * Cin * 9 + ky * 3 + kx] : wc[(ky * 3 + kx) * CO + co];
                    // pixel (h, w0 + p): g row h-ky+1 -> dy = 2-ky ; col w0+p-kx+1 -> j = p - kx + 2
                    a0 = fmaf(wv, gv[co][2 - ky][2 - kx], a0);
                    a1 = fmaf(wv, gv[co][2 - ky][3 - kx], a1);
                }
        if (two) {
            const float a01[2] = {a0, a1};
            store_k<in_t, 2, true>(dp, 2, true, a01);
        } else {
            dp[0] = from_f32<in_t>(a0);
            if (w0 + 1 < W) dp[1] = from_f32<in_t>(a1);
        }
    }
}

// bwd data, FOUR pixels per thread (W % 4 == 0).  PMC of the two-pixel kernel at [8,64,128,128]
// (profiles/r03_conv3x3s_waves_variant_negative.txt): half of the wave cycles are issue stalls -- its two accumulators
// are ONE packed register, i.e. a single dependent chain of 54 v_pk_fma_f32 per input channel -- and a thread moves 8
// bytes per channel.  Four pixels give two independent chains, 16-byte loads / stores and the 3 x 6 neighbourhood of dout
// amortised over twice the outputs.  Same grid convention with wq = W / 4.
template <int CO, typename in_t, bool NATIVE>
__global__ __launch_bounds__(256) void conv3x3s_bwd_data4_kernel(const float *__restrict__ g, const float *__restrict__ wt,
                                                                 in_t *__restrict__ dx, int B, int Cin, int H, int W,
                                                                 int cps, const in_t *__restrict__ addend) {
    const int wq = W / 4;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)B * H * wq) return;
    const int q = (int)(idx % (unsigned)wq);
    const unsigned r = idx / (unsigned)wq;
    const int h = (int)(r % (unsigned)H), b = (int)(r / (unsigned)H);
    const int w0 = q * 4;
    const long HW = (long)H * W;
    const int c_lo = blockIdx.y * cps, c_hi = c_lo + cps < Cin ? c_lo + cps : Cin;
    // gv[co][dy][j]: g at row h+dy-1, col w0+j-1 (0 outside the image); columns w0 .. w0+3 are one 16-byte load
    float gv[CO][3][6];
    const float *gp = g + (long)b * CO * HW;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int hh = h + dy - 1;
        const float rmask = (hh >= 0 && hh < H) ? 1.f : 0.f;
        const int ro = (hh < 0 ? 0 : (hh > H - 1 ? H - 1 : hh)) * W;
        const float ml = w0 > 0 ? rmask : 0.f, mr = w0 + 4 < W ? rmask : 0.f;
        const int ol = ro + (w0 > 0 ? w0 - 1 : 0), orr = ro + (w0 + 4 < W ? w0 + 4 : W - 1);
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const float4 c = *reinterpret_cast<const float4 *>(gp + co * HW + ro + w0);
            gv[co][dy][1] = c.x * rmask; gv[co][dy][2] = c.y * rmask; gv[co][dy][3] = c.z * rmask; gv[co][dy][4] = c.w * rmask;
            gv[co][dy][0] = gp[co * HW + ol] * ml;
            gv[co][dy][5] = gp[co * HW + orr] * mr;
        }
    }
    in_t *dp = dx + ((long)b * Cin + c_lo) * HW + (long)h * W + w0;
    const in_t *ap = addend ? addend + ((long)b * Cin + c_lo) * HW + (long)h * W + w0 : nullptr;
    for (int ci = c_lo; ci < c_hi; ++ci, dp += HW) {
        const float *wc = NATIVE ? wt + (long)ci * 9 : wt + (long)ci * 9 * CO;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (ap) load_k<in_t, 4, true>(ap + (long)(ci - c_lo) * HW, 4, true, a);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float wv = NATIVE ? wc[(long)co * Cin * 9 + ky * 3 + kx] : wc[(ky * 3 + kx) * CO + co];
                    // pixel (h, w0 + p): g row h-ky+1 -> dy = 2-ky ; col w0+p-kx+1 -> j = p - kx + 2
#pragma unroll
                    for (int pxl = 0; pxl < 4; ++pxl) a[pxl] = fmaf(wv, gv[co][2 - ky][pxl + 2 - kx], a[pxl]);
                }
        store_k<in_t, 4, true>(dp, 4, true, a);
    }
}

// dW[co][ci][ky][kx] += sum_{b,y,x} g[b,co,y,x] * x[b,ci,y+ky-1,x+kx-1] ;  dbias[co] += sum g
// grid (chunks of 64*gpl groups over B*H*ceil(W/4), ceil(Cin / 4)); block 256 = 4 waves = 4 input channels;
// lane = group of 4 consecutive pixels of a row (the 3 x 6 input neighbourhood is shared by the 4 pixels:
// 9 + 6 load instructions per 216 FMAs), CO*9 register accumulators, wave reduction + one atomic per weight.
template <int CO, typename in_t>
__global__ __launch_bounds__(256) void conv3x3s_bwd_weight_kernel(const in_t *__restrict__ x, const float *__restrict__ g,
                                                                  float *__restrict__ dW, float *__restrict__ dbias,
                                                                  int B, int Cin, int H, int W, int gpl) {
    const int lane = threadIdx.x & 63;
    const int ci = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ci >= Cin) return;
    const int wq = (W + 3) / 4;
    const long HW = (long)H * W, NG = (long)B * H * wq;
    const long base = (long)blockIdx.x * 64 * gpl;
    const bool vec = (W & 3) == 0;
    float acc[CO][9], gs[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        gs[co] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[co][t] = 0.f;
    }
#pragma unroll 2
    for (int it = 0; it < gpl; ++it) {
        long gidx = base + (long)it * 64 + lane;
        const float live = gidx < NG ? 1.f : 0.f;  // lanes past the end re-read the last group with weight 0
        gidx = gidx < NG ? gidx : NG - 1;
        const int q = (int)(gidx % wq);
        const long r = gidx / wq;
        const int h = (int)(r % H), b = (int)(r / H);
        const int w0 = q * 4;
        const in_t *xp = x + ((long)b * Cin + ci) * HW;
        float xv[3][6];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int hh = h + dy - 1;
            const float rmask = (hh >= 0 && hh < H) ? 1.f : 0.f;
            const in_t *rp = xp + (hh < 0 ? 0 : (hh > H - 1 ? H - 1 : hh)) * W;
            if (vec) {
                float c[4];
                load_k<in_t, 4, true>(rp + w0, 4, true, c);
                xv[dy][1] = c[0] * rmask; xv[dy][2] = c[1] * rmask; xv[dy][3] = c[2] * rmask; xv[dy][4] = c[3] * rmask;
                xv[dy][0] = to_f32(rp[w0 > 0 ? w0 - 1 : 0]) * (w0 > 0 ? rmask : 0.f);
                xv[dy][5] = to_f32(rp[w0 + 4 < W ? w0 + 4 : W - 1]) * (w0 + 4 < W ? rmask : 0.f);
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int ww = w0 + j - 1;
                    xv[dy][j] = to_f32(rp[ww < 0 ? 0 : (ww > W - 1 ? W - 1 : ww)]) * ((ww >= 0 && ww < W) ? rmask : 0.f);
                }
            }
        }
        const float *gp = g + (long)b * CO * HW + (long)h * W;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            float gq[4];
            if (vec) {
                const float4 c = *reinterpret_cast<const float4 *>(gp + co * HW + w0);
                gq[0] = c.x * live; gq[1] = c.y * live; gq[2] = c.z * live; gq[3] = c.w * live;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) gq[j] = gp[co * HW + (w0 + j < W ? w0 + j : W - 1)] * (w0 + j < W ? live : 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gs[co] += gq[j];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc[co][ky * 3 + kx] = fmaf(gq[j], xv[ky][j + kx], acc[co][ky * 3 + kx]);
            }
        }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float s = wave_sum(acc[co][t]);
            if (lane == 0) atomicAdd(dW + ((long)co * Cin + ci) * 9 + t, s);
        }
        if (dbias != nullptr && ci == 0) {
            const float s = wave_sum(gs[co]);
            if (lane == 0) atomicAdd(dbias + co, s);
        }
    }
}

// ---- weight gradient, row-walking form --------------------------------------------------------------------
// dW[co][ci][ky][kx] = sum_{b,h,w} g[b][co][h][w] x[b][ci][h+ky-1][w+kx-1] is a reduction over every pixel with
// CO*9 results per input channel: ~0.5 GFLOP and 36 MB of reads for [8, 64, 128, 128] -- a memory-bound op that
// MIOpen's implicit-GEMM kernel (with its NCHW->NHWC transposes and zero-fills) needs 107 us for.
// wave = one input channel (wave-uniform), lanes = (W/4 column groups) x (64 / (W/4) row strips); a lane walks
// down its strip with a rolling 3-row window: ONE 16-byte x load and CO 16-byte g loads per 4 pixels and
// CO*36 FMAs; the left / right halo pixels come from the neighbouring lanes (ds_bpermute), not from memory.
// No integer division anywhere (the first version spent its time in 64-bit div/mod).  Each wave writes its
// CO*10 partial sums (wave_sum4 batches) to part[block][ci][*]; a second kernel adds the blocks in fixed
// order: deterministic, no atomics, no zero-fill.
template <int CO, typename in_t>
__global__ __launch_bounds__(256) void conv3x3s_wgrad_rows_kernel(const in_t *__restrict__ x, const float *__restrict__ g,
                                                                  float *__restrict__ part, int Cin, int H, int W,
                                                                  int lwq, int rs) {
    constexpr int NV = CO * 10, NV4 = (NV + 3) & ~3;
    const int lane = threadIdx.x & 63;
    const int ci = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    if (ci >= Cin) return;
    const int b = blockIdx.z;
    const int wq = 1 << lwq, q = lane & (wq - 1), st = lane >> lwq, S = 64 >> lwq;
    const int h0 = (blockIdx.x * S + st) * rs;
    const long HW = (long)H * W;
    const in_t *xp = x + ((long)b * Cin + ci) * HW + 4 * q;
    const float *gp = g + (long)b * CO * HW + 4 * q;
    const float lm = q > 0 ? 1.f : 0.f, rm = q < wq - 1 ? 1.f : 0.f;
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
    float r0[6], r1[6], r2[6];
    auto load_row = [&](int hh, float(&r)[6]) {
        const float m = (hh >= 0 && hh < H) ? 1.f : 0.f;
        const int hc = hh < 0 ? 0 : (hh > H - 1 ? H - 1 : hh);
        float c[4];
        load_k<in_t, 4, true>(xp + (long)hc * W, 4, true, c);
        r[1] = c[0] * m; r[2] = c[1] * m; r[3] = c[2] * m; r[4] = c[3] * m;
        r[0] = __shfl_up(r[4], 1) * lm;
        r[5] = __shfl_down(r[1], 1) * rm;
    };
    load_row(h0 - 1, r0);
    load_row(h0, r1);
    for (int i = 0; i < rs; ++i) {
        const int h = h0 + i;
        load_row(h + 1, r2);
        const float gm = h < H ? 1.f : 0.f;
        const long go = (long)(h < H ? h : H - 1) * W;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const float4 c = *reinterpret_cast<const float4 *>(gp + co * HW + go);
            const float gq[4] = {c.x * gm, c.y * gm, c.z * gm, c.w * gm};
            v[CO * 9 + co] += (gq[0] + gq[1]) + (gq[2] + gq[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    v[co * 9 + kx] = fmaf(gq[j], r0[j + kx], v[co * 9 + kx]);
                    v[co * 9 + 3 + kx] = fmaf(gq[j], r1[j + kx], v[co * 9 + 3 + kx]);
                    v[co * 9 + 6 + kx] = fmaf(gq[j], r2[j + kx], v[co * 9 + 6 + kx]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            r0[j] = r1[j];
            r1[j] = r2[j];
        }
    }
    float *dst = part + (((long)blockIdx.z * gridDim.x + blockIdx.x) * Cin + ci) * NV4;
#pragma unroll
    for (int i = 0; i < NV4; i += 4) {
        const float r = wave_sum4_swap(v[i], i + 1 < NV ? v[i + 1] : 0.f, i + 2 < NV ? v[i + 2] : 0.f,
                                  i + 3 < NV ? v[i + 3] : 0.f);
        if (lane >= 12 && lane < 16) dst[i + lane - 12] = r;
    }
}

// 16 lanes per result stride over the blocks' partials (fixed order), row-local DPP sum at the end
template <int CO>
__global__ __launch_bounds__(256) void conv3x3s_wgrad_sum_kernel(const float *__restrict__ part, float *__restrict__ dW,
                                                                 float *__restrict__ dbias, int Cin, int nblk) {
    constexpr int NV = CO * 10, NV4 = (NV + 3) & ~3;
    const int sub = threadIdx.x & 15;
    int idx = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = idx < Cin * NV4;
    idx = live ? idx : Cin * NV4 - 1;
    float s = 0.f;
    for (int k = sub; k < nblk; k += 16) s += part[(long)k * Cin * NV4 + idx];
    s += dpp_mov<MMU_DPP_ROW_SHR(1), 0xf>(0.f, s);
    s += dpp_mov<MMU_DPP_ROW_SHR(2), 0xf>(0.f, s);
    s += dpp_mov<MMU_DPP_ROW_SHR(4), 0xf>(0.f, s);
    s += dpp_mov<MMU_DPP_ROW_SHR(8), 0xf>(0.f, s);   // lane 15 of each row of 16 holds the total
    const int ci = idx / NV4, e = idx - ci * NV4;
    if (!live || sub != 15 || e >= NV) return;
    if (e < CO * 9) {
        const int co = e / 9, t = e - co * 9;
        dW[((long)co * Cin + ci) * 9 + t] = s;
    } else if (ci == 0 && dbias != nullptr) {
        dbias[e - CO * 9] = s;
    }
}

// rows per strip / row blocks of the row-walking weight gradient; 0 = shape not covered (W/4 must be a power
// of two <= 64)
bool wgrad_rows_plan(int B, int Cin, int H, int W, int &lwq, int &rs, int &nrb) {
    if (W < 4 || (W & 3) != 0) return false;
    const int wq = W / 4;
    if (wq > 64 || (wq & (wq - 1)) != 0) return false;
    lwq = 0;
    while ((1 << lwq) < wq) ++lwq;
    const int S = 64 / wq;
    long want = 4096 / ((long)B * Cin);  // row blocks wanted for ~4096 waves (16,384 measured slower: each wave pays the
                                         // 15 four-value wave sums of its epilogue whatever its strip count)
    if (want < 1) want = 1;
    long r = H / (S * want);
    rs = r < 1 ? 1 : (r > 16 ? 16 : (int)r);
    nrb = (H + S * rs - 1) / (S * rs);
    return true;
}

// how many input-channel slices so that the chip sees >= ~8192 waves (8 per SIMD: the channel loop is a
// load -> FMA chain per iteration, latency hiding comes from other waves)
int channel_splits(long threads, int cin) {
    static const long target = []() { const char *e = getenv("MMU_CONV3X3S_SPLIT_THREADS"); return e ? atol(e) : 524288L; }();
    int s = 1;
    while (threads * s < target && cin / (s * 2) >= 4) s *= 2;
    return s;
}

int check(const mmu_conv3x3s_params *p, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->batch > 0 && p->in_channels > 0 && p->height > 0 && p->width > 0, "%s: empty tensor", name);
    MMU_CHECK(p->out_channels == 1 || p->out_channels == 2 || p->out_channels == 6 || p->out_channels == 8,
              "%s: out_channels must be 1, 2, 6 or 8 (got %d)", name, p->out_channels);
    MMU_CHECK((long)p->batch * p->in_channels * p->height * p->width < (1L << 40), "%s: tensor too large", name);
    return 0;
}

#define CO_DISPATCH_T(co, ...)                                        \
    switch (co) {                                                     \
        case 1: { constexpr int CO = 1; __VA_ARGS__ } break;          \
        case 2: { constexpr int CO = 2; __VA_ARGS__ } break;          \
        case 6: { constexpr int CO = 6; __VA_ARGS__ } break;          \
        default: { constexpr int CO = 8; __VA_ARGS__ } break;         \
    }
// CO = out_channels, in_t = element type of input / dinput (p->in_dtype)
#define CO_DISPATCH(co, ...)                                          \
    if (p->in_dtype == MMU_DTYPE_BF16) {                              \
        using in_t = bf16_t;                                          \
        CO_DISPATCH_T(co, __VA_ARGS__)                                \
    } else {                                                          \
        using in_t = float;                                           \
        CO_DISPATCH_T(co, __VA_ARGS__)                                \
    }

}  // namespace

extern "C" int mmu_conv3x3_small_fwd_splits(int batch, int in_channels, int height, int width) {
    return channel_splits((long)batch * height * ((width + 3) / 4), in_channels);
}

extern "C" int mmu_conv3x3_small_fwd(const mmu_conv3x3s_params *p, void *stream) {
    if (int r = check(p, "conv3x3_small_fwd")) return r;
    MMU_CHECK(p->input && p->weight_t && p->out, "conv3x3_small_fwd: input, weight_t, out are required");
    const long total = (long)p->batch * p->height * ((p->width + 3) / 4);
    const int splits = channel_splits(total, p->in_channels);
    MMU_CHECK(splits == 1 || p->workspace != nullptr,
              "conv3x3_small_fwd: workspace of mmu_conv3x3_small_fwd_splits() x out elements is required");
    const int cps = (p->in_channels + splits - 1) / splits;
    hipStream_t st = (hipStream_t)stream;
    float *part = splits == 1 ? p->out : p->workspace;
    dim3 grid((unsigned)((total + 255) / 256), splits);
    if (p->weight_native) {
        CO_DISPATCH(p->out_channels, conv3x3s_fwd_kernel<CO, in_t, true><<<grid, 256, 0, st>>>(
                                         (const in_t *)p->input, p->weight_t, p->bias, part, p->batch, p->in_channels,
                                         p->height, p->width, cps);)
    } else {
        CO_DISPATCH(p->out_channels, conv3x3s_fwd_kernel<CO, in_t, false><<<grid, 256, 0, st>>>(
                                         (const in_t *)p->input, p->weight_t, p->bias, part, p->batch, p->in_channels,
                                         p->height, p->width, cps);)
    }
    MMU_HIP_LAUNCH_CHECK("conv3x3_small_fwd");
    if (splits > 1) {
        const long n = (long)p->batch * p->out_channels * p->height * p->width;
        if (splits >= 8 && n <= (1L << 21))
            conv3x3s_sum_splits_wide_kernel<<<(unsigned)((n + 63) / 64), 256, 0, st>>>(part, p->out, n, splits);
        else
            conv3x3s_sum_splits_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(part, p->out, n, splits);
        MMU_HIP_LAUNCH_CHECK("conv3x3_small_fwd(sum)");
    }
    return 0;
}

extern "C" size_t mmu_conv3x3_small_wgrad_workspace_floats(int batch, int in_channels, int out_channels, int height,
                                                           int width) {
    int lwq = 0, rs = 0, nrb = 0;
    if (batch <= 0 || in_channels <= 0 || !wgrad_rows_plan(batch, in_channels, height, width, lwq, rs, nrb)) return 0;
    return (size_t)nrb * batch * in_channels * ((out_channels * 10 + 3) & ~3);
}

int mmu_offset_dgrad_mfma_try(const mmu_conv3x3s_params *p, hipStream_t st);   // offset_conv_mfma.hip

extern "C" int mmu_conv3x3_small_bwd(const mmu_conv3x3s_params *p, void *stream) {
    if (int r = check(p, "conv3x3_small_bwd")) return r;
    MMU_CHECK(p->dout && p->weight_t, "conv3x3_small_bwd: dout and weight_t are required");
    hipStream_t st = (hipStream_t)stream;
    // the input gradient on the matrix cores where the shape allows (offset_conv_mfma.hip): 1 = done, 0 = not covered
    const int on_mfma = p->dinput ? mmu_offset_dgrad_mfma_try(p, st) : 0;
    if (on_mfma == 2) return 1;
    if (p->dinput && on_mfma == 0) {
        static const bool four_off = []() { const char *e = getenv("MMU_CONV3X3S_BWD4"); return e && e[0] == '0'; }();
        const size_t esz = p->in_dtype == MMU_DTYPE_BF16 ? 2 : 4;
        const bool four = !four_off && p->width % 4 == 0 && ((uintptr_t)p->dout & 15) == 0 &&
                          ((uintptr_t)p->dinput & (4 * esz - 1)) == 0 &&
                          (!p->dinput_addend || ((uintptr_t)p->dinput_addend & (4 * esz - 1)) == 0);
        const long total = (long)p->batch * p->height * (four ? p->width / 4 : (p->width + 1) / 2);
        // (the 54 loads of the dout neighbourhood are per thread: the four-pixel form keeps the slice count of the two-pixel one)
        const int splits = channel_splits(four ? 2 * total : total, p->in_channels);
        const int cps = (p->in_channels + splits - 1) / splits;
        dim3 grid((unsigned)((total + 255) / 256), splits);
        if (four && p->weight_native) {
            CO_DISPATCH(p->out_channels, conv3x3s_bwd_data4_kernel<CO, in_t, true><<<grid, 256, 0, st>>>(
                                             p->dout, p->weight_t, (in_t *)p->dinput, p->batch, p->in_channels, p->height,
                                             p->width, cps, (const in_t *)p->dinput_addend);)
        } else if (four) {
            CO_DISPATCH(p->out_channels, conv3x3s_bwd_data4_kernel<CO, in_t, false><<<grid, 256, 0, st>>>(
                                             p->dout, p->weight_t, (in_t *)p->dinput, p->batch, p->in_channels, p->height,
                                             p->width, cps, (const in_t *)p->dinput_addend);)
        } else if (p->weight_native) {
            CO_DISPATCH(p->out_channels, conv3x3s_bwd_data_kernel<CO, in_t, true><<<grid, 256, 0, st>>>(
                                             p->dout, p->weight_t, (in_t *)p->dinput, p->batch, p->in_channels, p->height,
                                             p->width, cps, (const in_t *)p->dinput_addend);)
        } else {
            CO_DISPATCH(p->out_channels, conv3x3s_bwd_data_kernel<CO, in_t, false><<<grid, 256, 0, st>>>(
                                             p->dout, p->weight_t, (in_t *)p->dinput, p->batch, p->in_channels, p->height,
                                             p->width, cps, (const in_t *)p->dinput_addend);)
        }
        MMU_HIP_LAUNCH_CHECK("conv3x3_small_bwd(data)");
    }
    int lwq = 0, rs = 0, nrb = 0;
    if (p->dweight && p->workspace && wgrad_rows_plan(p->batch, p->in_channels, p->height, p->width, lwq, rs, nrb)) {
        MMU_CHECK(p->input, "conv3x3_small_bwd: input is required for dweight");
        MMU_CHECK(((uintptr_t)p->input & 15) == 0 && ((uintptr_t)p->dout & 15) == 0,
                  "conv3x3_small_bwd: input and dout must be 16-byte aligned");

        dim3 grid(nrb, (p->in_channels + 3) / 4, p->batch);
        CO_DISPATCH(p->out_channels, {
            conv3x3s_wgrad_rows_kernel<CO, in_t><<<grid, 256, 0, st>>>((const in_t *)p->input, p->dout, p->workspace, p->in_channels,
                                                                 p->height, p->width, lwq, rs);
            MMU_HIP_LAUNCH_CHECK("conv3x3_small_bwd(weight rows)");
            constexpr int NV4 = (CO * 10 + 3) & ~3;
            const long job[8] = {1, (long)p->workspace, (long)p->dweight, (long)p->dbias, p->in_channels, (long)nrb * p->batch, CO, 0};
            if (!mmu_defer_job(job))   // (deferred_reduce.hip)
                conv3x3s_wgrad_sum_kernel<CO><<<(p->in_channels * NV4 + 15) / 16, 256, 0, st>>>(
                    p->workspace, p->dweight, p->dbias, p->in_channels, nrb * p->batch);
        })
        MMU_HIP_LAUNCH_CHECK("conv3x3_small_bwd(weight sum)");
    } else if (p->dweight) {
        MMU_CHECK(p->input, "conv3x3_small_bwd: input is required for dweight");
        hipError_t e = mmu_zero_async(p->dweight, (size_t)p->out_channels * p->in_channels * 9, st);
        if (e == hipSuccess && p->dbias) e = mmu_zero_async(p->dbias, p->out_channels, st);
        if (e != hipSuccess) return mmu_fail("conv3x3_small_bwd: memset: %s", hipGetErrorString(e));
        const long NG = (long)p->batch * p->height * ((p->width + 3) / 4);
        // groups per lane: ~4096 waves in total (the 54-value wave reduction + atomics at the end of a wave
        // cost as much as two iterations, so not more waves than the chip needs)
        long gpl_ = NG * p->in_channels / 64 / 4096;
        const int gpl = gpl_ < 2 ? 2 : (gpl_ > 32 ? 32 : (int)gpl_);
        dim3 grid((unsigned)((NG + 64L * gpl - 1) / (64L * gpl)), (p->in_channels + 3) / 4);
        CO_DISPATCH(p->out_channels, conv3x3s_bwd_weight_kernel<CO, in_t><<<grid, 256, 0, st>>>(
                                         (const in_t *)p->input, p->dout, p->dweight, p->dbias, p->batch, p->in_channels,
                                         p->height, p->width, gpl);)
        MMU_HIP_LAUNCH_CHECK("conv3x3_small_bwd(weight)");
    }
    return 0;
}
