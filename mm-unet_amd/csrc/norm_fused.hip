// norm_fused.hip -- GroupNorm [-> BatchNorm2d] [-> ReLU | tanh] as ONE normalisation, for gfx950.
//
// Every MMConv ends in GroupNorm (MMUNet.py:265 `self.gn`) and is followed, in the blocks that use it, by
// BatchNorm2d and usually ReLU (MMUNet.py:344-349, 357-359, 424-430, 436-452); its offset branch is
// GroupNorm -> tanh (:250).  As separate ATen/MIOpen ops that is 8 passes over the activation forward and
// 13 backward (GN, BN and ReLU each re-read and re-write it; 8 ms per training step).  Both normalisations
// are affine in x once the statistics are known, and both statistics follow from the per-(batch, channel)
// moments  s1 = sum_hw x,  s2 = sum_hw x^2 :
//     GroupNorm      y1 = a_bc x + d_bc        a = gamma_g r_bg,  d = beta_g - a mu_bg     (mu, r from s1, s2)
//     BatchNorm      y2 = A_bc x + D_bc        mean_c(y1), E_c(y1^2) in closed form from a, d, s1, s2
//     out = act(y2)
// so the forward is one moments pass + one apply pass, and the backward -- which needs only
// t1 = sum_hw g2, t2 = sum_hw g2 x  (g2 = dout * act'(y2)) per (batch, channel) -- one pass for those and one
// for  dx = c0_bc g2 + c1_bc x + c2_bc  (derivation in norm_fused.py).  The per-(b, c) algebra runs in a
// single small workgroup in double precision.
#include <type_traits>
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

__device__ __forceinline__ float act_fwd(float y, int act) {
    if (act == ACT_RELU) return fmaxf(y, 0.f);
    if (act == ACT_TANH) {
        const float e = __builtin_amdgcn_exp2f(-2.f * MMU_LOG2E * fabsf(y));  // e^{-2|y|}
        const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
        return y < 0.f ? -t : t;
    }
    return y;
}
// d act / d y expressed through y (pre-activation)
__device__ __forceinline__ float act_grad(float y, int act) {
    if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == ACT_TANH) {
        const float t = act_fwd(y, ACT_TANH);
        return 1.f - t * t;
    }
    return 1.f;
}

__device__ __forceinline__ float block_sum(float v, float *red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// moments per (b, c): grid (B*C), block 256 -- or 1024 for long rows when there are few of them (nf_moment_threads: 512
// rows of 16,384 floats on 256 threads each took 16.8 us = 2 TB/s, too few bytes in flight per CU).  BWD: t1 = sum g2, t2 = sum g2 x with g2 = dout * act'(A x + D)
// act_out (BWD, residual mode): the forward output relu(A x + D + residual); its sign is the ReLU mask
template <bool BWD, typename xin_t, typename act_t>
__global__ __launch_bounds__(1024) void nf_moments_kernel(const xin_t *__restrict__ x, const act_t *__restrict__ dout,
                                                         const float *__restrict__ A, const float *__restrict__ D,
                                                         float *__restrict__ m1, float *__restrict__ m2, int HW,
                                                         int act, const act_t *__restrict__ act_out,
                                                         const float *__restrict__ pre_bias, int C) {
    // pre_bias (forward only): the moments written are those of x + pre_bias[c]
    __shared__ float red[16];
    const long base = (long)blockIdx.x * HW;
    const float Av = BWD ? A[blockIdx.x] : 0.f, Dv = BWD ? D[blockIdx.x] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if ((HW & 3) == 0) {
        for (int i = threadIdx.x; i < HW / 4; i += blockDim.x) {
            float xv[4];
            load_k<xin_t, 4, true>(x + base + 4 * i, 4, true, xv);
            if (BWD) {
                float gv[4], ov[4] = {0.f, 0.f, 0.f, 0.f};
                load_k<act_t, 4, true>(dout + base + 4 * i, 4, true, gv);
                if (act_out) load_k<act_t, 4, true>(act_out + base + 4 * i, 4, true, ov);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float g2 = gv[j] * (act_out ? (ov[j] > 0.f ? 1.f : 0.f) : act_grad(fmaf(Av, xv[j], Dv), act));
                    s1 += g2;
                    s2 = fmaf(g2, xv[j], s2);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s1 += xv[j];
                    s2 = fmaf(xv[j], xv[j], s2);
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const float xv = to_f32(x[base + i]);
            if (BWD) {
                const float g2 = to_f32(dout[base + i]) * (act_out ? (to_f32(act_out[base + i]) > 0.f ? 1.f : 0.f)
                                                                   : act_grad(fmaf(Av, xv, Dv), act));
                s1 += g2;
                s2 = fmaf(g2, xv, s2);
            } else {
                s1 += xv;
                s2 = fmaf(xv, xv, s2);
            }
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        if (!BWD && pre_bias) {
            const double bv = pre_bias[blockIdx.x % C], d1 = s1, d2 = s2;
            s1 = (float)(d1 + bv * HW);
            s2 = (float)(d2 + 2 * bv * d1 + bv * bv * HW);
        }
        m1[blockIdx.x] = s1;
        m2[blockIdx.x] = s2;
    }
}

struct FinArgs {
    int B, C, G, HW, has_bn, training, has_gn;
    float eps_g, eps_b, momentum;
    float *s1, *s2;                // [B*C] moments of x (+ pre_bias: rewritten in place by the forward finalize)
    const float *pre_bias;         // [C] or NULL: the normalisation sees x + pre_bias[c]
    float *dpre_bias;              // [C] or NULL
    const float *gn_w, *gn_b;      // [C] or NULL (1 / 0)
    const float *bn_w, *bn_b;      // [C] or NULL
    float *run_mean, *run_var;     // [C] (updated when training) or NULL
    float *mu, *rstd;              // [B*G] GroupNorm statistics (saved)
    float *bmean, *brstd;          // [C] BatchNorm statistics used (saved)
    float *A, *D;                  // [B*C] out = act(A x + D)
    // backward
    float *t1, *t2;                // [B*C]
    float *c0, *c1, *c2;           // [B*C] dx = c0 g2 + c1 x + c2
    float *dgn_w, *dgn_b, *dbn_w, *dbn_b;  // [C] parameter gradients (may be NULL)
    float *scratch;                // [2*B*C + 2*B*G + 3*C] floats of workspace for the backward algebra
};

// Forward constants of ONE (batch, channel) row, computed by the 256 threads of the apply block that owns the row
// (round 2: this algebra was a single-workgroup kernel between the two passes -- 7 us of launch chain per call for
// microseconds of work; every apply block now derives its own A, D from the moments: ~B * C/G * 2 loads and a few
// hundred double operations, once per row of >= 1,024 elements).  p.s1 / p.s2 already hold the moments of
// x + pre_bias (nf_moments_kernel).  Results every block needs anyway are also what the backward reads later: the row's
// block writes A, D; the first channel of a group writes the group statistics of its batch item; batch item 0 writes
// the channel's BatchNorm statistics and updates the running ones.
__device__ void nf_fwd_constants(const FinArgs &p, int bc, bool writer, float &A_out, float &D_out) {
    // writer: exactly one block per row stores the saved statistics / updates the running ones
    __shared__ double sh_s1[64], sh_s2[64], sh_mu[64], sh_rs[64];   // [b'][channel of the group] / [b']
    __shared__ float sh_ad[2];
    const int cpg = p.C / p.G;
    const int b = bc / p.C, c = bc - b * p.C, g = c / cpg;
    const double n = (double)cpg * p.HW, N = (double)p.B * p.HW;
    const int tid = threadIdx.x;
    // the BatchNorm stage needs the group statistics of every batch item; batch-local if there is no BatchNorm (or
    // it runs on running statistics)
    const bool all_b = p.has_bn && p.training;
    const int b_lo = all_b ? 0 : b, nb = all_b ? p.B : 1;
    // (B * cpg <= 64 and B <= 64 are checked by the host; larger problems keep the finalize kernel)
    if (tid < nb * cpg) {
        const int bb = b_lo + tid / cpg, cc = g * cpg + tid % cpg;
        sh_s1[tid] = p.s1[bb * p.C + cc];
        sh_s2[tid] = p.s2[bb * p.C + cc];
    }
    __syncthreads();
    if (tid < nb) {
        double mu = 0, rs = 1;
        if (p.has_gn) {
            double a1 = 0, a2 = 0;
            for (int j = 0; j < cpg; ++j) {
                a1 += sh_s1[tid * cpg + j];
                a2 += sh_s2[tid * cpg + j];
            }
            mu = a1 / n;
            double var = a2 / n - mu * mu;
            var = var < 0 ? 0 : var;
            rs = 1.0 / sqrt(var + (double)p.eps_g);
        }
        sh_mu[tid] = (double)(float)mu;   // the float values the backward will read back
        sh_rs[tid] = (double)(float)rs;
    }
    __syncthreads();
    if (tid == 0) {
        const int jc = c - g * cpg, ib = b - b_lo;
        const double gw = p.gn_w ? p.gn_w[c] : 1.0, gb = p.gn_b ? p.gn_b[c] : 0.0;
        double a = gw * sh_rs[ib], d = gb - a * sh_mu[ib];
        if (writer && jc == 0) {   // the group statistics of this batch item (saved for the backward)
            p.mu[b * p.G + g] = (float)sh_mu[ib];
            p.rstd[b * p.G + g] = (float)sh_rs[ib];
        }
        if (p.has_bn) {
            double m, rb;
            if (p.training) {
                double sm = 0, sq = 0;
                for (int bb = 0; bb < p.B; ++bb) {
                    const double ab = gw * sh_rs[bb], db = gb - ab * sh_mu[bb];
                    const double s1 = sh_s1[bb * cpg + jc], s2 = sh_s2[bb * cpg + jc];
                    sm += ab * s1 + db * p.HW;
                    sq += ab * ab * s2 + 2 * ab * db * s1 + db * db * p.HW;
                }
                m = sm / N;
                double v = sq / N - m * m;
                v = v < 0 ? 0 : v;
                rb = 1.0 / sqrt(v + (double)p.eps_b);
                if (writer && b == 0 && p.run_mean) {
                    p.run_mean[c] = (float)((1.0 - p.momentum) * p.run_mean[c] + p.momentum * m);
                    p.run_var[c] = (float)((1.0 - p.momentum) * p.run_var[c] + p.momentum * v * (N > 1 ? N / (N - 1) : 1.0));
                }
            } else {
                m = p.run_mean[c];
                rb = 1.0 / sqrt((double)p.run_var[c] + (double)p.eps_b);
            }
            // the float copies are what the backward reads: use the same rounded values here
            const float mf = (float)m, rf = (float)rb;
            if (writer && b == 0) {
                p.bmean[c] = mf;
                p.brstd[c] = rf;
            }
            const double k = (p.bn_w ? p.bn_w[c] : 1.0) * rf;
            d = (d - mf) * k + (p.bn_b ? p.bn_b[c] : 0.0);
            a = a * k;
        }
        if (p.pre_bias) d += a * p.pre_bias[c];  // act(A (x + bias) + D) as act(A x + D')
        if (writer) {
            p.A[bc] = (float)a;
            p.D[bc] = (float)d;
        }
        sh_ad[0] = (float)a;
        sh_ad[1] = (float)d;
    }
    __syncthreads();
    A_out = sh_ad[0];
    D_out = sh_ad[1];
}

// one workgroup; all per-(b,c) algebra in double (forward: only for shapes nf_fwd_constants does not take)
__global__ __launch_bounds__(1024) void nf_finalize_fwd_kernel(FinArgs p) {
    const int cpg = p.C / p.G;
    const double n = (double)cpg * p.HW, N = (double)p.B * p.HW;
    if (p.pre_bias) {  // moments of x + bias from the moments of x
        for (int bc = threadIdx.x; bc < p.B * p.C; bc += blockDim.x) {
            const double bv = p.pre_bias[bc % p.C], s1 = p.s1[bc], s2 = p.s2[bc];
            p.s1[bc] = (float)(s1 + bv * p.HW);
            p.s2[bc] = (float)(s2 + 2 * bv * s1 + bv * bv * p.HW);
        }
        __syncthreads();
    }
    for (int bg = threadIdx.x; bg < p.B * p.G; bg += blockDim.x) {
        if (!p.has_gn) {  // BatchNorm alone: the "group" stage is the identity
            p.mu[bg] = 0.f;
            p.rstd[bg] = 1.f;
            continue;
        }
        const int b = bg / p.G, g = bg - b * p.G;
        double a1 = 0, a2 = 0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            a1 += p.s1[b * p.C + c];
            a2 += p.s2[b * p.C + c];
        }
        const double mu = a1 / n;
        double var = a2 / n - mu * mu;
        var = var < 0 ? 0 : var;
        p.mu[bg] = (float)mu;
        p.rstd[bg] = (float)(1.0 / sqrt(var + (double)p.eps_g));
    }
    __syncthreads();
    if (p.has_bn) {
        for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
            double m, rb;
            if (p.training) {
                const int g = c / cpg;
                const double gw = p.gn_w ? p.gn_w[c] : 1.0, gb = p.gn_b ? p.gn_b[c] : 0.0;
                double sm = 0, sq = 0;
                for (int b = 0; b < p.B; ++b) {
                    const double a = gw * p.rstd[b * p.G + g], d = gb - a * p.mu[b * p.G + g];
                    const double s1 = p.s1[b * p.C + c], s2 = p.s2[b * p.C + c];
                    sm += a * s1 + d * p.HW;
                    sq += a * a * s2 + 2 * a * d * s1 + d * d * p.HW;
                }
                m = sm / N;
                double v = sq / N - m * m;
                v = v < 0 ? 0 : v;
                rb = 1.0 / sqrt(v + (double)p.eps_b);
                if (p.run_mean) {
                    p.run_mean[c] = (float)((1.0 - p.momentum) * p.run_mean[c] + p.momentum * m);
                    p.run_var[c] = (float)((1.0 - p.momentum) * p.run_var[c] + p.momentum * v * (N > 1 ? N / (N - 1) : 1.0));
                }
            } else {
                m = p.run_mean[c];
                rb = 1.0 / sqrt((double)p.run_var[c] + (double)p.eps_b);
            }
            p.bmean[c] = (float)m;
            p.brstd[c] = (float)rb;
        }
    }
    __syncthreads();
    for (int bc = threadIdx.x; bc < p.B * p.C; bc += blockDim.x) {
        const int b = bc / p.C, c = bc - b * p.C, g = c / cpg;
        const double gw = p.gn_w ? p.gn_w[c] : 1.0, gb = p.gn_b ? p.gn_b[c] : 0.0;
        double a = gw * p.rstd[b * p.G + g], d = gb - a * p.mu[b * p.G + g];
        if (p.has_bn) {
            const double k = (p.bn_w ? p.bn_w[c] : 1.0) * p.brstd[c];
            d = (d - p.bmean[c]) * k + (p.bn_b ? p.bn_b[c] : 0.0);
            a = a * k;
        }
        if (p.pre_bias) d += a * p.pre_bias[c];  // act(A (x + bias) + D) as act(A x + D')
        p.A[bc] = (float)a;
        p.D[bc] = (float)d;
    }
}

__global__ __launch_bounds__(1024) void nf_finalize_bwd_kernel(FinArgs p) {
    const int cpg = p.C / p.G;
    const double n = (double)cpg * p.HW, N = (double)p.B * p.HW, HW = p.HW;
    // workspace: mexact[C] (double) | u1[B*C] u2[B*C] M1[B*G] M2[B*G] kk[C] ee[C] ff[C]
    // mexact: the batch mean of y1 recomputed in double from the moments.  The float copy saved by the forward
    // is good enough for the output, but here sum_b (p s1 + q HW) must cancel to zero (BatchNorm removes any
    // per-channel shift, so d(gn bias) is analytically 0); with the rounded mean it leaves 1e-5 of noise.
    if (p.pre_bias) {  // sum_hw g2 (x + bias)
        for (int bc = threadIdx.x; bc < p.B * p.C; bc += blockDim.x) p.t2[bc] += p.pre_bias[bc % p.C] * p.t1[bc];
        __syncthreads();
    }
    double *mexact = reinterpret_cast<double *>(p.scratch);
    float *u1 = p.scratch + 2 * p.C, *u2 = u1 + p.B * p.C, *M1 = u2 + p.B * p.C, *M2 = M1 + p.B * p.G;
    float *kk = M2 + p.B * p.G, *ee = kk + p.C, *ff = ee + p.C;
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        double k = 1, e = 0, f = 0;
        if (p.has_bn) {
            const int g = c / cpg;
            const double gw = p.gn_w ? p.gn_w[c] : 1.0, gb = p.gn_b ? p.gn_b[c] : 0.0;
            const double rb = p.brstd[c];
            double m = p.bmean[c];
            if (p.training) {
                double sm = 0;
                for (int b = 0; b < p.B; ++b) {
                    const double a = gw * p.rstd[b * p.G + g], d = gb - a * p.mu[b * p.G + g];
                    sm += a * p.s1[b * p.C + c] + d * p.HW;
                }
                m = sm / N;
            }
            mexact[c] = m;
            double db = 0, dg = 0;
            for (int b = 0; b < p.B; ++b) {
                const double a = gw * p.rstd[b * p.G + g], d = gb - a * p.mu[b * p.G + g];
                const double pq = a * rb, qq = (d - m) * rb;
                db += p.t1[b * p.C + c];
                dg += pq * p.t2[b * p.C + c] + qq * p.t1[b * p.C + c];
            }
            if (p.dbn_w) p.dbn_w[c] = (float)dg;
            if (p.dbn_b) p.dbn_b[c] = (float)db;
            k = (p.bn_w ? p.bn_w[c] : 1.0) * rb;
            if (p.training) {
                e = db / N;
                f = dg / N;
            }
        }
        kk[c] = (float)k;
        ee[c] = (float)e;
        ff[c] = (float)f;
    }
    __syncthreads();
    for (int bc = threadIdx.x; bc < p.B * p.C; bc += blockDim.x) {
        const int b = bc / p.C, c = bc - b * p.C, g = c / cpg;
        const double gw = p.gn_w ? p.gn_w[c] : 1.0, gb = p.gn_b ? p.gn_b[c] : 0.0;
        const double a = gw * p.rstd[b * p.G + g], d = gb - a * p.mu[b * p.G + g];
        double pq = 0, qq = 0;
        if (p.has_bn) {
            pq = a * p.brstd[c];
            qq = (d - mexact[c]) * p.brstd[c];
        }
        const double s1 = p.s1[bc], s2 = p.s2[bc], t1 = p.t1[bc], t2 = p.t2[bc];
        const double k = kk[c], e = ee[c], f = ff[c];
        u1[bc] = (float)(k * (t1 - e * HW - f * (pq * s1 + qq * HW)));
        u2[bc] = (float)(k * (t2 - e * s1 - f * (pq * s2 + qq * s1)));
    }
    __syncthreads();
    for (int bg = threadIdx.x; bg < p.B * p.G; bg += blockDim.x) {
        const int b = bg / p.G, g = bg - b * p.G;
        const double mu = p.mu[bg], r = p.rstd[bg];
        double a1 = 0, a2 = 0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const double gw = p.gn_w ? p.gn_w[c] : 1.0;
            a1 += gw * u1[b * p.C + c];
            a2 += gw * r * (u2[b * p.C + c] - mu * u1[b * p.C + c]);
        }
        M1[bg] = p.has_gn ? (float)(a1 / n) : 0.f;
        M2[bg] = p.has_gn ? (float)(a2 / n) : 0.f;
    }
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        const int g = c / cpg;
        double dw = 0, dbv = 0;
        for (int b = 0; b < p.B; ++b) {
            const double mu = p.mu[b * p.G + g], r = p.rstd[b * p.G + g];
            dw += r * (u2[b * p.C + c] - mu * u1[b * p.C + c]);
            dbv += u1[b * p.C + c];
        }
        if (p.dgn_w) p.dgn_w[c] = (float)dw;
        if (p.dgn_b) p.dgn_b[c] = (float)dbv;
    }
    __syncthreads();
    for (int bc = threadIdx.x; bc < p.B * p.C; bc += blockDim.x) {
        const int b = bc / p.C, c = bc - b * p.C, g = c / cpg;
        const double gw = p.gn_w ? p.gn_w[c] : 1.0, gb = p.gn_b ? p.gn_b[c] : 0.0;
        const double mu = p.mu[b * p.G + g], r = p.rstd[b * p.G + g];
        const double a = gw * r, d = gb - a * mu;
        double pq = 0, qq = 0;
        if (p.has_bn) {
            pq = a * p.brstd[c];
            qq = (d - mexact[c]) * p.brstd[c];
        }
        const double k = kk[c], e = ee[c], f = ff[c];
        const double m1 = M1[b * p.G + g], m2 = M2[b * p.G + g];
        const double rgk = r * gw * k;
        const double k0 = rgk, k1 = -rgk * f * pq - r * r * m2, k2 = -rgk * e - rgk * f * qq - r * m1 + r * r * mu * m2;
        p.c0[bc] = (float)k0;
        p.c1[bc] = (float)k1;
        // in terms of the raw x: c1 (x + bias) + c2
        p.c2[bc] = (float)(k2 + (p.pre_bias ? k1 * p.pre_bias[c] : 0.0));
        if (p.dpre_bias) u1[bc] = (float)(k0 * p.t1[bc] + k1 * p.s1[bc] + k2 * HW);  // sum_hw dx of this (b, c)
    }
    if (p.dpre_bias) {
        __syncthreads();
        for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
            double sacc = 0;
            for (int b = 0; b < p.B; ++b) sacc += u1[b * p.C + c];
            p.dpre_bias[c] = (float)sacc;
        }
    }
}

// out = act(A x + D)  |  dx = c0 * dout * act'(A x + D) + c1 x + c2 ;  grid (ceil(HW/1024), B*C), block 256
// xin_t: element type of x and of dx; act_t: of out, the residual, dout, the saved forward output, d residual
template <bool BWD, typename xin_t, typename act_t>
__global__ __launch_bounds__(256) void nf_apply_kernel(const xin_t *__restrict__ x, const act_t *__restrict__ dout,
                                                       const float *__restrict__ A, const float *__restrict__ D,
                                                       const float *__restrict__ c0, const float *__restrict__ c1,
                                                       const float *__restrict__ c2, void *__restrict__ out_, int HW,
                                                       int act, int out_cb_batch, const act_t *__restrict__ res,
                                                       act_t *__restrict__ dres) {
    // res: FWD = residual added before the activation; BWD = the forward output (ReLU mask).  dres: BWD only.
    using out_t = typename std::conditional<BWD, xin_t, act_t>::type;
    out_t *__restrict__ out = static_cast<out_t *>(out_);
    const int bc = blockIdx.y;
    const float Av = A[bc], Dv = D[bc];
    const float k0 = BWD ? c0[bc] : 0.f, k1 = BWD ? c1[bc] : 0.f, k2 = BWD ? c2[bc] : 0.f;
    const long base = (long)bc * HW;
    // out_cb_batch = B > 0: the result is written channel-major, [C][B][HW] (dinput for a tokens-last consumer)
    long obase = base;
    if (out_cb_batch > 0) {
        const int C = gridDim.y / out_cb_batch, b = bc / C, c = bc - b * C;
        obase = ((long)c * out_cb_batch + b) * HW;
    }
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= HW) return;
    if ((HW & 3) == 0) {
        float xv[4], o[4], rv[4] = {0.f, 0.f, 0.f, 0.f};
        load_k<xin_t, 4, true>(x + base + i, 4, true, xv);
        if (res) load_k<act_t, 4, true>(res + base + i, 4, true, rv);
        if (BWD) {
            float gv[4], g2[4];
            load_k<act_t, 4, true>(dout + base + i, 4, true, gv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g2[j] = gv[j] * (res ? (rv[j] > 0.f ? 1.f : 0.f) : act_grad(fmaf(Av, xv[j], Dv), act));
                o[j] = fmaf(k0, g2[j], fmaf(k1, xv[j], k2));
            }
            if (dres) store_k<act_t, 4, true>(dres + base + i, 4, true, g2);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = act_fwd(fmaf(Av, xv[j], Dv) + rv[j], act);
        }
        store_k<out_t, 4, true>(out + obase + i, 4, true, o);
    } else {
        for (int j = 0; j < 4 && i + j < HW; ++j) {
            const float xv = to_f32(x[base + i + j]);
            const float rr = res ? to_f32(res[base + i + j]) : 0.f;
            if (BWD) {
                const float g2 = to_f32(dout[base + i + j]) * (res ? (rr > 0.f ? 1.f : 0.f) : act_grad(fmaf(Av, xv, Dv), act));
                out[obase + i + j] = from_f32<out_t>(fmaf(k0, g2, fmaf(k1, xv, k2)));
                if (dres) dres[base + i + j] = from_f32<act_t>(g2);
            } else {
                out[obase + i + j] = from_f32<out_t>(act_fwd(fmaf(Av, xv, Dv) + rr, act));
            }
        }
    }
}

// Backward constants of ONE (batch, channel) row, by the apply block that owns it: the algebra of
// nf_finalize_bwd_kernel restricted to the row's group -- every quantity of (b, c) depends only on the moments of the
// B x C/G rows of its group.  t2 is shifted by pre_bias here (the finalize kernel did it in place).  The row's block of
// batch item 0 writes the channel's parameter gradients.  Intermediates stay in double (the kernel rounded some of
// them to float on their way through global memory).
__device__ void nf_bwd_constants(const FinArgs &p, int bc, bool writer, float &c0_out, float &c1_out, float &c2_out) {
    __shared__ double T1[64], T2[64], S1[64], S2[64], U1[64], U2[64];   // [b'][channel of the group]
    __shared__ double MU[64], RS[64], M1s[64], M2s[64];                  // [b']
    __shared__ double KK[64], EE[64], FF[64], MX[64], GW[64], GB[64], RB[64];   // [channel of the group]
    __shared__ float sh_c[3];
    const int cpg = p.C / p.G;
    const int b = bc / p.C, c = bc - b * p.C, g = c / cpg, jc0 = c - g * cpg;
    const double n = (double)cpg * p.HW, N = (double)p.B * p.HW, HW = p.HW;
    const int tid = threadIdx.x;
    if (tid < p.B * cpg) {
        const int bb = tid / cpg, cc = g * cpg + tid % cpg;
        const double t1 = p.t1[bb * p.C + cc];
        T1[tid] = t1;
        T2[tid] = (double)p.t2[bb * p.C + cc] + (p.pre_bias ? (double)p.pre_bias[cc] * t1 : 0.0);   // sum_hw g2 (x + bias)
        S1[tid] = p.s1[bb * p.C + cc];
        S2[tid] = p.s2[bb * p.C + cc];
    }
    if (tid >= 64 && tid < 64 + p.B) {
        MU[tid - 64] = p.mu[(tid - 64) * p.G + g];
        RS[tid - 64] = p.rstd[(tid - 64) * p.G + g];
    }
    if (tid >= 128 && tid < 128 + cpg) {
        const int cc = g * cpg + tid - 128;
        GW[tid - 128] = p.gn_w ? p.gn_w[cc] : 1.0;
        GB[tid - 128] = p.gn_b ? p.gn_b[cc] : 0.0;
        RB[tid - 128] = p.has_bn ? p.brstd[cc] : 1.0;
    }
    __syncthreads();
    if (tid < cpg) {   // per channel of the group: exact batch mean, d(bn weight / bias), k, e, f
        const int cc = g * cpg + tid;
        double k = 1, e = 0, f = 0, m = 0;
        if (p.has_bn) {
            const double gw = GW[tid], gb = GB[tid], rb = RB[tid];
            m = p.bmean[cc];
            if (p.training) {
                double sm = 0;
                for (int bb = 0; bb < p.B; ++bb) {
                    const double a = gw * RS[bb], d = gb - a * MU[bb];
                    sm += a * S1[bb * cpg + tid] + d * p.HW;
                }
                m = sm / N;
            }
            double db = 0, dg = 0;
            for (int bb = 0; bb < p.B; ++bb) {
                const double a = gw * RS[bb], d = gb - a * MU[bb];
                const double pq = a * rb, qq = (d - m) * rb;
                db += T1[bb * cpg + tid];
                dg += pq * T2[bb * cpg + tid] + qq * T1[bb * cpg + tid];
            }
            if (writer && b == 0 && tid == jc0) {
                if (p.dbn_w) p.dbn_w[cc] = (float)dg;
                if (p.dbn_b) p.dbn_b[cc] = (float)db;
            }
            k = (p.bn_w ? p.bn_w[cc] : 1.0) * rb;
            if (p.training) {
                e = db / N;
                f = dg / N;
            }
        }
        KK[tid] = k; EE[tid] = e; FF[tid] = f; MX[tid] = m;
    }
    __syncthreads();
    if (tid < p.B * cpg) {   // u1 = sum_hw g1, u2 = sum_hw g1 x of every row of the group
        const int bb = tid / cpg, j = tid % cpg;
        const double a = GW[j] * RS[bb], d = GB[j] - a * MU[bb];
        double pq = 0, qq = 0;
        if (p.has_bn) {
            pq = a * RB[j];
            qq = (d - MX[j]) * RB[j];
        }
        const double k = KK[j], e = EE[j], f = FF[j];
        U1[tid] = k * (T1[tid] - e * HW - f * (pq * S1[tid] + qq * HW));
        U2[tid] = k * (T2[tid] - e * S1[tid] - f * (pq * S2[tid] + qq * S1[tid]));
    }
    __syncthreads();
    if (tid < p.B) {   // group means of gamma g1 and gamma g1 xhat, per batch item
        double a1 = 0, a2 = 0;
        for (int j = 0; j < cpg; ++j) {
            a1 += GW[j] * U1[tid * cpg + j];
            a2 += GW[j] * RS[tid] * (U2[tid * cpg + j] - MU[tid] * U1[tid * cpg + j]);
        }
        M1s[tid] = p.has_gn ? a1 / n : 0.0;
        M2s[tid] = p.has_gn ? a2 / n : 0.0;
    }
    __syncthreads();
    if (tid == 0) {
        const double gw = GW[jc0], gb = GB[jc0], k = KK[jc0], e = EE[jc0], f = FF[jc0];
        auto consts = [&](int bb, double &k0, double &k1, double &k2) {
            const double mu = MU[bb], r = RS[bb];
            const double a = gw * r, d = gb - a * mu;
            double pq = 0, qq = 0;
            if (p.has_bn) {
                pq = a * RB[jc0];
                qq = (d - MX[jc0]) * RB[jc0];
            }
            const double rgk = r * gw * k;
            k0 = rgk;
            k1 = -rgk * f * pq - r * r * M2s[bb];
            k2 = -rgk * e - rgk * f * qq - r * M1s[bb] + r * r * mu * M2s[bb];
        };
        double k0, k1, k2;
        consts(b, k0, k1, k2);
        sh_c[0] = (float)k0;
        sh_c[1] = (float)k1;
        sh_c[2] = (float)(k2 + (p.pre_bias ? k1 * p.pre_bias[c] : 0.0));   // in terms of the raw x: c1 (x + bias) + c2
        if (writer && b == 0) {   // this channel's parameter gradients
            double dw = 0, dbv = 0, dpb = 0;
            for (int bb = 0; bb < p.B; ++bb) {
                const double u1 = U1[bb * cpg + jc0], u2 = U2[bb * cpg + jc0];
                dw += RS[bb] * (u2 - MU[bb] * u1);
                dbv += u1;
                if (p.dpre_bias) {
                    double q0, q1, q2;
                    consts(bb, q0, q1, q2);
                    dpb += q0 * T1[bb * cpg + jc0] + q1 * S1[bb * cpg + jc0] + q2 * HW;   // sum_hw dx of row (bb, c)
                }
            }
            if (p.dgn_w) p.dgn_w[c] = (float)dw;
            if (p.dgn_b) p.dgn_b[c] = (float)dbv;
            if (p.dpre_bias) p.dpre_bias[c] = (float)dpb;
        }
    }
    __syncthreads();
    c0_out = sh_c[0];
    c1_out = sh_c[1];
    c2_out = sh_c[2];
}

// Forward apply with the constants computed in its prologue (nf_fwd_constants): grid (splits, B*C), block 256; a block
// walks its share of ONE (batch, channel) row.  HW % 4 == 0.
template <typename xin_t, typename act_t>
__global__ __launch_bounds__(256) void nf_apply_fwd_fused_kernel(FinArgs p, const xin_t *__restrict__ x,
                                                                 act_t *__restrict__ out, const act_t *__restrict__ res,
                                                                 int act) {
    const int bc = blockIdx.y;
    float Av, Dv;
    nf_fwd_constants(p, bc, blockIdx.x == 0, Av, Dv);
    const long base = (long)bc * p.HW;
    const int n4 = p.HW / 4;
    const int per = (n4 + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(lo + per, n4);
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        float xv[4], o[4], rv[4] = {0.f, 0.f, 0.f, 0.f};
        load_k<xin_t, 4, true>(x + base + 4 * i, 4, true, xv);
        if (res) load_k<act_t, 4, true>(res + base + 4 * i, 4, true, rv);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = act_fwd(fmaf(Av, xv[j], Dv) + rv[j], act);
        store_k<act_t, 4, true>(out + base + 4 * i, 4, true, o);
    }
}

// Backward apply with its constants from the prologue (nf_bwd_constants); same grid as the fused forward.
// out_cb_batch / res (= saved forward output: ReLU mask) / dres as in nf_apply_kernel.
template <typename xin_t, typename act_t>
__global__ __launch_bounds__(256) void nf_apply_bwd_fused_kernel(FinArgs p, const xin_t *__restrict__ x,
                                                                 const act_t *__restrict__ dout, xin_t *__restrict__ dx,
                                                                 int act, int out_cb_batch, const act_t *__restrict__ res,
                                                                 act_t *__restrict__ dres) {
    const int bc = blockIdx.y;
    float k0, k1, k2;
    nf_bwd_constants(p, bc, blockIdx.x == 0, k0, k1, k2);
    const float Av = p.A[bc], Dv = p.D[bc];
    const long base = (long)bc * p.HW;
    long obase = base;
    if (out_cb_batch > 0) {
        const int b = bc / p.C, c = bc - b * p.C;
        obase = ((long)c * out_cb_batch + b) * p.HW;
    }
    const int n4 = p.HW / 4;
    const int per = (n4 + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(lo + per, n4);
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        float xv[4], gv[4], rv[4] = {0.f, 0.f, 0.f, 0.f}, g2[4], o[4];
        load_k<xin_t, 4, true>(x + base + 4 * i, 4, true, xv);
        load_k<act_t, 4, true>(dout + base + 4 * i, 4, true, gv);
        if (res) load_k<act_t, 4, true>(res + base + 4 * i, 4, true, rv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g2[j] = gv[j] * (res ? (rv[j] > 0.f ? 1.f : 0.f) : act_grad(fmaf(Av, xv[j], Dv), act));
            o[j] = fmaf(k0, g2[j], fmaf(k1, xv[j], k2));
        }
        if (dres) store_k<act_t, 4, true>(dres + base + 4 * i, 4, true, g2);
        store_k<xin_t, 4, true>(dx + obase + 4 * i, 4, true, o);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small activations: ONE kernel each way.  On 16 x 16 / 32 x 32 maps (24 of MM-UNet's 47 MMConv blocks, two normalisations
// each) the moments pass and the apply pass are a few workgroups' work each and cost the ~4.6 us dependent-launch floor
// twice.  Here workgroup g owns GroupNorm group g for EVERY batch item -- all the rows that meet in the group statistics
// and in its channels' BatchNorm statistics -- so the whole normalisation is local to it: moments of its B * C/G rows
// (a wave per row), the per-row constants (the algebra of nf_fwd_constants / nf_bwd_constants for all its rows at once,
// same order of operations and the same float roundings of the saved statistics), then the apply pass over the same rows,
// which are still in L2.  Taken when B * C/G <= 64 rows of together <= 65,536 elements (nf_one_ok).
constexpr int ONE_FWD_THREADS = 512, ONE_FWD_WAVES = ONE_FWD_THREADS / 64;
constexpr int ONE_ITEMS = 32, ONE_BWD_ITEMS = 16;   // float4 items a lane holds: x forward; x and g2 backward
// (16 waves x 16 items spilled 32 VGPRs under the 128-register cap of a 1,024-thread workgroup)

// A wave's rows are w, w + 8, ...; a row is ipr = ceil(HW / 256) float4 items per lane; nf_one_ok guarantees
// rows_per_wave * ipr <= ONE_ITEMS, so the forward keeps its x in registers between the moments and the apply phase.
// Every parameter the algebra needs is fetched at the top (the loads fly together with those of x): after the first
// barrier nothing waits for global memory again.
template <typename xin_t, typename act_t, int IPR>
__global__ __launch_bounds__(ONE_FWD_THREADS) void nf_one_fwd_kernel(FinArgs p, const xin_t *__restrict__ x,
                                                                    act_t *__restrict__ out, const act_t *__restrict__ res,
                                                                    int act) {
    // IPR: float4 items per lane and row (a power of two >= ceil(HW / 256)); RPW rows per wave at most
    constexpr int RPW = ONE_ITEMS / IPR < 8 ? ONE_ITEMS / IPR : 8;   // (64 rows at most: 8 per wave)
    __shared__ double S1[64], S2[64], sh_mu[64], sh_rs[64];
    __shared__ float sh_m[64], sh_rb[64], shA[64], shD[64];
    const int cpg = p.C / p.G, g = blockIdx.x, rows = p.B * cpg;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n4 = p.HW / 4;
    const double n = (double)cpg * p.HW, N = (double)p.B * p.HW;
    // ---- parameters of this thread's row (tid < rows) / channel (tid < cpg)
    float gwf = 1.f, gbf = 0.f, bwf = 1.f, bbf = 0.f, pbf = 0.f, rmf = 0.f, rvf = 1.f;
    if (tid < rows) {
        const int c = g * cpg + tid % cpg;
        if (p.gn_w) gwf = p.gn_w[c];
        if (p.gn_b) gbf = p.gn_b[c];
        if (p.has_bn && p.bn_w) bwf = p.bn_w[c];
        if (p.has_bn && p.bn_b) bbf = p.bn_b[c];
        if (p.pre_bias) pbf = p.pre_bias[c];
        if (tid < cpg && p.has_bn && p.run_mean) {
            rmf = p.run_mean[c];
            rvf = p.run_var[c];
        }
    }
    // ---- x of this wave's rows -> registers (clamped addresses, no branches: every load is in flight before the first
    //      use); raw moments per row
    float xr[RPW][IPR][4];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = w + ONE_FWD_WAVES * q;
        const int rr = r < rows ? r : 0;
        const xin_t *xp = x + ((long)(rr / cpg) * p.C + g * cpg + rr % cpg) * p.HW;
#pragma unroll
        for (int k = 0; k < IPR; ++k) {
            const int i = lane + 64 * k;
            load_k<xin_t, 4, true>(xp + 4 * (i < n4 ? i : 0), 4, true, xr[q][k]);
            const bool ok = r < rows && i < n4;
#pragma unroll
            for (int j = 0; j < 4; ++j) xr[q][k][j] = ok ? xr[q][k][j] : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = w + ONE_FWD_WAVES * q;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < IPR; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s1 += xr[q][k][j];
                s2 = fmaf(xr[q][k][j], xr[q][k][j], s2);
            }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0 && r < rows) {
            S1[r] = s1;
            S2[r] = s2;
        }
    }
    __syncthreads();
    if (tid < rows) {   // moments of x + pre_bias (rounded to float as the two-pass kernels store them), saved
        float s1 = (float)S1[tid], s2 = (float)S2[tid];
        if (p.pre_bias) {
            const double bv = pbf, d1 = s1, d2 = s2;
            s1 = (float)(d1 + bv * p.HW);
            s2 = (float)(d2 + 2 * bv * d1 + bv * bv * p.HW);
            S1[tid] = s1;
            S2[tid] = s2;
        }
        const int bc = (tid / cpg) * p.C + g * cpg + tid % cpg;
        p.s1[bc] = s1;
        p.s2[bc] = s2;
    }
    __syncthreads();
    // ---- constants of every row of the group
    if (tid < p.B) {
        double mu = 0, rs = 1;
        if (p.has_gn) {
            double a1 = 0, a2 = 0;
            for (int j = 0; j < cpg; ++j) {
                a1 += S1[tid * cpg + j];
                a2 += S2[tid * cpg + j];
            }
            mu = a1 / n;
            double var = a2 / n - mu * mu;
            var = var < 0 ? 0 : var;
            rs = 1.0 / sqrt(var + (double)p.eps_g);
        }
        const float muf = (float)mu, rsf = (float)rs;   // the float values the backward reads back
        sh_mu[tid] = muf;
        sh_rs[tid] = rsf;
        p.mu[tid * p.G + g] = muf;
        p.rstd[tid * p.G + g] = rsf;
    }
    __syncthreads();
    if (tid < cpg && p.has_bn) {
        const int c = g * cpg + tid;
        const double gw = gwf, gb = gbf;
        double m, rb;
        if (p.training) {
            double sm = 0, sq = 0;
            for (int bb = 0; bb < p.B; ++bb) {
                const double ab = gw * sh_rs[bb], db = gb - ab * sh_mu[bb];
                const double s1 = S1[bb * cpg + tid], s2 = S2[bb * cpg + tid];
                sm += ab * s1 + db * p.HW;
                sq += ab * ab * s2 + 2 * ab * db * s1 + db * db * p.HW;
            }
            m = sm / N;
            double v = sq / N - m * m;
            v = v < 0 ? 0 : v;
            rb = 1.0 / sqrt(v + (double)p.eps_b);
            if (p.run_mean) {
                p.run_mean[c] = (float)((1.0 - p.momentum) * rmf + p.momentum * m);
                p.run_var[c] = (float)((1.0 - p.momentum) * rvf + p.momentum * v * (N > 1 ? N / (N - 1) : 1.0));
            }
        } else {
            m = rmf;
            rb = 1.0 / sqrt((double)rvf + (double)p.eps_b);
        }
        const float mf = (float)m, rf = (float)rb;
        p.bmean[c] = mf;
        p.brstd[c] = rf;
        sh_m[tid] = mf;
        sh_rb[tid] = rf;
    }
    __syncthreads();
    if (tid < rows) {
        const int bb = tid / cpg, j = tid % cpg, c = g * cpg + j;
        const double gw = gwf, gb = gbf;
        double a = gw * sh_rs[bb], d = gb - a * sh_mu[bb];
        if (p.has_bn) {
            const double k = (double)bwf * sh_rb[j];
            d = (d - sh_m[j]) * k + (double)bbf;
            a = a * k;
        }
        if (p.pre_bias) d += a * (double)pbf;
        p.A[bb * p.C + c] = (float)a;
        p.D[bb * p.C + c] = (float)d;
        shA[tid] = (float)a;
        shD[tid] = (float)d;
    }
    __syncthreads();
    // ---- apply, from the registers
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = w + ONE_FWD_WAVES * q;
        if (r < rows) {
            const long rbase = ((long)(r / cpg) * p.C + g * cpg + r % cpg) * p.HW;
            const float Av = shA[r], Dv = shD[r];
#pragma unroll
            for (int k = 0; k < IPR; ++k) {
                const int i = lane + 64 * k;
                if (i < n4) {
                    float o[4], rv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (res) load_k<act_t, 4, true>(res + rbase + 4 * i, 4, true, rv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = act_fwd(fmaf(Av, xr[q][k][j], Dv) + rv[j], act);
                    store_k<act_t, 4, true>(out + rbase + 4 * i, 4, true, o);
                }
            }
        }
    }
}

template <typename xin_t, typename act_t, int IPR>
__global__ __launch_bounds__(ONE_FWD_THREADS) void nf_one_bwd_kernel(FinArgs p, const xin_t *__restrict__ x,
                                                                    const act_t *__restrict__ dout, xin_t *__restrict__ dx,
                                                                    int act, int out_cb_batch, const act_t *__restrict__ res,
                                                                    act_t *__restrict__ dres) {
    // x and g2 = dout * act' of this wave's rows stay in registers between the two phases (ONE_BWD_ITEMS float4 each)
    constexpr int RPW = ONE_BWD_ITEMS / IPR < 8 ? ONE_BWD_ITEMS / IPR : 8;
    __shared__ double T1[64], T2[64], S1[64], S2[64], U1[64], U2[64];   // [b][channel of the group]
    __shared__ double MU[64], RS[64], M1s[64], M2s[64];                  // [b]
    __shared__ double KK[64], EE[64], FF[64], MX[64], GW[64], GB[64], RB[64], BM[64], BW[64], PB[64];   // [channel]
    __shared__ float shc[3][64];
    const int cpg = p.C / p.G, g = blockIdx.x, rows = p.B * cpg;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n4 = p.HW / 4;
    const double n = (double)cpg * p.HW, N = (double)p.B * p.HW, HW = p.HW;
    // ---- everything the algebra reads from global memory, fetched now (in flight together with the rows)
    if (tid < rows) {
        const int bc = (tid / cpg) * p.C + g * cpg + tid % cpg;
        S1[tid] = p.s1[bc];
        S2[tid] = p.s2[bc];
    }
    if (tid >= 64 && tid < 64 + p.B) {
        MU[tid - 64] = p.mu[(tid - 64) * p.G + g];
        RS[tid - 64] = p.rstd[(tid - 64) * p.G + g];
    }
    if (tid >= 128 && tid < 128 + cpg) {
        const int cc = g * cpg + tid - 128;
        GW[tid - 128] = p.gn_w ? p.gn_w[cc] : 1.0;
        GB[tid - 128] = p.gn_b ? p.gn_b[cc] : 0.0;
        RB[tid - 128] = p.has_bn ? p.brstd[cc] : 1.0;
        BM[tid - 128] = p.has_bn ? p.bmean[cc] : 0.0;
        BW[tid - 128] = (p.has_bn && p.bn_w) ? p.bn_w[cc] : 1.0;
        PB[tid - 128] = p.pre_bias ? p.pre_bias[cc] : 0.0;
    }
    // ---- rows -> registers (clamped addresses, no branches); t1 = sum g2, t2 = sum g2 x per row
    float xr[RPW][IPR][4], gr[RPW][IPR][4];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = w + ONE_FWD_WAVES * q;
        const int rr = r < rows ? r : 0;
        const int bc = (rr / cpg) * p.C + g * cpg + rr % cpg;
        const long base = (long)bc * p.HW;
        const float Av = p.A[bc], Dv = p.D[bc];
#pragma unroll
        for (int k = 0; k < IPR; ++k) {
            const int i = lane + 64 * k;
            const long off = base + 4 * (i < n4 ? i : 0);
            float gv[4], ov[4] = {0.f, 0.f, 0.f, 0.f};
            load_k<xin_t, 4, true>(x + off, 4, true, xr[q][k]);
            load_k<act_t, 4, true>(dout + off, 4, true, gv);
            if (res) load_k<act_t, 4, true>(res + off, 4, true, ov);
            const bool ok = r < rows && i < n4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g2 = gv[j] * (res ? (ov[j] > 0.f ? 1.f : 0.f) : act_grad(fmaf(Av, xr[q][k][j], Dv), act));
                gr[q][k][j] = ok ? g2 : 0.f;
                xr[q][k][j] = ok ? xr[q][k][j] : 0.f;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = w + ONE_FWD_WAVES * q;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < IPR; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s1 += gr[q][k][j];
                s2 = fmaf(gr[q][k][j], xr[q][k][j], s2);
            }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0 && r < rows) {
            T1[r] = s1;
            T2[r] = s2;   // (+ pre_bias * t1 below, once the prefetched bias is visible)
        }
    }
    __syncthreads();
    if (tid < rows && p.pre_bias) T2[tid] += PB[tid % cpg] * T1[tid];   // sum_hw g2 (x + bias)
    __syncthreads();
    if (tid < cpg) {   // per channel: exact batch mean, d(bn weight / bias), k, e, f
        const int cc = g * cpg + tid;
        double k = 1, e = 0, f = 0, m = 0;
        if (p.has_bn) {
            const double gw = GW[tid], gb = GB[tid], rb = RB[tid];
            m = BM[tid];
            if (p.training) {
                double sm = 0;
                for (int bb = 0; bb < p.B; ++bb) {
                    const double a = gw * RS[bb], d = gb - a * MU[bb];
                    sm += a * S1[bb * cpg + tid] + d * p.HW;
                }
                m = sm / N;
            }
            double db = 0, dg = 0;
            for (int bb = 0; bb < p.B; ++bb) {
                const double a = gw * RS[bb], d = gb - a * MU[bb];
                const double pq = a * rb, qq = (d - m) * rb;
                db += T1[bb * cpg + tid];
                dg += pq * T2[bb * cpg + tid] + qq * T1[bb * cpg + tid];
            }
            if (p.dbn_w) p.dbn_w[cc] = (float)dg;
            if (p.dbn_b) p.dbn_b[cc] = (float)db;
            k = BW[tid] * rb;
            if (p.training) {
                e = db / N;
                f = dg / N;
            }
        }
        KK[tid] = k; EE[tid] = e; FF[tid] = f; MX[tid] = m;
    }
    __syncthreads();
    if (tid < rows) {   // u1 = sum_hw g1, u2 = sum_hw g1 x of every row
        const int bb = tid / cpg, j = tid % cpg;
        const double a = GW[j] * RS[bb], d = GB[j] - a * MU[bb];
        double pq = 0, qq = 0;
        if (p.has_bn) {
            pq = a * RB[j];
            qq = (d - MX[j]) * RB[j];
        }
        const double k = KK[j], e = EE[j], f = FF[j];
        U1[tid] = k * (T1[tid] - e * HW - f * (pq * S1[tid] + qq * HW));
        U2[tid] = k * (T2[tid] - e * S1[tid] - f * (pq * S2[tid] + qq * S1[tid]));
    }
    __syncthreads();
    if (tid < p.B) {   // group means of gamma g1 and gamma g1 xhat, per batch item
        double a1 = 0, a2 = 0;
        for (int j = 0; j < cpg; ++j) {
            a1 += GW[j] * U1[tid * cpg + j];
            a2 += GW[j] * RS[tid] * (U2[tid * cpg + j] - MU[tid] * U1[tid * cpg + j]);
        }
        M1s[tid] = p.has_gn ? a1 / n : 0.0;
        M2s[tid] = p.has_gn ? a2 / n : 0.0;
    }
    __syncthreads();
    auto consts = [&](int bb, int j, double &k0, double &k1, double &k2) {
        const double gw = GW[j], gb = GB[j], k = KK[j], e = EE[j], f = FF[j];
        const double mu = MU[bb], r = RS[bb];
        const double a = gw * r, d = gb - a * mu;
        double pq = 0, qq = 0;
        if (p.has_bn) {
            pq = a * RB[j];
            qq = (d - MX[j]) * RB[j];
        }
        const double rgk = r * gw * k;
        k0 = rgk;
        k1 = -rgk * f * pq - r * r * M2s[bb];
        k2 = -rgk * e - rgk * f * qq - r * M1s[bb] + r * r * mu * M2s[bb];
    };
    if (tid < rows) {
        const int bb = tid / cpg, j = tid % cpg;
        double k0, k1, k2;
        consts(bb, j, k0, k1, k2);
        shc[0][tid] = (float)k0;
        shc[1][tid] = (float)k1;
        shc[2][tid] = (float)(k2 + (p.pre_bias ? k1 * PB[j] : 0.0));   // in terms of the raw x
    }
    if (tid >= 64 && tid < 64 + cpg) {   // this channel's parameter gradients
        const int j = tid - 64, c = g * cpg + j;
        double dw = 0, dbv = 0, dpb = 0;
        for (int bb = 0; bb < p.B; ++bb) {
            const double u1 = U1[bb * cpg + j], u2 = U2[bb * cpg + j];
            dw += RS[bb] * (u2 - MU[bb] * u1);
            dbv += u1;
            if (p.dpre_bias) {
                double q0, q1, q2;
                consts(bb, j, q0, q1, q2);
                dpb += q0 * T1[bb * cpg + j] + q1 * S1[bb * cpg + j] + q2 * HW;   // sum_hw dx of row (bb, c)
            }
        }
        if (p.dgn_w) p.dgn_w[c] = (float)dw;
        if (p.dgn_b) p.dgn_b[c] = (float)dbv;
        if (p.dpre_bias) p.dpre_bias[c] = (float)dpb;
    }
    __syncthreads();
    // ---- dx = c0 g2 + c1 x + c2, from the registers
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = w + ONE_FWD_WAVES * q;
        if (r < rows) {
            const int bb = r / cpg, c = g * cpg + r % cpg;
            const long base = ((long)bb * p.C + c) * p.HW;
            const long obase = out_cb_batch > 0 ? ((long)c * out_cb_batch + bb) * p.HW : base;
            const float k0 = shc[0][r], k1 = shc[1][r], k2 = shc[2][r];
#pragma unroll
            for (int k = 0; k < IPR; ++k) {
                const int i = lane + 64 * k;
                if (i < n4) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(k0, gr[q][k][j], fmaf(k1, xr[q][k][j], k2));
                    if (dres) store_k<act_t, 4, true>(dres + base + 4 * i, 4, true, gr[q][k]);
                    store_k<xin_t, 4, true>(dx + obase + 4 * i, 4, true, o);
                }
            }
        }
    }
}

// runtime dtype codes -> the two element types
#define NF_TYPES(xd, ad, ...)                                   \
    do {                                                        \
        if ((xd) == MMU_DTYPE_F32 && (ad) == MMU_DTYPE_F32) {   \
            using xin_t = float; using act_t = float;           \
            __VA_ARGS__                                         \
        } else if ((xd) == MMU_DTYPE_F32) {                     \
            using xin_t = float; using act_t = bf16_t;          \
            __VA_ARGS__                                         \
        } else if ((ad) == MMU_DTYPE_F32) {                     \
            using xin_t = bf16_t; using act_t = float;          \
            __VA_ARGS__                                         \
        } else {                                                \
            using xin_t = bf16_t; using act_t = bf16_t;         \
            __VA_ARGS__                                         \
        }                                                       \
    } while (0)

int fill(const mmu_norm_params *p, FinArgs &a, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->batch > 0 && p->channels > 0 && p->hw > 0 && p->groups > 0 && p->channels % p->groups == 0,
              "%s: need batch, channels, hw > 0 and channels divisible by groups", name);
    MMU_CHECK(p->act >= 0 && p->act <= 2, "%s: unknown activation %d", name, p->act);
    MMU_CHECK((p->x_dtype == MMU_DTYPE_F32 || p->x_dtype == MMU_DTYPE_BF16) &&
                  (p->act_dtype == MMU_DTYPE_F32 || p->act_dtype == MMU_DTYPE_BF16),
              "%s: unsupported element type (%d, %d)", name, p->x_dtype, p->act_dtype);
    MMU_CHECK((long)p->batch * p->channels < 65536, "%s: batch * channels must be < 65536", name);
    a = FinArgs{};
    a.B = p->batch; a.C = p->channels; a.G = p->groups; a.HW = p->hw; a.has_bn = p->has_bn; a.training = p->training;
    a.has_gn = p->has_gn;
    if (!p->has_gn && (!p->has_bn || p->groups != p->channels))
        return mmu_fail("%s: has_gn = 0 needs has_bn = 1 and groups == channels", name);
    a.eps_g = p->gn_eps; a.eps_b = p->bn_eps; a.momentum = p->momentum;
    a.pre_bias = p->pre_bias;
    a.s1 = p->s1; a.s2 = p->s2; a.gn_w = p->gn_weight; a.gn_b = p->gn_bias; a.bn_w = p->bn_weight; a.bn_b = p->bn_bias;
    a.run_mean = p->running_mean; a.run_var = p->running_var; a.mu = p->mu; a.rstd = p->rstd;
    a.bmean = p->bn_mean; a.brstd = p->bn_rstd; a.A = p->scale; a.D = p->shift;
    return 0;
}

}  // namespace

// launch shapes of the streaming kernels: enough bytes in flight when the rows are few
static inline int nf_moment_threads(int BC, int HW) { return (HW / 4 >= 2048 && BC <= 2048) ? 1024 : 256; }
// apply kernels: a few thousand workgroups at most, each with >= 16 KB of its row (a block's prologue computes the row's
// constants: too small a share and the prologue dominates)
static inline int nf_one_ipr(int HW) {   // float4 items per lane and row, rounded up to a power of two
    int ipr = 1;
    while (ipr * 256 < HW) ipr *= 2;
    return ipr;
}
// one kernel each way (nf_one_*): every row of a group in one workgroup, small enough to be re-read from L2
// Measured (profiles/r03_norm_one_kernel.txt): a workgroup streams only ~16 GB/s (the bytes in flight per CU over the
// memory latency), so a group's rows must be small -- at 128 KB per workgroup the single kernel took 15 us against 13 us
// for the two passes on 2,048 workgroups each; at <= 64 KB (forward) / <= 32 KB (backward: x and dout) it wins.
static inline bool nf_one_ok(const FinArgs &a, int items = ONE_ITEMS) {
    static const bool off = []() { const char *e = getenv("MMU_NF_ONE"); return e && e[0] == '0'; }();
    const int cpg = a.C / a.G;
    const int rows = a.B * cpg, rpw = (rows + ONE_FWD_WAVES - 1) / ONE_FWD_WAVES;
    return !off && (a.HW & 3) == 0 && rows <= 64 && a.B <= 64 && a.HW <= 4 * 64 * items && rpw * nf_one_ipr(a.HW) <= items &&
           (long)rows * a.HW <= (items == ONE_ITEMS ? 16384 : 8192);
}
static inline int nf_apply_splits(int BC, int HW) {
    int splits = (HW / 4 + 1023) / 1024;
    while (splits > 1 && (long)splits * BC > 8192) --splits;
    return splits;
}

extern "C" int mmu_norm_fused_fwd(const mmu_norm_params *p, void *stream) {
    FinArgs a;
    if (int r = fill(p, a, "norm_fused_fwd")) return r;
    MMU_CHECK(p->input && p->out && p->s1 && p->s2 && p->mu && p->rstd && p->scale && p->shift,
              "norm_fused_fwd: input, out, s1, s2, mu, rstd, scale, shift are required");
    MMU_CHECK(!p->has_bn || (p->bn_mean && p->bn_rstd), "norm_fused_fwd: bn_mean / bn_rstd buffers are required");
    MMU_CHECK(!p->has_bn || p->training || (p->running_mean && p->running_var),
              "norm_fused_fwd: eval-mode BatchNorm needs running statistics");
    hipStream_t st = (hipStream_t)stream;
    const int BC = a.B * a.C;
    MMU_CHECK(!p->residual || p->act == ACT_RELU, "norm_fused_fwd: a residual input needs act = ReLU");
    dim3 grid((a.HW + 1023) / 1024, BC);
    // the per-(b, c) algebra runs in the apply kernel's prologue when its LDS tables hold the problem (always in
    // MM-UNet); otherwise in the single-workgroup kernel between the passes
    const int cpg = a.C / a.G;
    const bool fused = (a.HW & 3) == 0 && a.B * cpg <= 64 && a.B <= 64 && getenv("MMU_NF_FINALIZE_KERNEL") == nullptr;
    if (nf_one_ok(a) && !(p->has_bn && !p->training && !(p->running_mean && p->running_var))) {
#define NF_ONE_FWD(I)                                                                                                  \
    nf_one_fwd_kernel<xin_t, act_t, I><<<a.G, ONE_FWD_THREADS, 0, st>>>(a, (const xin_t *)p->input, (act_t *)p->out, \
                                                                        (const act_t *)p->residual, p->act)
        NF_TYPES(p->x_dtype, p->act_dtype, {
            switch (nf_one_ipr(a.HW)) {
                case 1: NF_ONE_FWD(1); break;
                case 2: NF_ONE_FWD(2); break;
                case 4: NF_ONE_FWD(4); break;
                case 8: NF_ONE_FWD(8); break;
                case 16: NF_ONE_FWD(16); break;
                default: NF_ONE_FWD(32); break;
            }
        });
#undef NF_ONE_FWD
        MMU_HIP_LAUNCH_CHECK("norm_fused_fwd(one)");
        return 0;
    }
    NF_TYPES(p->x_dtype, p->act_dtype, {
        nf_moments_kernel<false, xin_t, act_t><<<BC, nf_moment_threads(BC, a.HW), 0, st>>>((const xin_t *)p->input, nullptr, nullptr, nullptr, p->s1,
                                                                  p->s2, a.HW, 0, nullptr, fused ? p->pre_bias : nullptr,
                                                                  a.C);
        if (fused) {
            const int splits = nf_apply_splits(BC, a.HW);
            nf_apply_fwd_fused_kernel<xin_t, act_t><<<dim3(splits, BC), 256, 0, st>>>(
                a, (const xin_t *)p->input, (act_t *)p->out, (const act_t *)p->residual, p->act);
        } else {
            nf_finalize_fwd_kernel<<<1, BC >= 1024 ? 1024 : 256, 0, st>>>(a);
            nf_apply_kernel<false, xin_t, act_t><<<grid, 256, 0, st>>>((const xin_t *)p->input, nullptr, a.A, a.D, nullptr,
                                                                      nullptr, nullptr, p->out, a.HW, p->act, 0,
                                                                      (const act_t *)p->residual, nullptr);
        }
    });
    MMU_HIP_LAUNCH_CHECK("norm_fused_fwd");
    return 0;
}

extern "C" int mmu_norm_fused_bwd(const mmu_norm_params *p, void *stream) {
    FinArgs a;
    if (int r = fill(p, a, "norm_fused_bwd")) return r;
    MMU_CHECK(p->input && p->dout && p->dinput && p->s1 && p->s2 && p->mu && p->rstd && p->scale && p->shift &&
                  p->workspace,
              "norm_fused_bwd: input, dout, dinput, saved statistics and workspace are required");
    hipStream_t st = (hipStream_t)stream;
    const int BC = a.B * a.C;
    // workspace: t1[BC] t2[BC] c0[BC] c1[BC] c2[BC] (+1 float if needed for 8-byte alignment) | scratch
    float *ws = p->workspace;
    float *t1 = ws, *t2 = ws + BC;
    a.t1 = t1; a.t2 = t2; a.c0 = ws + 2 * BC; a.c1 = ws + 3 * BC; a.c2 = ws + 4 * BC;
    a.scratch = ws + 5 * BC;
    if ((uintptr_t)a.scratch & 7) a.scratch += 1;  // the first entry of the scratch area is a double array
    a.dgn_w = p->dgn_weight; a.dgn_b = p->dgn_bias; a.dbn_w = p->dbn_weight; a.dbn_b = p->dbn_bias;
    a.dpre_bias = p->pre_bias ? p->dpre_bias : nullptr;
    MMU_CHECK((p->act_out == nullptr) == (p->dresidual == nullptr) && (!p->act_out || p->act == ACT_RELU),
              "norm_fused_bwd: act_out and dresidual go together (residual mode, ReLU only)");
    dim3 grid((a.HW + 1023) / 1024, BC);
    const int cpg = a.C / a.G;
    const bool fused = (a.HW & 3) == 0 && a.B * cpg <= 64 && a.B <= 64 && getenv("MMU_NF_FINALIZE_KERNEL") == nullptr;
    if (nf_one_ok(a, ONE_BWD_ITEMS)) {
#define NF_ONE_BWD(I)                                                                                      \
    nf_one_bwd_kernel<xin_t, act_t, I><<<a.G, ONE_FWD_THREADS, 0, st>>>(                                   \
        a, (const xin_t *)p->input, (const act_t *)p->dout, (xin_t *)p->dinput, p->act,                    \
        p->dinput_channel_major ? a.B : 0, (const act_t *)p->act_out, (act_t *)p->dresidual)
        NF_TYPES(p->x_dtype, p->act_dtype, {
            switch (nf_one_ipr(a.HW)) {
                case 1: NF_ONE_BWD(1); break;
                case 2: NF_ONE_BWD(2); break;
                case 4: NF_ONE_BWD(4); break;
                case 8: NF_ONE_BWD(8); break;
                default: NF_ONE_BWD(16); break;
            }
        });
#undef NF_ONE_BWD
        MMU_HIP_LAUNCH_CHECK("norm_fused_bwd(one)");
        return 0;
    }
    NF_TYPES(p->x_dtype, p->act_dtype, {
        nf_moments_kernel<true, xin_t, act_t><<<BC, nf_moment_threads(BC, a.HW), 0, st>>>((const xin_t *)p->input, (const act_t *)p->dout, a.A, a.D,
                                                                 t1, t2, a.HW, p->act, (const act_t *)p->act_out, nullptr, a.C);
        if (fused) {
            const int splits = nf_apply_splits(BC, a.HW);
            nf_apply_bwd_fused_kernel<xin_t, act_t><<<dim3(splits, BC), 256, 0, st>>>(
                a, (const xin_t *)p->input, (const act_t *)p->dout, (xin_t *)p->dinput, p->act,
                p->dinput_channel_major ? a.B : 0, (const act_t *)p->act_out, (act_t *)p->dresidual);
        } else {
            nf_finalize_bwd_kernel<<<1, BC >= 1024 ? 1024 : 256, 0, st>>>(a);
            nf_apply_kernel<true, xin_t, act_t><<<grid, 256, 0, st>>>((const xin_t *)p->input, (const act_t *)p->dout, a.A, a.D,
                                                                     a.c0, a.c1, a.c2, p->dinput, a.HW, p->act,
                                                                     p->dinput_channel_major ? a.B : 0,
                                                                     (const act_t *)p->act_out, (act_t *)p->dresidual);
        }
    });
    MMU_HIP_LAUNCH_CHECK("norm_fused_bwd");
    return 0;
}

extern "C" size_t mmu_norm_fused_workspace_floats(int batch, int channels, int groups) {
    return (size_t)7 * batch * channels + (size_t)2 * batch * groups + (size_t)5 * channels + 2;
}
