/*
 * mmu_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the arithmetic of the two CUDA extensions the
 * reference MM-UNet depends on:
 *   selective scan fwd : requirements/Mamba/mamba/csrc/selective_scan/selective_scan_fwd_kernel.cuh:147-298
 *   selective scan bwd : requirements/Mamba/mamba/csrc/selective_scan/selective_scan_bwd_kernel.cuh:161-487
 *   causal conv1d fwd  : requirements/Mamba/causal-conv1d/csrc/causal_conv1d_fwd.cu:99-118
 *   causal conv1d bwd  : requirements/Mamba/causal-conv1d/csrc/causal_conv1d_bwd.cu:153-222
 * and pinned against the reference's own pure-torch oracles
 *   selective_scan_ref (mamba_ssm/ops/selective_scan_interface.py:86-152)
 *   causal_conv1d_ref  (causal_conv1d/causal_conv1d_interface.py:49-65)
 * through the golden vectors in tests/golden/ (tools/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (mm-unet_amd/) never does.
 *
 * Everything is a sequential walk over time in fp32, one (batch, channel)
 * at a time -- deliberately the most literal form of the recurrence.
 * OpenMP (if compiled with -fopenmp) only spreads (batch, channel) pairs
 * over cores for the cpu_baseline timing; results do not depend on it
 * except for the fp32 summation order of dA/dD/dbias/dB/dC across threads,
 * which is made deterministic by per-(b,d) partials reduced in fixed order.
 *
 * All tensors are dense row-major:
 *   u, delta, z, out, out_z, dout, du, ddelta, dz : [batch][dim][seqlen]
 *   A, dA                                         : [dim][dstate]
 *   B, C, dB, dC                                  : [batch][ngroups][dstate][seqlen]
 *   D, delta_bias, dD, ddelta_bias                : [dim]
 *   last_state                                    : [batch][dim][dstate]
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* One source, two builds: float (the arithmetic of the kernels: libmmu_oracle.so) and -DMMU_ORACLE_DOUBLE (every value and
 * every function in double: libmmu_oracle64.so, entry points mmu_oracle64_*) -- the float64 "truth" that the gradient
 * tests measure both the build and the float32 reference against (tests/golden/mmnet_128_train_fp64.npz). */
#ifdef MMU_ORACLE_DOUBLE
typedef double real;
#define R_EXP exp
#define R_EXP2 exp2
#define R_LOG1P log1p
#define mmu_oracle_num_threads mmu_oracle64_num_threads
#define mmu_oracle_selective_scan_fwd mmu_oracle64_selective_scan_fwd
#define mmu_oracle_selective_scan_bwd mmu_oracle64_selective_scan_bwd
#define mmu_oracle_causal_conv1d_fwd mmu_oracle64_causal_conv1d_fwd
#define mmu_oracle_causal_conv1d_bwd mmu_oracle64_causal_conv1d_bwd
#define MMU_LOG2E 1.4426950408889634
#else
typedef float real;
#define R_EXP expf
#define R_EXP2 exp2f
#define R_LOG1P log1pf
#define MMU_LOG2E 1.4426950408889634f
#endif

int mmu_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* softplus with the kernel's threshold: selective_scan_fwd_kernel.cuh:153-156 */
static inline real softplus_thr(real x) { return x <= 20.f ? R_LOG1P(R_EXP(x)) : x; }

/* ------------------------------------------------------------------ */
/* selective scan forward                                              */
/* ------------------------------------------------------------------ */
void mmu_oracle_selective_scan_fwd(
    const real *u, const real *delta, const real *A, const real *B, const real *C,
    const real *D /*nullable*/, const real *z /*nullable*/, const real *delta_bias /*nullable*/,
    int delta_softplus,
    real *out /*nullable*/, real *out_z /*nullable, needs z*/, real *last_state /*nullable*/,
    int batch, int dim, int seqlen, int dstate, int ngroups)
{
    const int dpg = dim / ngroups;
    #pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
        for (int d = 0; d < dim; ++d) {
            const int g = d / dpg;
            const real *ub = u + ((size_t)b * dim + d) * seqlen;
            const real *db = delta + ((size_t)b * dim + d) * seqlen;
            const real *zb = z ? z + ((size_t)b * dim + d) * seqlen : NULL;
            const real *Bb = B + ((size_t)b * ngroups + g) * dstate * seqlen;
            const real *Cb = C + ((size_t)b * ngroups + g) * dstate * seqlen;
            real *ob = out ? out + ((size_t)b * dim + d) * seqlen : NULL;
            real *ozb = out_z ? out_z + ((size_t)b * dim + d) * seqlen : NULL;
            const real Dv = D ? D[d] : 0.f;
            const real bias = delta_bias ? delta_bias[d] : 0.f;
            real h[256];
            real a2[256];
            for (int n = 0; n < dstate; ++n) { h[n] = 0.f; a2[n] = A[(size_t)d * dstate + n] * MMU_LOG2E; }
            for (int t = 0; t < seqlen; ++t) {
                real dl = db[t] + bias;
                if (delta_softplus) dl = softplus_thr(dl);
                const real uv = ub[t];
                const real du_ = dl * uv;
                real y = Dv * uv;
                for (int n = 0; n < dstate; ++n) {
                    /* R_EXP2(delta * A * log2e): selective_scan_fwd_kernel.cuh:168-171,216 */
                    const real a = R_EXP2(dl * a2[n]);
                    h[n] = a * h[n] + du_ * Bb[(size_t)n * seqlen + t];
                    y += h[n] * Cb[(size_t)n * seqlen + t];
                }
                if (ob) ob[t] = y;
                if (ozb) { const real zv = zb[t]; ozb[t] = y * (zv / (1.f + R_EXP(-zv))); }
            }
            if (last_state)
                for (int n = 0; n < dstate; ++n) last_state[((size_t)b * dim + d) * dstate + n] = h[n];
        }
    }
}

/* ------------------------------------------------------------------ */
/* selective scan backward                                             */
/* math: SURVEY.md section 8a (restated from selective_scan_bwd_kernel.cuh) */
/* ------------------------------------------------------------------ */
void mmu_oracle_selective_scan_bwd(
    const real *u, const real *delta, const real *A, const real *B, const real *C,
    const real *D, const real *z, const real *delta_bias, const real *dout,
    int delta_softplus,
    real *du, real *ddelta, real *dA, real *dB, real *dC,
    real *dD /*nullable*/, real *dz /*nullable*/, real *ddelta_bias /*nullable*/,
    int batch, int dim, int seqlen, int dstate, int ngroups)
{
    const int dpg = dim / ngroups;
    const size_t nbd = (size_t)batch * dim;
    /* per-(b,d) partials so that the cross-thread reduction order is fixed */
    real *pdA = (real *)calloc(nbd * dstate, sizeof(real));
    real *pdD = (real *)calloc(nbd, sizeof(real));
    real *pdb = (real *)calloc(nbd, sizeof(real));
    /* dB/dC contributions of one channel: [b][d][n][t] would be huge; instead
       parallelise over (b, g) and walk the channels of the group serially. */
    memset(dB, 0, sizeof(real) * (size_t)batch * ngroups * dstate * seqlen);
    memset(dC, 0, sizeof(real) * (size_t)batch * ngroups * dstate * seqlen);

    #pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
        for (int g = 0; g < ngroups; ++g) {
            real *hs = (real *)malloc(sizeof(real) * (size_t)seqlen * dstate); /* h_t per n */
            real *dls = (real *)malloc(sizeof(real) * (size_t)seqlen);          /* softplus'd delta */
            const real *Bb = B + ((size_t)b * ngroups + g) * dstate * seqlen;
            const real *Cb = C + ((size_t)b * ngroups + g) * dstate * seqlen;
            real *dBb = dB + ((size_t)b * ngroups + g) * dstate * seqlen;
            real *dCb = dC + ((size_t)b * ngroups + g) * dstate * seqlen;
            for (int d = g * dpg; d < (g + 1) * dpg; ++d) {
                const size_t bd = (size_t)b * dim + d;
                const real *ub = u + bd * seqlen, *db = delta + bd * seqlen, *gb = dout + bd * seqlen;
                const real *zb = z ? z + bd * seqlen : NULL;
                real *dub = du + bd * seqlen, *ddb = ddelta + bd * seqlen;
                real *dzb = dz ? dz + bd * seqlen : NULL;
                const real Dv = D ? D[d] : 0.f;
                const real bias = delta_bias ? delta_bias[d] : 0.f;
                real a2[256], An[256], gcar[256], dAacc[256];
                for (int n = 0; n < dstate; ++n) {
                    An[n] = A[(size_t)d * dstate + n]; a2[n] = An[n] * MMU_LOG2E; gcar[n] = 0.f; dAacc[n] = 0.f;
                }
                /* forward recompute, keep every state */
                {
                    real h[256];
                    for (int n = 0; n < dstate; ++n) h[n] = 0.f;
                    for (int t = 0; t < seqlen; ++t) {
                        real dl = db[t] + bias;
                        if (delta_softplus) dl = softplus_thr(dl);
                        dls[t] = dl;
                        const real du_ = dl * ub[t];
                        for (int n = 0; n < dstate; ++n) {
                            h[n] = R_EXP2(dl * a2[n]) * h[n] + du_ * Bb[(size_t)n * seqlen + t];
                            hs[(size_t)t * dstate + n] = h[n];
                        }
                    }
                }
                real dDacc = 0.f, dbacc = 0.f;
                /* reverse walk: g_t = C_t dy_t + a_{t+1} g_{t+1}; gcar holds a_{t+1} g_{t+1} */
                for (int t = seqlen - 1; t >= 0; --t) {
                    const real dl = dls[t], uv = ub[t];
                    real dy = gb[t];
                    if (zb) {
                        /* selective_scan_bwd_kernel.cuh:186-191 */
                        const real zv = zb[t];
                        const real sg = 1.f / (1.f + R_EXP(-zv));
                        real y = Dv * uv;
                        for (int n = 0; n < dstate; ++n) y += hs[(size_t)t * dstate + n] * Cb[(size_t)n * seqlen + t];
                        if (dzb) dzb[t] = dy * y * sg * (1.f + zv * (1.f - sg));
                        dy *= zv * sg;
                    }
                    real du_t = Dv * dy;
                    real ddl = 0.f;
                    dDacc += dy * uv;
                    for (int n = 0; n < dstate; ++n) {
                        const real Bv = Bb[(size_t)n * seqlen + t], Cv = Cb[(size_t)n * seqlen + t];
                        const real h = hs[(size_t)t * dstate + n];
                        const real a = R_EXP2(dl * a2[n]);
                        const real bt = dl * uv * Bv;
                        const real ahprev = h - bt;                 /* a_t * h_{t-1} */
                        const real gt = Cv * dy + gcar[n];
                        du_t += gt * dl * Bv;
                        ddl += gt * uv * Bv + gt * An[n] * ahprev;
                        dAacc[n] += gt * dl * ahprev;
                        dBb[(size_t)n * seqlen + t] += gt * dl * uv;
                        dCb[(size_t)n * seqlen + t] += dy * h;
                        gcar[n] = a * gt;
                    }
                    dub[t] = du_t;
                    if (delta_softplus) {
                        /* selective_scan_bwd_kernel.cuh:439-453 */
                        const real x = db[t] + bias;
                        ddl = x <= 20.f ? ddl / (1.f + R_EXP(-x)) : ddl;
                    }
                    ddb[t] = ddl;
                    dbacc += ddl;
                }
                for (int n = 0; n < dstate; ++n) pdA[bd * dstate + n] = dAacc[n];
                pdD[bd] = dDacc; pdb[bd] = dbacc;
            }
            free(hs); free(dls);
        }
    }
    for (int d = 0; d < dim; ++d) {
        for (int n = 0; n < dstate; ++n) {
            real s = 0.f;
            for (int b = 0; b < batch; ++b) s += pdA[((size_t)b * dim + d) * dstate + n];
            dA[(size_t)d * dstate + n] = s;
        }
        real s1 = 0.f, s2 = 0.f;
        for (int b = 0; b < batch; ++b) { s1 += pdD[(size_t)b * dim + d]; s2 += pdb[(size_t)b * dim + d]; }
        if (dD) dD[d] = s1;
        if (ddelta_bias) ddelta_bias[d] = s2;
    }
    free(pdA); free(pdD); free(pdb);
}

/* ------------------------------------------------------------------ */
/* depthwise causal conv1d (+ optional SiLU)                           */
/* x, out : [batch][dim][seqlen]; weight : [dim][width]; bias : [dim]  */
/* ------------------------------------------------------------------ */
void mmu_oracle_causal_conv1d_fwd(
    const real *x, const real *weight, const real *bias /*nullable*/, int silu,
    real *out, int batch, int dim, int seqlen, int width)
{
    #pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
        for (int d = 0; d < dim; ++d) {
            const real *xb = x + ((size_t)b * dim + d) * seqlen;
            real *ob = out + ((size_t)b * dim + d) * seqlen;
            const real *w = weight + (size_t)d * width;
            const real bv = bias ? bias[d] : 0.f;
            for (int t = 0; t < seqlen; ++t) {
                /* causal_conv1d_fwd.cu:99-118 : out[t] = bias + sum_w W[w] x[t-(width-1-w)] */
                real p = bv;
                for (int k = 0; k < width; ++k) {
                    const int s = t - (width - 1 - k);
                    if (s >= 0) p += w[k] * xb[s];
                }
                ob[t] = silu ? p / (1.f + R_EXP(-p)) : p;
            }
        }
    }
}

void mmu_oracle_causal_conv1d_bwd(
    const real *x, const real *weight, const real *bias /*nullable*/, const real *dout, int silu,
    real *dx, real *dweight, real *dbias /*nullable*/,
    int batch, int dim, int seqlen, int width)
{
    const size_t nbd = (size_t)batch * dim;
    real *pdw = (real *)calloc(nbd * width, sizeof(real));
    real *pdb = (real *)calloc(nbd, sizeof(real));
    #pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
        for (int d = 0; d < dim; ++d) {
            const size_t bd = (size_t)b * dim + d;
            const real *xb = x + bd * seqlen, *gb = dout + bd * seqlen;
            real *dxb = dx + bd * seqlen;
            const real *w = weight + (size_t)d * width;
            const real bv = bias ? bias[d] : 0.f;
            real *dp = (real *)malloc(sizeof(real) * (size_t)seqlen);
            real dwacc[8] = {0}, dbacc = 0.f;
            for (int t = 0; t < seqlen; ++t) {
                real g = gb[t];
                if (silu) {
                    /* recompute the pre-activation: causal_conv1d_bwd.cu:153-175 */
                    real p = bv;
                    for (int k = 0; k < width; ++k) {
                        const int s = t - (width - 1 - k);
                        if (s >= 0) p += w[k] * xb[s];
                    }
                    const real sg = 1.f / (1.f + R_EXP(-p));
                    g = g * sg * (1.f + p * (1.f - sg));
                }
                dp[t] = g;
                dbacc += g;
                for (int k = 0; k < width; ++k) {
                    const int s = t - (width - 1 - k);
                    if (s >= 0) dwacc[k] += xb[s] * g;
                }
            }
            for (int t = 0; t < seqlen; ++t) {
                /* dx[t] = sum_w W[w] dp[t + width-1-w] */
                real acc = 0.f;
                for (int k = 0; k < width; ++k) {
                    const int s = t + (width - 1 - k);
                    if (s < seqlen) acc += w[k] * dp[s];
                }
                dxb[t] = acc;
            }
            for (int k = 0; k < width; ++k) pdw[bd * width + k] = dwacc[k];
            pdb[bd] = dbacc;
            free(dp);
        }
    }
    for (int d = 0; d < dim; ++d) {
        for (int k = 0; k < width; ++k) {
            real s = 0.f;
            for (int b = 0; b < batch; ++b) s += pdw[((size_t)b * dim + d) * width + k];
            dweight[(size_t)d * width + k] = s;
        }
        if (dbias) {
            real s = 0.f;
            for (int b = 0; b < batch; ++b) s += pdb[(size_t)b * dim + d];
            dbias[d] = s;
        }
    }
    free(pdw); free(pdb);
}
