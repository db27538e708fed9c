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

#define MMU_LOG2E 1.4426950408889634f

int mmu_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* softplus with the kernel's threshold: selective_scan_fwd_kernel.cuh:153-156 */
static inline float softplus_thr(float x) { return x <= 20.f ? log1pf(expf(x)) : x; }

/* ------------------------------------------------------------------ */
/* selective scan forward                                              */
/* ------------------------------------------------------------------ */
void mmu_oracle_selective_scan_fwd(
    const float *u, const float *delta, const float *A, const float *B, const float *C,
    const float *D /*nullable*/, const float *z /*nullable*/, const float *delta_bias /*nullable*/,
    int delta_softplus,
    float *out /*nullable*/, float *out_z /*nullable, needs z*/, float *last_state /*nullable*/,
    int batch, int dim, int seqlen, int dstate, int ngroups)
{
    const int dpg = dim / ngroups;
    #pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
        for (int d = 0; d < dim; ++d) {
            const int g = d / dpg;
            const float *ub = u + ((size_t)b * dim + d) * seqlen;
            const float *db = delta + ((size_t)b * dim + d) * seqlen;
            const float *zb = z ? z + ((size_t)b * dim + d) * seqlen : NULL;
            const float *Bb = B + ((size_t)b * ngroups + g) * dstate * seqlen;
            const float *Cb = C + ((size_t)b * ngroups + g) * dstate * seqlen;
            float *ob = out ? out + ((size_t)b * dim + d) * seqlen : NULL;
            float *ozb = out_z ? out_z + ((size_t)b * dim + d) * seqlen : NULL;
            const float Dv = D ? D[d] : 0.f;
            const float bias = delta_bias ? delta_bias[d] : 0.f;
            float h[256];
            float a2[256];
            for (int n = 0; n < dstate; ++n) { h[n] = 0.f; a2[n] = A[(size_t)d * dstate + n] * MMU_LOG2E; }
            for (int t = 0; t < seqlen; ++t) {
                float dl = db[t] + bias;
                if (delta_softplus) dl = softplus_thr(dl);
                const float uv = ub[t];
                const float du_ = dl * uv;
                float y = Dv * uv;
                for (int n = 0; n < dstate; ++n) {
                    /* exp2f(delta * A * log2e): selective_scan_fwd_kernel.cuh:168-171,216 */
                    const float a = exp2f(dl * a2[n]);
                    h[n] = a * h[n] + du_ * Bb[(size_t)n * seqlen + t];
                    y += h[n] * Cb[(size_t)n * seqlen + t];
                }
                if (ob) ob[t] = y;
                if (ozb) { const float zv = zb[t]; ozb[t] = y * (zv / (1.f + expf(-zv))); }
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
    const float *u, const float *delta, const float *A, const float *B, const float *C,
    const float *D, const float *z, const float *delta_bias, const float *dout,
    int delta_softplus,
    float *du, float *ddelta, float *dA, float *dB, float *dC,
    float *dD /*nullable*/, float *dz /*nullable*/, float *ddelta_bias /*nullable*/,
    int batch, int dim, int seqlen, int dstate, int ngroups)
{
    const int dpg = dim / ngroups;
    const size_t nbd = (size_t)batch * dim;
    /* per-(b,d) partials so that the cross-thread reduction order is fixed */
    float *pdA = (float *)calloc(nbd * dstate, sizeof(float));
    float *pdD = (float *)calloc(nbd, sizeof(float));
    float *pdb = (float *)calloc(nbd, sizeof(float));
    /* dB/dC contributions of one channel: [b][d][n][t] would be huge; instead
       parallelise over (b, g) and walk the channels of the group serially. */
    memset(dB, 0, sizeof(float) * (size_t)batch * ngroups * dstate * seqlen);
    memset(dC, 0, sizeof(float) * (size_t)batch * ngroups * dstate * seqlen);

    #pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
        for (int g = 0; g < ngroups; ++g) {
            float *hs = (float *)malloc(sizeof(float) * (size_t)seqlen * dstate); /* h_t per n */
            float *dls = (float *)malloc(sizeof(float) * (size_t)seqlen);          /* softplus'd delta */
            const float *Bb = B + ((size_t)b * ngroups + g) * dstate * seqlen;
            const float *Cb = C + ((size_t)b * ngroups + g) * dstate * seqlen;
            float *dBb = dB + ((size_t)b * ngroups + g) * dstate * seqlen;
            float *dCb = dC + ((size_t)b * ngroups + g) * dstate * seqlen;
            for (int d = g * dpg; d < (g + 1) * dpg; ++d) {
                const size_t bd = (size_t)b * dim + d;
                const float *ub = u + bd * seqlen, *db = delta + bd * seqlen, *gb = dout + bd * seqlen;
                const float *zb = z ? z + bd * seqlen : NULL;
                float *dub = du + bd * seqlen, *ddb = ddelta + bd * seqlen;
                float *dzb = dz ? dz + bd * seqlen : NULL;
                const float Dv = D ? D[d] : 0.f;
                const float bias = delta_bias ? delta_bias[d] : 0.f;
                float a2[256], An[256], gcar[256], dAacc[256];
                for (int n = 0; n < dstate; ++n) {
                    An[n] = A[(size_t)d * dstate + n]; a2[n] = An[n] * MMU_LOG2E; gcar[n] = 0.f; dAacc[n] = 0.f;
                }
                /* forward recompute, keep every state */
                {
                    float h[256];
                    for (int n = 0; n < dstate; ++n) h[n] = 0.f;
                    for (int t = 0; t < seqlen; ++t) {
                        float dl = db[t] + bias;
                        if (delta_softplus) dl = softplus_thr(dl);
                        dls[t] = dl;
                        const float du_ = dl * ub[t];
                        for (int n = 0; n < dstate; ++n) {
                            h[n] = exp2f(dl * a2[n]) * h[n] + du_ * Bb[(size_t)n * seqlen + t];
                            hs[(size_t)t * dstate + n] = h[n];
                        }
                    }
                }
                float dDacc = 0.f, dbacc = 0.f;
                /* reverse walk: g_t = C_t dy_t + a_{t+1} g_{t+1}; gcar holds a_{t+1} g_{t+1} */
                for (int t = seqlen - 1; t >= 0; --t) {
                    const float dl = dls[t], uv = ub[t];
                    float dy = gb[t];
                    if (zb) {
                        /* selective_scan_bwd_kernel.cuh:186-191 */
                        const float zv = zb[t];
                        const float sg = 1.f / (1.f + expf(-zv));
                        float y = Dv * uv;
                        for (int n = 0; n < dstate; ++n) y += hs[(size_t)t * dstate + n] * Cb[(size_t)n * seqlen + t];
                        if (dzb) dzb[t] = dy * y * sg * (1.f + zv * (1.f - sg));
                        dy *= zv * sg;
                    }
                    float du_t = Dv * dy;
                    float ddl = 0.f;
                    dDacc += dy * uv;
                    for (int n = 0; n < dstate; ++n) {
                        const float Bv = Bb[(size_t)n * seqlen + t], Cv = Cb[(size_t)n * seqlen + t];
                        const float h = hs[(size_t)t * dstate + n];
                        const float a = exp2f(dl * a2[n]);
                        const float bt = dl * uv * Bv;
                        const float ahprev = h - bt;                 /* a_t * h_{t-1} */
                        const float gt = Cv * dy + gcar[n];
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
                        const float x = db[t] + bias;
                        ddl = x <= 20.f ? ddl / (1.f + expf(-x)) : ddl;
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
            float s = 0.f;
            for (int b = 0; b < batch; ++b) s += pdA[((size_t)b * dim + d) * dstate + n];
            dA[(size_t)d * dstate + n] = s;
        }
        float s1 = 0.f, s2 = 0.f;
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
    const float *x, const float *weight, const float *bias /*nullable*/, int silu,
    float *out, int batch, int dim, int seqlen, int width)
{
    #pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
        for (int d = 0; d < dim; ++d) {
            const float *xb = x + ((size_t)b * dim + d) * seqlen;
            float *ob = out + ((size_t)b * dim + d) * seqlen;
            const float *w = weight + (size_t)d * width;
            const float bv = bias ? bias[d] : 0.f;
            for (int t = 0; t < seqlen; ++t) {
                /* causal_conv1d_fwd.cu:99-118 : out[t] = bias + sum_w W[w] x[t-(width-1-w)] */
                float p = bv;
                for (int k = 0; k < width; ++k) {
                    const int s = t - (width - 1 - k);
                    if (s >= 0) p += w[k] * xb[s];
                }
                ob[t] = silu ? p / (1.f + expf(-p)) : p;
            }
        }
    }
}

void mmu_oracle_causal_conv1d_bwd(
    const float *x, const float *weight, const float *bias /*nullable*/, const float *dout, int silu,
    float *dx, float *dweight, float *dbias /*nullable*/,
    int batch, int dim, int seqlen, int width)
{
    const size_t nbd = (size_t)batch * dim;
    float *pdw = (float *)calloc(nbd * width, sizeof(float));
    float *pdb = (float *)calloc(nbd, sizeof(float));
    #pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
        for (int d = 0; d < dim; ++d) {
            const size_t bd = (size_t)b * dim + d;
            const float *xb = x + bd * seqlen, *gb = dout + bd * seqlen;
            float *dxb = dx + bd * seqlen;
            const float *w = weight + (size_t)d * width;
            const float bv = bias ? bias[d] : 0.f;
            float *dp = (float *)malloc(sizeof(float) * (size_t)seqlen);
            float dwacc[8] = {0}, dbacc = 0.f;
            for (int t = 0; t < seqlen; ++t) {
                float g = gb[t];
                if (silu) {
                    /* recompute the pre-activation: causal_conv1d_bwd.cu:153-175 */
                    float p = bv;
                    for (int k = 0; k < width; ++k) {
                        const int s = t - (width - 1 - k);
                        if (s >= 0) p += w[k] * xb[s];
                    }
                    const float sg = 1.f / (1.f + expf(-p));
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
                float acc = 0.f;
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
            float s = 0.f;
            for (int b = 0; b < batch; ++b) s += pdw[((size_t)b * dim + d) * width + k];
            dweight[(size_t)d * width + k] = s;
        }
        if (dbias) {
            float s = 0.f;
            for (int b = 0; b < batch; ++b) s += pdb[(size_t)b * dim + d];
            dbias[d] = s;
        }
    }
    free(pdw); free(pdb);
}
