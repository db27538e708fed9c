// tri_fused.hip -- f2: the token orders of the tri-directional ("v3") Mamba block folded into its causal conv1d and
// its gate, so that neither a re-ordered copy of xz nor three gated scan outputs ever exist.
//
// requirements/mamba_simple.py:212-270 runs mamba_inner (selective_scan_interface.py:155-289) three times over the same
// xz = (x, z): as is, token-reversed and slice-interleaved (token i of slice s -> position i*nslices + s), and adds the
// three results after undoing the orders.  Two facts make the copies unnecessary:
//   * only the conv1d reads x: one kernel reads every x row ONCE and writes the three conv1d(+SiLU) outputs, each in
//     its own scan order (1 read + 3 writes of a [dim][batch][L] tensor, where the re-ordering pass + three conv1d passes
//     moved 12);
//   * the gate is the same z in every direction, and out_dir = y_dir * silu(z_dir) is elementwise in the direction's order:
//     out_f + unflip(out_b) + unslice(out_s) = silu(z) * (y_f + unflip(y_b) + unslice(y_s)).  The three scans run
//     without z (two / three streams fewer each way) and one kernel sums, re-orders and gates.
// The backward mirrors both: d total -> (dz, dy_f, dy_b, dy_s in scan order) in one pass; the three d conv_out -> dx and
// the conv weight / bias gradients of the three directions in one pass over x.
//
// Tiling: a workgroup takes TI = 64 positions i of EVERY slice of one (batch, channel) row: nslices runs of 64
// contiguous tokens in the natural order (= the flipped order read backwards) and ONE run of 64 * nslices contiguous
// positions in the slice order; the transpose between the two happens in LDS (odd row pitch: conflict-free both ways).
// The conv taps reach 3 tokens back in each order: the natural neighbours are three halo columns on either side of a
// run, the slice-order neighbours are slices s-1..s-3 at the same i, wrapping to the top slices at i-1.
// float32 or bfloat16 activations (float32 arithmetic, weights and weight gradients), conv width 4, 4 <= nslices <= 64.
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

constexpr int TI = 64;       // positions per slice and tile
constexpr int TP = TI + 7;   // LDS row pitch: columns 0..TI+5 hold positions i0-3 .. i0+TI+2 (odd pitch)
constexpr int TGP = TI + 1;  // gate kernels: no halo

struct TcArgs {
    int batch, dim, L, ns, Ls, ntiles, nblk;
    const void *x;            // activations: io_t (float or bf16_t); weights, biases, partial sums: float
    long x_bs, x_ds;
    const float *w[3], *b[3];
    void *out[3];
    const void *g[3];
    void *dx;
    long dx_bs, dx_ds;
    float *ws[3];   // per direction [batch][dim][nblk][5]
};

struct TgArgs {
    int batch, dim, L, ns, Ls;
    const void *z;
    long z_bs, z_ds;
    const void *y[3];
    void *out;
    long out_bs, out_ds;
    const void *dout;
    long dout_bs, dout_ds;
    void *dz;
    long dz_bs, dz_ds;
    void *dy[3];
};

__device__ __forceinline__ float silu_(float a) { return a * sigmoidf_(a); }
__device__ __forceinline__ float dsilu_(float a) {
    const float sg = sigmoidf_(a);
    return sg * (1.f + a * (1.f - sg));
}

struct Taps {
    float w[3][4], b[3];
};
__device__ __forceinline__ void load_taps(const TcArgs &p, int d, Taps &t) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int m = 0; m < 4; ++m) t.w[k][m] = p.w[k][d * 4 + m];
        t.b[k] = p.b[k] ? p.b[k][d] : 0.f;
    }
}

// x positions i0-3 .. i0+TI+2 of every slice (whatever token sits there: a run's neighbours in memory are its
// neighbours in the natural order, also across a slice boundary); tokens outside [0, L) read as 0.
// KS slices per wave are loaded before the first LDS store (all of them when nslices is a template constant): a
// workgroup with 18-73 KB of LDS has 2-8 waves per SIMD, one exposed memory round trip per slice is what it cannot hide.
template <int KS, typename io_t>
__device__ __forceinline__ void stage_x(const io_t *__restrict__ xr, float *xt, int ns, int Ls, int L, int i0, int tx, int ry) {
    for (int sb = ry; sb < ns; sb += 4 * KS) {
        // (raw values in the registers, converted where they are consumed: a conversion next to its load is a wait on it)
        io_t a[KS], h[KS];
        const io_t zero = from_f32<io_t>(0.f);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int sl = sb + 4 * k;
            const int t0 = sl * Ls + i0 - 3;
            const int t = t0 + tx, th = t0 + 64 + tx;
            a[k] = (sl < ns && t >= 0 && t < L) ? xr[t] : zero;
            h[k] = (sl < ns && tx < 6 && th >= 0 && th < L) ? xr[th] : zero;
        }
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int sl = sb + 4 * k;
            if (sl < ns) {
                xt[sl * TP + tx] = to_f32(a[k]);
                if (tx < 6) xt[sl * TP + 64 + tx] = to_f32(h[k]);
            }
        }
    }
}

// conv1d pre-activation of the slice order at (slice s, column c = 3 + il of the staged tile; i = i0 + il)
__device__ __forceinline__ float slice_pre(const float *xt, const float (&w)[4], float bias, int ns, int s, int il, int i0) {
    float a = bias;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        int sp = s - 3 + m, col = 3 + il;
        bool none = false;
        if (sp < 0) {
            sp += ns;
            col -= 1;
            none = (i0 + il == 0);   // position q - (3 - m) < 0
        }
        const float v = xt[sp * TP + col];
        a = fmaf(w[m], none ? 0.f : v, a);
    }
    return a;
}

template <int NS, typename io_t>
__global__ __launch_bounds__(256) void tri_conv_fwd_kernel(TcArgs p) {
    extern __shared__ float lds[];
    const int ns = NS > 0 ? NS : p.ns;
    const int Ls = p.Ls, L = p.L;
    const int row = blockIdx.y;
    const int d = row / p.batch, b = row - d * p.batch;
    const int i0 = blockIdx.x * TI;
    const int tx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const io_t *xr = (const io_t *)p.x + (long)b * p.x_bs + (long)d * p.x_ds;
    const long ob = (long)row * L;
    constexpr int KS = NS > 0 ? NS / 4 : 1;
    Taps tp;
    load_taps(p, d, tp);
    stage_x<KS, io_t>(xr, lds, ns, Ls, L, i0, tx, ry);
    __syncthreads();
    const int ni = Ls - i0 < TI ? Ls - i0 : TI;
    io_t *of = (io_t *)p.out[0] + ob, *obk = (io_t *)p.out[1] + ob, *os = (io_t *)p.out[2] + ob + (long)i0 * ns;
    if (tx < ni) {
#pragma unroll 4
        for (int sl = ry; sl < ns; sl += 4) {
            const float *r = lds + sl * TP + tx;
            float v[7];
#pragma unroll
            for (int m = 0; m < 7; ++m) v[m] = r[m];
            float af = tp.b[0], ab = tp.b[1];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                af = fmaf(tp.w[0][m], v[m], af);            // x[t-3+m]
                ab = fmaf(tp.w[1][3 - m], v[3 + m], ab);    // x_flip[p'-3+(3-m)] = x[t+m]
            }
            const int t = sl * Ls + i0 + tx;
            of[t] = from_f32<io_t>(silu_(af));
            obk[L - 1 - t] = from_f32<io_t>(silu_(ab));
        }
    }
    for (int j = threadIdx.x; j < ni * ns; j += 256) {
        const int il = j / ns, s = j - il * ns;
        os[j] = from_f32<io_t>(silu_(slice_pre(lds, tp.w[2], tp.b[2], ns, s, il, i0)));
    }
}

// A workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... of one row and keeps the 15 weight-gradient sums
// (3 directions x (4 taps + bias)) in registers across them; the block sums are plain stores into the workspace and are
// added in a fixed order afterwards (causal_conv1d.hip's scheme: bit-reproducible, nothing to zero).
template <int NS, typename io_t>
__global__ __launch_bounds__(256, NS > 0 ? 2 : 1) void tri_conv_bwd_kernel(TcArgs p) {
    extern __shared__ float lds[];
    __shared__ float red[4][15];
    const int ns = NS > 0 ? NS : p.ns;
    const int Ls = p.Ls, L = p.L;
    const int row = blockIdx.y;
    const int d = row / p.batch, b = row - d * p.batch;
    const int tx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const io_t *xr = (const io_t *)p.x + (long)b * p.x_bs + (long)d * p.x_ds;
    io_t *dxr = (io_t *)p.dx + (long)b * p.dx_bs + (long)d * p.dx_ds;
    const long ob = (long)row * L;
    const io_t *gf = (const io_t *)p.g[0] + ob, *gb = (const io_t *)p.g[1] + ob, *gs = (const io_t *)p.g[2] + ob;
    float *xt = lds, *df = xt + ns * TP, *db = df + ns * TP, *dsl = db + ns * TP;
    Taps tp;
    load_taps(p, d, tp);
    constexpr int KS = NS > 0 ? NS / 4 : 16;                       // slices per wave (nslices <= 64)
    constexpr int JP = NS > 0 ? ((TI + 1) * NS + 255) / 256 : 17;  // slice-order items per thread
    float acc[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) acc[k] = 0.f;
    // One tile's global loads (x with its halos, the three dout tiles) go into registers and are issued a whole tile
    // ahead: the loads of tile k+1 are in flight under the gradient phase of tile k (2 workgroups per CU at 64 slices:
    // nothing else hides a memory round trip).
    // (raw values in the registers, converted where they are consumed: a conversion next to its load is a wait on it)
    const io_t zero = from_f32<io_t>(0.f);
    io_t xa[KS], xh[2], gfa[KS], gba[KS], gfe = zero, gbe = zero, gsa[JP];
    const int esl = threadIdx.x / 3, ec = TI + (int)threadIdx.x - 3 * esl;   // the 3 extra columns of every slice
    auto issue = [&](int i0) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int sl = ry + 4 * k;
            const int tl = sl * Ls + i0 - 3 + tx;
            xa[k] = (sl < ns && tl >= 0 && tl < L) ? xr[tl] : zero;
            // natural order at positions i0 .. i0+TI+2, flipped order at the tokens i0-3 .. i0+TI-1: column c of df / db
            const int t = sl * Ls + i0 + tx, tb = t - 3;
            gfa[k] = (sl < ns && t < L) ? gf[t] : zero;
            gba[k] = (sl < ns && tb >= 0 && tb < L) ? gb[L - 1 - tb] : zero;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {   // x columns 64..69 of every slice
            const int j = threadIdx.x + 256 * k, sl = j / 6;
            const int th = sl * Ls + i0 - 3 + 64 + (j - 6 * sl);
            xh[k] = (j < 6 * ns && th >= 0 && th < L) ? xr[th] : zero;
        }
        if (threadIdx.x < 3 * ns) {
            const int t = esl * Ls + i0 + ec, tb = t - 3;
            gfe = t < L ? gf[t] : zero;
            gbe = (tb >= 0 && tb < L) ? gb[L - 1 - tb] : zero;
        }
        // slice order at i0 .. i0+TI (column c of dsl)
#pragma unroll
        for (int k = 0; k < JP; ++k) {
            const int j = threadIdx.x + 256 * k;
            gsa[k] = (j < (TI + 1) * ns && (long)i0 * ns + j < L) ? gs[(long)i0 * ns + j] : zero;
        }
    };
    issue((int)blockIdx.x * TI);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int i0 = tile * TI;
        const int ni = Ls - i0 < TI ? Ls - i0 : TI;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int sl = ry + 4 * k;
            if (sl < ns) xt[sl * TP + tx] = to_f32(xa[k]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = threadIdx.x + 256 * k, sl = j / 6;
            if (j < 6 * ns) xt[sl * TP + 64 + (j - 6 * sl)] = to_f32(xh[k]);
        }
        __syncthreads();
        // dp = dout * silu'(pre): both natural-order tiles read the same window x[c .. c+3] of the staged row
        auto dp_fb = [&](int sl, int c, float g_f, float g_b) {
            const float *r = xt + sl * TP + c;
            const float v0 = r[0], v1 = r[1], v2 = r[2], v3 = r[3];
            const float pf = fmaf(tp.w[0][3], v3, fmaf(tp.w[0][2], v2, fmaf(tp.w[0][1], v1, fmaf(tp.w[0][0], v0, tp.b[0]))));
            const float pb = fmaf(tp.w[1][0], v3, fmaf(tp.w[1][1], v2, fmaf(tp.w[1][2], v1, fmaf(tp.w[1][3], v0, tp.b[1]))));
            df[sl * TP + c] = g_f * dsilu_(pf);
            db[sl * TP + c] = g_b * dsilu_(pb);
        };
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int sl = ry + 4 * k;
            if (sl < ns) dp_fb(sl, tx, to_f32(gfa[k]), to_f32(gba[k]));
        }
        if (threadIdx.x < 3 * ns) dp_fb(esl, ec, to_f32(gfe), to_f32(gbe));
#pragma unroll
        for (int k = 0; k < JP; ++k) {
            const int j = threadIdx.x + 256 * k;
            if (j < (TI + 1) * ns) {
                const int c = j / ns, s = j - c * ns;
                dsl[s * TP + c] = to_f32(gsa[k]) * dsilu_(slice_pre(xt, tp.w[2], tp.b[2], ns, s, c, i0));
            }
        }
        if (tile + (int)gridDim.x < p.ntiles) issue((tile + (int)gridDim.x) * TI);
        __syncthreads();
        if (tx < ni) {
#pragma unroll 4
            for (int sl = ry; sl < ns; sl += 4) {
                const float *rx = xt + sl * TP + tx, *rf = df + sl * TP + tx, *rb = db + sl * TP + tx;
                float a = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    a = fmaf(tp.w[0][m], rf[3 - m], a);   // natural: dp(t + 3 - m)
                    a = fmaf(tp.w[1][m], rb[m], a);       // flipped: dp at token t - (3 - m)
                    int sp = sl + 3 - m, c = tx;          // sliced: dp(q + 3 - m)
                    if (sp >= ns) {
                        sp -= ns;
                        c += 1;
                    }
                    a = fmaf(tp.w[2][m], dsl[sp * TP + c], a);
                }
                dxr[sl * Ls + i0 + tx] = from_f32<io_t>(a);
                const float dpf = rf[0], dpb = rb[3], dps = dsl[sl * TP + tx];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[m] = fmaf(rx[m], dpf, acc[m]);                 // x[t-3+m]
                    acc[5 + 3 - m] = fmaf(rx[3 + m], dpb, acc[5 + 3 - m]);   // W_b[3-m] multiplies x[t+m]
                    int sp = sl - 3 + m, col = 3 + tx;
                    bool none = false;
                    if (sp < 0) {
                        sp += ns;
                        col -= 1;
                        none = (i0 + tx == 0);
                    }
                    const float xv = xt[sp * TP + col];
                    acc[10 + m] = fmaf(none ? 0.f : xv, dps, acc[10 + m]);
                }
                acc[4] += dpf;
                acc[9] += dpb;
                acc[14] += dps;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const float s = wave_sum(acc[k]);
        if (tx == 0) red[ry][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 15) {
        const int k = threadIdx.x / 5, m = threadIdx.x - 5 * k;
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        p.ws[k][((((long)b * p.dim + d) * gridDim.x) + blockIdx.x) * 5 + m] = s;
    }
}

// dW[k][d][m], db[k][d] = sums over (batch, block) of the partials in a fixed order.  grid (dim, 3), block 64.
struct TrArgs {
    const float *ws[3];
    float *dw[3], *dbias[3];
    int batch, dim, nblk;
};
__global__ __launch_bounds__(64) void tri_wgrad_reduce_kernel(TrArgs p) {
    const int d = blockIdx.x, k = blockIdx.y, lane = threadIdx.x;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int n = p.batch * p.nblk;
    for (int i = lane; i < n; i += 64) {
        const int b = i / p.nblk, q = i - b * p.nblk;
        const float *r = p.ws[k] + ((((long)b * p.dim + d) * p.nblk) + q) * 5;
#pragma unroll
        for (int m = 0; m < 5; ++m) acc[m] += r[m];
    }
    const float v4 = wave_sum4(acc[0], acc[1], acc[2], acc[3]);
    const float vb = wave_sum(acc[4]);
    if (lane >= 12 && lane < 16) p.dw[k][(long)d * 4 + lane - 12] = v4;
    if (lane == 0 && p.dbias[k]) p.dbias[k][d] = vb;
}

template <int NS, typename io_t>
__global__ __launch_bounds__(256) void tri_gate_fwd_kernel(TgArgs p) {
    __shared__ float tile[64 * TGP];
    const int ns = NS > 0 ? NS : p.ns;
    const int Ls = p.Ls, L = p.L;
    const int row = blockIdx.y;
    const int d = row / p.batch, b = row - d * p.batch;
    const int i0 = blockIdx.x * TI;
    const int tx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int ni = Ls - i0 < TI ? Ls - i0 : TI;
    const long ob = (long)row * L;
    const io_t *src = (const io_t *)p.y[2] + ob + (long)i0 * ns;
    for (int j = threadIdx.x; j < ni * ns; j += 256) {
        const int il = j / ns, s = j - il * ns;
        tile[s * TGP + il] = to_f32(src[j]);
    }
    __syncthreads();
    if (tx >= ni) return;
    const io_t *zr = (const io_t *)p.z + (long)b * p.z_bs + (long)d * p.z_ds;
    const io_t *yf = (const io_t *)p.y[0] + ob, *yb = (const io_t *)p.y[1] + ob;
    io_t *o = (io_t *)p.out + (long)b * p.out_bs + (long)d * p.out_ds;
    for (int sl = ry; sl < ns; sl += 4) {
        const int t = sl * Ls + i0 + tx;
        const float zz = to_f32(zr[t]);
        o[t] = from_f32<io_t>((to_f32(yf[t]) + to_f32(yb[L - 1 - t]) + tile[sl * TGP + tx]) * silu_(zz));
    }
}

template <int NS, typename io_t>
__global__ __launch_bounds__(256) void tri_gate_bwd_kernel(TgArgs p) {
    __shared__ float tile[64 * TGP];
    const int ns = NS > 0 ? NS : p.ns;
    const int Ls = p.Ls, L = p.L;
    const int row = blockIdx.y;
    const int d = row / p.batch, b = row - d * p.batch;
    const int i0 = blockIdx.x * TI;
    const int tx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int ni = Ls - i0 < TI ? Ls - i0 : TI;
    const long ob = (long)row * L;
    const io_t *src = (const io_t *)p.y[2] + ob + (long)i0 * ns;
    for (int j = threadIdx.x; j < ni * ns; j += 256) {
        const int il = j / ns, s = j - il * ns;
        tile[s * TGP + il] = to_f32(src[j]);
    }
    __syncthreads();
    if (tx < ni) {
        const io_t *zr = (const io_t *)p.z + (long)b * p.z_bs + (long)d * p.z_ds;
        const io_t *gr = (const io_t *)p.dout + (long)b * p.dout_bs + (long)d * p.dout_ds;
        io_t *dzr = (io_t *)p.dz + (long)b * p.dz_bs + (long)d * p.dz_ds;
        const io_t *yf = (const io_t *)p.y[0] + ob, *yb = (const io_t *)p.y[1] + ob;
        io_t *df = (io_t *)p.dy[0] + ob, *db = (io_t *)p.dy[1] + ob;
        for (int sl = ry; sl < ns; sl += 4) {
            const int t = sl * Ls + i0 + tx;
            const float zz = to_f32(zr[t]), g = to_f32(gr[t]);
            const float sum = to_f32(yf[t]) + to_f32(yb[L - 1 - t]) + tile[sl * TGP + tx];
            const float sg = sigmoidf_(zz);
            dzr[t] = from_f32<io_t>(g * sum * sg * (1.f + zz * (1.f - sg)));
            const float dy = g * zz * sg;
            df[t] = from_f32<io_t>(dy);
            db[L - 1 - t] = from_f32<io_t>(dy);
            tile[sl * TGP + tx] = dy;    // (a slot is read and written by its own thread only in this phase)
        }
    }
    __syncthreads();
    io_t *dst = (io_t *)p.dy[2] + ob + (long)i0 * ns;
    for (int j = threadIdx.x; j < ni * ns; j += 256) {
        const int il = j / ns, s = j - il * ns;
        dst[j] = from_f32<io_t>(tile[s * TGP + il]);
    }
}

inline int tri_bwd_blocks(int rows, int ntiles) {
    long per_row = (4096 + rows - 1) / rows;
    if (per_row < 1) per_row = 1;
    return (int)(per_row < ntiles ? per_row : ntiles);
}

int tc_check(const mmu_tri_conv_params *p, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->dtype == MMU_DTYPE_F32 || p->dtype == MMU_DTYPE_BF16, "%s: float32 or bfloat16 (got dtype %d)", name, p->dtype);
    MMU_CHECK(p->batch > 0 && p->dim > 0 && p->seqlen > 0, "%s: empty tensor", name);
    MMU_CHECK(p->nslices >= 4 && p->nslices <= 64, "%s: 4..64 slices (got %d)", name, p->nslices);
    MMU_CHECK(p->seqlen % p->nslices == 0, "%s: seqlen %d must be divisible by nslices %d", name, p->seqlen, p->nslices);
    MMU_CHECK((long)p->batch * p->dim < 65536, "%s: more than 65535 rows", name);
    MMU_CHECK(p->x && p->weight_f && p->weight_b && p->weight_s, "%s: x and the three weights are required", name);
    return 0;
}

void tc_fill(const mmu_tri_conv_params *p, TcArgs &a) {
    a.batch = p->batch; a.dim = p->dim; a.L = p->seqlen; a.ns = p->nslices; a.Ls = p->seqlen / p->nslices;
    a.ntiles = (a.Ls + TI - 1) / TI;
    a.x = p->x; a.x_bs = p->x_bs; a.x_ds = p->x_ds;
    a.w[0] = p->weight_f; a.w[1] = p->weight_b; a.w[2] = p->weight_s;
    a.b[0] = p->bias_f; a.b[1] = p->bias_b; a.b[2] = p->bias_s;
    a.out[0] = p->out_f; a.out[1] = p->out_b; a.out[2] = p->out_s;
    a.g[0] = p->dout_f; a.g[1] = p->dout_b; a.g[2] = p->dout_s;
    a.dx = p->dx; a.dx_bs = p->dx_bs; a.dx_ds = p->dx_ds;
}

int tg_check(const mmu_tri_gate_params *p, const char *name) {
    MMU_CHECK(p != nullptr, "%s: null params", name);
    MMU_CHECK(p->dtype == MMU_DTYPE_F32 || p->dtype == MMU_DTYPE_BF16, "%s: float32 or bfloat16 (got dtype %d)", name, p->dtype);
    MMU_CHECK(p->batch > 0 && p->dim > 0 && p->seqlen > 0, "%s: empty tensor", name);
    MMU_CHECK(p->nslices >= 1 && p->nslices <= 64, "%s: 1..64 slices (got %d)", name, p->nslices);
    MMU_CHECK(p->seqlen % p->nslices == 0, "%s: seqlen %d must be divisible by nslices %d", name, p->seqlen, p->nslices);
    MMU_CHECK((long)p->batch * p->dim < 65536, "%s: more than 65535 rows", name);
    MMU_CHECK(p->z && p->y_f && p->y_b && p->y_s, "%s: z and the three scan outputs are required", name);
    return 0;
}

void tg_fill(const mmu_tri_gate_params *p, TgArgs &a) {
    a.batch = p->batch; a.dim = p->dim; a.L = p->seqlen; a.ns = p->nslices; a.Ls = p->seqlen / p->nslices;
    a.z = p->z; a.z_bs = p->z_bs; a.z_ds = p->z_ds;
    a.y[0] = p->y_f; a.y[1] = p->y_b; a.y[2] = p->y_s;
    a.dy[0] = p->dy_f; a.dy[1] = p->dy_b; a.dy[2] = p->dy_s;
    a.out = p->out; a.out_bs = p->out_bs; a.out_ds = p->out_ds;
    a.dout = p->dout; a.dout_bs = p->dout_bs; a.dout_ds = p->dout_ds;
    a.dz = p->dz; a.dz_bs = p->dz_bs; a.dz_ds = p->dz_ds;
}

#define TRI_NS_(ns, NS, ...)                             \
    switch (ns) {                                        \
    case 16: { constexpr int NS = 16; __VA_ARGS__; } break; \
    case 32: { constexpr int NS = 32; __VA_ARGS__; } break; \
    case 64: { constexpr int NS = 64; __VA_ARGS__; } break; \
    default: { constexpr int NS = 0; __VA_ARGS__; } break;  \
    }
// (slice count, element type) -> template arguments NS, io_t
#define TRI_NS(ns, NS, ...)                                               \
    if (p->dtype == MMU_DTYPE_BF16) {                                     \
        using io_t = bf16_t;                                              \
        TRI_NS_(ns, NS, __VA_ARGS__)                                      \
    } else {                                                              \
        using io_t = float;                                               \
        TRI_NS_(ns, NS, __VA_ARGS__)                                      \
    }

unsigned long long g_bwd_lds[8];

}  // namespace

extern "C" int mmu_tri_conv_fwd(const mmu_tri_conv_params *p, void *stream) {
    if (int r = tc_check(p, "tri_conv_fwd")) return r;
    MMU_CHECK(p->out_f && p->out_b && p->out_s, "tri_conv_fwd: the three outputs are required");
    TcArgs a = {};
    tc_fill(p, a);
    const int lds = a.ns * TP * 4;
    dim3 grid(a.ntiles, p->batch * p->dim);
    TRI_NS(a.ns, NS, tri_conv_fwd_kernel<NS, io_t><<<grid, 256, lds, (hipStream_t)stream>>>(a));
    MMU_HIP_LAUNCH_CHECK("tri_conv_fwd");
    return 0;
}

extern "C" size_t mmu_tri_conv_bwd_workspace_floats(int batch, int dim, int seqlen, int nslices) {
    if (nslices <= 0) return 0;
    const int ntiles = (seqlen / nslices + TI - 1) / TI;
    return (size_t)3 * batch * dim * tri_bwd_blocks(batch * dim, ntiles) * 5;
}

extern "C" int mmu_tri_conv_bwd(const mmu_tri_conv_params *p, void *stream) {
    if (int r = tc_check(p, "tri_conv_bwd")) return r;
    MMU_CHECK(p->dout_f && p->dout_b && p->dout_s && p->dx && p->workspace && p->dweight_f && p->dweight_b &&
                  p->dweight_s,
              "tri_conv_bwd: the three dout, dx, the three dweight and the workspace are required");
    TcArgs a = {};
    tc_fill(p, a);
    a.nblk = tri_bwd_blocks(p->batch * p->dim, a.ntiles);
    const size_t per = (size_t)p->batch * p->dim * a.nblk * 5;
    for (int k = 0; k < 3; ++k) a.ws[k] = p->workspace + k * per;
    const int lds = 4 * a.ns * TP * 4;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(a.nblk, p->batch * p->dim);
    int slot = a.ns == 16 ? 1 : a.ns == 32 ? 2 : a.ns == 64 ? 3 : 0;
    hipError_t e = hipSuccess;
    TRI_NS(a.ns, NS, {
        if (lds > 64 * 1024) e = mmu_set_lds_once(tri_conv_bwd_kernel<NS, io_t>, lds, g_bwd_lds[slot + (p->dtype == MMU_DTYPE_BF16 ? 4 : 0)]);
        if (e == hipSuccess) tri_conv_bwd_kernel<NS, io_t><<<grid, 256, lds, st>>>(a);
    });
    MMU_CHECK(e == hipSuccess, "tri_conv_bwd: %s", hipGetErrorString(e));
    MMU_HIP_LAUNCH_CHECK("tri_conv_bwd");
    // the ordered sums of the per-block partials: three jobs of causal_conv1d's kind inside a deferred scope
    float *const dw[3] = {p->dweight_f, p->dweight_b, p->dweight_s}, *const dbs[3] = {p->dbias_f, p->dbias_b, p->dbias_s};
    bool deferred = true;
    for (int k = 0; k < 3 && deferred; ++k) {
        const long job[8] = {2, (long)a.ws[k], (long)dw[k], (long)dbs[k], p->batch, p->dim, (long)a.nblk, 4};
        const bool ok = mmu_defer_job(job);
        MMU_CHECK(ok || k == 0, "tri_conv_bwd: the deferred scope closed between two of the three jobs");
        deferred = ok;
    }
    if (!deferred) {
        TrArgs r = {};
        for (int k = 0; k < 3; ++k) { r.ws[k] = a.ws[k]; r.dw[k] = dw[k]; r.dbias[k] = dbs[k]; }
        r.batch = p->batch; r.dim = p->dim; r.nblk = a.nblk;
        tri_wgrad_reduce_kernel<<<dim3(p->dim, 3), 64, 0, st>>>(r);
        MMU_HIP_LAUNCH_CHECK("tri_conv_bwd(reduce)");
    }
    return 0;
}

extern "C" int mmu_tri_gate_fwd(const mmu_tri_gate_params *p, void *stream) {
    if (int r = tg_check(p, "tri_gate_fwd")) return r;
    MMU_CHECK(p->out, "tri_gate_fwd: out is required");
    TgArgs a = {};
    tg_fill(p, a);
    dim3 grid((a.Ls + TI - 1) / TI, p->batch * p->dim);
    TRI_NS(a.ns, NS, tri_gate_fwd_kernel<NS, io_t><<<grid, 256, 0, (hipStream_t)stream>>>(a));
    MMU_HIP_LAUNCH_CHECK("tri_gate_fwd");
    return 0;
}

extern "C" int mmu_tri_gate_bwd(const mmu_tri_gate_params *p, void *stream) {
    if (int r = tg_check(p, "tri_gate_bwd")) return r;
    MMU_CHECK(p->dout && p->dz && p->dy_f && p->dy_b && p->dy_s, "tri_gate_bwd: dout, dz and the three dy are required");
    TgArgs a = {};
    tg_fill(p, a);
    dim3 grid((a.Ls + TI - 1) / TI, p->batch * p->dim);
    TRI_NS(a.ns, NS, tri_gate_bwd_kernel<NS, io_t><<<grid, 256, 0, (hipStream_t)stream>>>(a));
    MMU_HIP_LAUNCH_CHECK("tri_gate_bwd");
    return 0;
}
