// deferred_reduce.hip -- the final sums of the weight gradients of a backward pass in ONE launch, for gfx950.
//
// Where it sits: the weight gradients of MM-UNet's projections / DSC convolutions (gemm_nt_splitk.hip), offset
// convolutions (conv3x3_small.hip) and causal conv1d layers (causal_conv1d.hip) are produced as per-workgroup partial
// sums plus a small kernel that adds them in a fixed order -- 131 such kernels per training step, each a few workgroups
// of work at the ~4.6 us dependent-launch floor in the middle of the backward chain (0.65 ms of a 37 ms step).  Nothing
// reads a weight gradient before the optimizer, so between mmu_deferred_begin() and mmu_deferred_launch() those launchers
// only RECORD their reduction (a row of eight int64: kind, partials, destinations, shape) and one kernel runs them all,
// with the same summation order as the kernels it replaces (bit-identical results).  The caller keeps the partial
// buffers alive until then and provides the device copies of the job table and of the (job, workgroup) work list
// (mm-unet_amd/deferred.py: inside a captured graph the table is written after the capture).
#include <mutex>
#include <vector>
#include "mmu_common.h"
#include "../../include/mmunet_amd.h"

namespace {

std::mutex g_mu;
bool g_active = false;
std::vector<long> g_rows;   // 8 per job

constexpr int ROW = 8;
#define MMU_DEFER_REP 8

// The kinds with 64 results per block of 1,024 threads (0, 1, 3, 6) take REP consecutive blocks per workgroup and load
// for all of them before reducing any: at one block per workgroup the launch was 86,000 workgroups each waiting for one
// or two dependent loads (440 us for 0.1 ms of traffic).  The order of the additions per result is unchanged.
constexpr int REP = MMU_DEFER_REP;

// kind 0: gemm_nt_splitk slab sums.  {0, part, c, -, n, slabs, rows << 32 | cols, transpose}
__device__ void job_gemm_nt(const long *row, int blk, float (*sums)[16][64]) {
    const float *part = reinterpret_cast<const float *>(row[1]);
    float *c = reinterpret_cast<float *>(row[2]);
    const long n = row[4];
    const int slabs = (int)row[5], rows = (int)(row[6] >> 32), cols = (int)(row[6] & 0xffffffff);
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    if ((n & 3) == 0 && REP == 8) {
        // four consecutive results per lane (16-byte loads: a quarter of the load instructions), two such groups per thread;
        // every result still adds its slabs k = g, g + 16, ... in the same order
        float4 (*sums4)[16][64] = reinterpret_cast<float4 (*)[16][64]>(&sums[0][0][0]);   // [2][16][64] float4 = the same 32 KB
        const long b4 = (long)blk * (64 * REP) + o * 4;
        float4 s4[2];
        s4[0] = s4[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = g;
        for (; k + 48 < slabs; k += 64) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const long i = b4 + r * 256;
                if (i < n) {
                    const float4 v0 = *reinterpret_cast<const float4 *>(part + (long)k * n + i);
                    const float4 v1 = *reinterpret_cast<const float4 *>(part + (long)(k + 16) * n + i);
                    const float4 v2 = *reinterpret_cast<const float4 *>(part + (long)(k + 32) * n + i);
                    const float4 v3 = *reinterpret_cast<const float4 *>(part + (long)(k + 48) * n + i);
                    s4[r].x += v0.x; s4[r].x += v1.x; s4[r].x += v2.x; s4[r].x += v3.x;
                    s4[r].y += v0.y; s4[r].y += v1.y; s4[r].y += v2.y; s4[r].y += v3.y;
                    s4[r].z += v0.z; s4[r].z += v1.z; s4[r].z += v2.z; s4[r].z += v3.z;
                    s4[r].w += v0.w; s4[r].w += v1.w; s4[r].w += v2.w; s4[r].w += v3.w;
                }
            }
        }
        for (; k < slabs; k += 16) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const long i = b4 + r * 256;
                if (i < n) {
                    const float4 v = *reinterpret_cast<const float4 *>(part + (long)k * n + i);
                    s4[r].x += v.x; s4[r].y += v.y; s4[r].z += v.z; s4[r].w += v.w;
                }
            }
        }
        sums4[0][g][o] = s4[0];
        sums4[1][g][o] = s4[1];
        __syncthreads();
        if (g < 2) {
            const long i = b4 + g * 256;
            if (i < n) {
                float4 t = sums4[g][0][o];
#pragma unroll
                for (int q = 1; q < 16; ++q) {
                    const float4 v = sums4[g][q][o];
                    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
                }
                const float tv[4] = {t.x, t.y, t.z, t.w};
                if (row[7]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const long ii = i + j, r = ii / cols, q = ii - r * cols;
                        c[q * rows + r] = tv[j];
                    }
                } else {
                    *reinterpret_cast<float4 *>(c + i) = t;
                }
            }
        }
        return;
    }
    const long base = (long)blk * (64 * REP) + o;
    float s[REP];
#pragma unroll
    for (int r = 0; r < REP; ++r) s[r] = 0.f;
    int k = g;
    for (; k + 48 < slabs; k += 64) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const long i = base + r * 64;
            if (i < n) {
                const float v0 = part[(long)k * n + i], v1 = part[(long)(k + 16) * n + i];
                const float v2 = part[(long)(k + 32) * n + i], v3 = part[(long)(k + 48) * n + i];
                s[r] += v0; s[r] += v1; s[r] += v2; s[r] += v3;
            }
        }
    }
    for (; k < slabs; k += 16) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const long i = base + r * 64;
            if (i < n) s[r] += part[(long)k * n + i];
        }
    }
#pragma unroll
    for (int r = 0; r < REP; ++r) sums[r][g][o] = s[r];
    __syncthreads();
    const long i = base + g * 64;
    if (g < REP && i < n) {
        float t = sums[g][0][o];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += sums[g][q][o];
        if (row[7]) {
            const long r = i / cols, q = i - r * cols;
            c[q * rows + r] = t;
        } else {
            c[i] = t;
        }
    }
}

// kind 1: conv3x3_small weight / bias gradient.  {1, part, dW, dbias, Cin, nblk, CO, -}; 64 results per block
__device__ void job_conv3x3s(const long *row, int blk) {
    const float *part = reinterpret_cast<const float *>(row[1]);
    float *dW = reinterpret_cast<float *>(row[2]);
    float *dbias = reinterpret_cast<float *>(row[3]);
    const int Cin = (int)row[4], nblk = (int)row[5], CO = (int)row[6];
    const int NV = CO * 10, NV4 = (NV + 3) & ~3;
    const int sub = threadIdx.x & 15;
    int idx[REP];
    bool live[REP];
    float s[REP];
#pragma unroll
    for (int r = 0; r < REP; ++r) {
        idx[r] = (blk * REP + r) * 64 + (threadIdx.x >> 4);
        live[r] = idx[r] < Cin * NV4;
        idx[r] = live[r] ? idx[r] : Cin * NV4 - 1;
        s[r] = 0.f;
    }
    int k = sub;
    for (; k + 48 < nblk; k += 64) {     // four partial rows in flight per result, added in the same order
        float v[4][REP];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < REP; ++r) v[u][r] = part[(long)(k + 16 * u) * Cin * NV4 + idx[r]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < REP; ++r) s[r] += v[u][r];
    }
    for (; k < nblk; k += 16) {
#pragma unroll
        for (int r = 0; r < REP; ++r) s[r] += part[(long)k * Cin * NV4 + idx[r]];
    }
#pragma unroll
    for (int r = 0; r < REP; ++r) {
        float v = s[r];
        v += dpp_mov<MMU_DPP_ROW_SHR(1), 0xf>(0.f, v);
        v += dpp_mov<MMU_DPP_ROW_SHR(2), 0xf>(0.f, v);
        v += dpp_mov<MMU_DPP_ROW_SHR(4), 0xf>(0.f, v);
        v += dpp_mov<MMU_DPP_ROW_SHR(8), 0xf>(0.f, v);   // lane 15 of each row of 16 holds the total
        const int ci = idx[r] / NV4, e = idx[r] - ci * NV4;
        if (!live[r] || sub != 15 || e >= NV) continue;
        if (e < CO * 9) {
            const int co = e / 9, t = e - co * 9;
            dW[((long)co * Cin + ci) * 9 + t] = v;
        } else if (ci == 0 && dbias != nullptr) {
            dbias[e - CO * 9] = v;
        }
    }
}

// kind 2: causal conv1d weight / bias gradient.  {2, ws, dweight, dbias, batch, dim, nblk, width}; a wave per channel
__device__ void job_conv1d(const long *row, int blk) {
    const float *ws = reinterpret_cast<const float *>(row[1]);
    float *dweight = reinterpret_cast<float *>(row[2]);
    float *dbias = reinterpret_cast<float *>(row[3]);
    const int batch = (int)row[4], dim = (int)row[5], nblk = (int)row[6], width = (int)row[7];
    const int lane = threadIdx.x & 63;
    const int d = blk * 16 + (threadIdx.x >> 6);
    if (d >= dim) return;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int n = batch * nblk;
    for (int i = lane; i < n; i += 64) {
        const int b = i / nblk, k = i - b * nblk;
        const float *r = ws + ((((long)b * dim + d) * nblk) + k) * 5;
#pragma unroll
        for (int m = 0; m < 5; ++m) acc[m] += r[m];
    }
    const float v4 = wave_sum4(acc[0], acc[1], acc[2], acc[3]);
    const float vb = wave_sum(acc[4]);
    if (lane >= 12 && lane < 16) {
        const int k = (lane - 12) - (4 - width);
        if (k >= 0) dweight[(long)d * width + k] = v4;
    }
    if (lane == 0 && dbias) dbias[d] = vb;
}

// kind 3: rows of a partial buffer.  {3, part, dst, dst2, n1, nparts, stride, ntot}: dst[i < n1], dst2[i - n1] for the rest
__device__ void job_rows(const long *row, int blk, float (*sums)[16][64]) {
    const float *part = reinterpret_cast<const float *>(row[1]);
    float *dst = reinterpret_cast<float *>(row[2]);
    float *dst2 = reinterpret_cast<float *>(row[3]);
    const int n1 = (int)row[4], nparts = (int)row[5], stride = (int)row[6], ntot = (int)row[7];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int base = blk * (64 * REP) + o;
    float s[REP];
#pragma unroll
    for (int r = 0; r < REP; ++r) s[r] = 0.f;
    int k = g;
    for (; k + 48 < nparts; k += 64) {   // four partial rows in flight per result, added in the same order
        float v[4][REP];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < REP; ++r)
                v[u][r] = base + r * 64 < ntot ? part[(long)(k + 16 * u) * stride + base + r * 64] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < REP; ++r) s[r] += v[u][r];
    }
    for (; k < nparts; k += 16) {
#pragma unroll
        for (int r = 0; r < REP; ++r)
            if (base + r * 64 < ntot) s[r] += part[(long)k * stride + base + r * 64];
    }
#pragma unroll
    for (int r = 0; r < REP; ++r) sums[r][g][o] = s[r];
    __syncthreads();
    const int i = base + g * 64;
    if (g < REP && i < ntot) {
        float t = sums[g][0][o];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += sums[g][q][o];
        if (i < n1)
            dst[i] = t;
        else if (dst2 != nullptr)
            dst2[i - n1] = t;
    }
}

// kind 6: weight gradient of the matrix-core convolutions from its per-workgroup tile partials (conv3x3_wgrad_sum_kernel of
// conv3x3_wgrad_mfma.hip, conv_s2_wgrad_sum_kernel of conv_s2_mfma.hip; same order: 16 lanes per output stride over the
// partials, row-local DPP sum).  {6, ws, dW, Cin | Cout << 32, K (0: the 3 x 3 stride-1 layout), n_cic, wg_per_cc,
// TCO | TCI << 32}; 1,024 tile elements per workgroup.
// Walked in the PARTIALS' order (a thread = one element of a tile, consecutive threads consecutive elements: coalesced; the
// output-ordered form read one float per 64 KB-strided line: 147 us for 150 MB), with the additions in the order of the
// kernels it replaces: sixteen class sums k = c, c + 16, ... and their row-shift tree.
__device__ void job_conv_tiles(const long *row, int blk) {
    const float *ws = reinterpret_cast<const float *>(row[1]);
    float *dW = reinterpret_cast<float *>(row[2]);
    const int Cin = (int)(row[3] & 0xffffffff), Cout = (int)(row[3] >> 32), K = (int)row[4];
    const int n_cic = (int)row[5], wg_per_cc = (int)row[6];
    const int TCO = (int)(row[7] & 0xffffffff), TCI = (int)(row[7] >> 32);
    const int S = K ? 4 : 9;
    const unsigned tile = (unsigned)TCO * TCI * S;
    const unsigned total = (unsigned)n_cic * (Cout / TCO) * tile;
    const unsigned g = (unsigned)blk * 1024 + threadIdx.x;
    if (g >= total) return;
    const unsigned cc = g / tile, e = g - cc * tile;
    const float *src = ws + (long)cc * wg_per_cc * tile + e;
    float a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = 0.f;
    int k = 0;
    for (; k + 16 <= wg_per_cc; k += 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] += src[(long)(k + c) * tile];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (k + c < wg_per_cc) a[c] += src[(long)(k + c) * tile];
    // lane 15 of the row-shift sum of sixteen lanes holding a[0..15] (s += shr1, shr2, shr4, shr8)
    const float x1 = a[1] + a[0], x3 = a[3] + a[2], x5 = a[5] + a[4], x7 = a[7] + a[6];
    const float x9 = a[9] + a[8], x11 = a[11] + a[10], x13 = a[13] + a[12], x15 = a[15] + a[14];
    const float y3 = x3 + x1, y7 = x7 + x5, y11 = x11 + x9, y15 = x15 + x13;
    const float z7 = y7 + y3, z15 = y15 + y11;
    const float sum = z15 + z7;
    // where the element goes
    const unsigned cot = cc / (unsigned)n_cic, vt = cc - cot * n_cic;
    const unsigned col = e / (unsigned)(TCI * S), rem = e - col * (TCI * S);
    const unsigned vl = rem / (unsigned)S, sidx = rem - vl * S;
    const unsigned co = cot * TCO + col, v = vt * TCI + vl;
    if (K == 0) {
        dW[((long)co * Cin + v) * 9 + sidx] = sum;
    } else {   // kh = 2a + 1 - py, kw = 2b + 1 - px with (py, px) = the phase of virtual channel v (conv_s2_mfma.hip)
        const unsigned ph = v / (unsigned)Cin, c = v - ph * Cin;
        const int kh = 2 * (int)(sidx >> 1) + 1 - (int)(ph >> 1), kw = 2 * (int)(sidx & 1) + 1 - (int)(ph & 1);
        if (kh < K && kw < K) dW[(((long)co * Cin + c) * K + kh) * K + kw] = sum;
    }
}

// kind 7: a wave per output over strided partial rows (conv7_wsum_kernel of conv7x7_small.hip;
// pw1_sum_kernel of pointwise_one.hip: results from n1 on go to dst2).
// {7, part, dst, dst2 or 0, n_out, nparts, stride, n1 (0: all to dst)}; 16 outputs per workgroup
__device__ void job_wave_rows(const long *row, int blk) {
    const float *part = reinterpret_cast<const float *>(row[1]);
    float *dst = reinterpret_cast<float *>(row[2]);
    float *dst2 = reinterpret_cast<float *>(row[3]);
    const int n_out = (int)row[4], nparts = (int)row[5], stride = (int)row[6], n1 = (int)row[7];
    const int lane = threadIdx.x & 63;
    const int i = blk * 16 + (threadIdx.x >> 6);
    if (i >= n_out) return;
    float s = 0.f;
    for (int k = lane; k < nparts; k += 64) s += part[(long)k * stride + i];
    s = wave_sum(s);
    if (lane == 0) {
        if (n1 == 0 || i < n1)
            dst[i] = s;
        else
            dst2[i - n1] = s;
    }
}

// kinds 4 / 5: the selective scan's dA / dD / d delta_bias from its per-(batch, tile) partials -- reduce_partials_w8 (+
// reduce_slices_w8) of selective_scan_bwd_w8.hip and reduce_partials (+ reduce_slices) of selective_scan.hip, same order:
// per slice of 512 rows eight strided row sums met in order, then the slices in order.  One workgroup per channel d; its
// four quarters take four slices at a time.
//   {4, part8, dA, dD, dbias, BT, dim, A or 0}: part8[bt][d][wave 0..7][4] = (dA[2w], dA[2w+1], dD share, dbias share)
//   {5, part,  dA, dD, dbias, BC, dim | N << 32, A or 0}: part[bc][d][N + 2] = (dA[0..N), dD, dbias)
// A given: dA is stored multiplied by A (the gradient of A_log, see mmu_scan_bwd_params.dA_times_A).
__device__ void job_scan(const long *row, int d, float *red /* 1024 */, float *tot /* 32 */, bool w8) {
    const float *part = reinterpret_cast<const float *>(row[1]);
    float *dA = reinterpret_cast<float *>(row[2]);
    float *dD = reinterpret_cast<float *>(row[3]);
    float *dbias = reinterpret_cast<float *>(row[4]);
    const int BT = (int)row[5], dim = (int)(row[6] & 0xffffffff);
    const int N = w8 ? 16 : (int)(row[6] >> 32), M = w8 ? 32 : N + 2;
    const float *Asc = reinterpret_cast<const float *>(row[7]);
    const int n_slices = (BT + 511) / 512;
    const int lt = threadIdx.x & 255, sw = threadIdx.x >> 8;
    const int jl = lt & 31, r = lt >> 5;
    for (int j0 = 0; j0 < M; j0 += 32) {
        const int j = j0 + jl;
        float total = 0.f;
        for (int s0 = 0; s0 < n_slices; s0 += 4) {
            const int sl = s0 + sw;
            float s = 0.f;
            if (sl < n_slices && j < M) {
                const int r0 = sl * 512, r1 = min(r0 + 512, BT);
                int bt = r0 + r;
                for (; bt + 56 < r1; bt += 64) {        // eight loads in flight, added in the same order
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = part[((long)(bt + 8 * u) * dim + d) * M + j];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += v[u];
                }
                for (; bt < r1; bt += 8) s += part[((long)bt * dim + d) * M + j];
            }
            red[threadIdx.x] = s;
            __syncthreads();
            if (threadIdx.x < 32) {
                for (int q = 0; q < 4 && s0 + q < n_slices; ++q) {
                    float t = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) t += red[q * 256 + k * 32 + threadIdx.x];
                    total = (s0 + q == 0) ? t : total + t;
                }
            }
            __syncthreads();
        }
        if (w8) {
            if (threadIdx.x < 32) tot[threadIdx.x] = total;
        } else if (threadIdx.x < 32) {
            if (j < N)
                dA[(long)d * N + j] = Asc ? total * Asc[(long)d * N + j] : total;
            else if (j == N) {
                if (dD) dD[d] = total;
            } else if (j == N + 1) {
                if (dbias) dbias[d] = total;
            }
        }
    }
    if (!w8) return;
    __syncthreads();
    const int j = threadIdx.x;
    if (j < 16) {
        const float v = tot[(j >> 1) * 4 + (j & 1)];
        dA[(long)d * 16 + j] = Asc ? v * Asc[(long)d * 16 + j] : v;
    } else if (j < 18) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) s += tot[wv * 4 + 2 + (j - 16)];
        float *o = j == 16 ? dD : dbias;
        if (o) o[d] = s;
    }
}

__global__ __launch_bounds__(1024) void deferred_reduce_kernel(const long *__restrict__ table, const int *__restrict__ work) {
    __shared__ float sums[REP][16][64];
    __shared__ float tot[32];
    const int job = work[2 * blockIdx.x], blk = work[2 * blockIdx.x + 1];
    const long *row = table + (long)job * ROW;
    const int kind = (int)row[0];   // workgroup-uniform
    if (kind == 0)
        job_gemm_nt(row, blk, sums);
    else if (kind == 1)
        job_conv3x3s(row, blk);
    else if (kind == 2)
        job_conv1d(row, blk);
    else if (kind == 3)
        job_rows(row, blk, sums);
    else if (kind == 4 || kind == 5)
        job_scan(row, blk, &sums[0][0][0], tot, kind == 4);
    else if (kind == 6)
        job_conv_tiles(row, blk);
    else if (kind == 7)
        job_wave_rows(row, blk);
}

}  // namespace

// ---- used by the launchers of the three kernel families (declared in mmu_common.h)
bool mmu_defer_job(const long (&row)[8]) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_active) return false;
    g_rows.insert(g_rows.end(), row, row + ROW);
    return true;
}

extern "C" void mmu_deferred_begin(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_active = true;
    g_rows.clear();
}

extern "C" void mmu_deferred_pause(int paused) {   // a call whose result is read right away (dtype conversion ...)
    std::lock_guard<std::mutex> lk(g_mu);
    g_active = paused == 0;
}

extern "C" void mmu_deferred_end(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_active = false;
    g_rows.clear();
}

extern "C" int mmu_deferred_jobs(int64_t *rows_out, int max_jobs) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int n = (int)(g_rows.size() / ROW);
    if (rows_out != nullptr) {
        const int m = n < max_jobs ? n : max_jobs;
        for (long i = 0; i < (long)m * ROW; ++i) rows_out[i] = g_rows[i];
    }
    return n;
}

// Workgroups deferred_reduce_kernel needs for one recorded job (its row as mmu_deferred_jobs returns it): the block
// decomposition of each kind lives HERE, next to the job_* bodies that index by it -- the host builds its (job, block)
// work list from this, not from a copy of the formulas.
extern "C" int mmu_deferred_job_workgroups(const int64_t *r) {
    if (r == nullptr) return 0;
    long nb;
    switch ((int)r[0]) {
        case 0: nb = (r[4] + 63) / 64; break;                                  // gemm_nt slab sums: 64 results per block
        case 1: nb = (r[4] * ((r[6] * 10 + 3) & ~3L) + 63) / 64; break;        // conv3x3_small: padded rows of 10 taps
        case 2: nb = (r[5] + 15) / 16; break;                                  // conv1d-style: 16 channels per workgroup
        case 4: case 5: nb = r[6] & 0xffffffffL; break;                        // scan parameters: a workgroup per channel
        case 6: {                                                              // conv tile partials: a thread per element
            const long tco = r[7] & 0xffffffffL, tci = r[7] >> 32;
            nb = (r[5] * ((r[3] >> 32) / tco) * tco * tci * (r[4] ? 4 : 9) + 1023) / 1024;
            break;
        }
        case 7: nb = (r[4] + 15) / 16; break;                                  // wave rows
        default: nb = (r[7] + 63) / 64; break;                                 // (3) row sums
    }
    if (r[0] == 0 || r[0] == 1 || r[0] == 3) nb = (nb + REP - 1) / REP;        // REP blocks of 64 results per workgroup
    return (int)nb;
}

extern "C" int mmu_deferred_launch(const int64_t *table, const int32_t *work, int n_work, void *stream) {
    MMU_CHECK(table != nullptr && work != nullptr && n_work > 0, "deferred_launch: table, work list and a positive count are required");
    static_assert(sizeof(long) == sizeof(int64_t), "the table is read as long");
    deferred_reduce_kernel<<<n_work, 1024, 0, (hipStream_t)stream>>>(reinterpret_cast<const long *>(table), work);
    MMU_HIP_LAUNCH_CHECK("deferred_launch");
    return 0;
}
