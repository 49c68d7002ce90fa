// Register-tiled variants of the shape-generic fp32 kernels for the common filter widths
// (Pw in {3,5,7,9}, stride in {1,2}; any C for analysis, C in {1,3} for synthesis; 2-D and 3-D).
// Each thread owns 4 consecutive output columns, so one LDS row window of 3*s + Pw values feeds
// 4*Pw FMAs per channel and the filter row arrives as one wave-uniform (scalar) vector load.
// cdl_analysis / cdl_synthesis (cdl_generic.hip) try these first and fall back to the untiled
// kernels for every other shape.
#include "cdl_common.h"

namespace {

constexpr int TX = 64, TY = 16;   // output tile (columns x rows) of a 256-thread workgroup
constexpr int PXT = 4;            // columns per thread
constexpr int MCH = 8;            // code channels per register pass

// ------------------------------------------------------------------------------------------
template <int PW, int SW>
__global__ __launch_bounds__(256) void k_analysis_t(cdl_geom g, const float *__restrict__ x,
                                                    const float *__restrict__ w, float alpha,
                                                    const float *__restrict__ zin,
                                                    const float *__restrict__ gate,
                                                    const float *__restrict__ tau,
                                                    float *__restrict__ out, int tilesX, int tilesY,
                                                    int PH, int PWp, int mper, cdl_prox_args px)
{
    extern __shared__ float patch[];                       // [C][Pd][PH][PWp]
    constexpr int WL = (PXT - 1) * SW + PW;                // row window per thread
    const int Dz = g.D / g.sd, Hz = g.H / g.sh, Wz = g.W / g.sw;
    int b = blockIdx.x;
    const int tx = b % tilesX; b /= tilesX;
    const int ty = b % tilesY; b /= tilesY;
    const int zd = b;
    const int n = blockIdx.y;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int zy = ty * TY + ly, zx0 = tx * TX + lx * PXT;
    const int y0 = ty * TY * g.sh - g.ph, x0 = tx * TX * SW - g.pw, d0 = zd * g.sd - g.pd;

    const int plane = PH * PWp, pvol = g.C * g.Pd * plane;
    for (int i = threadIdx.x; i < pvol; i += 256) {
        const int px = i % PWp;
        int r = i / PWp;
        const int py = r % PH; r /= PH;
        const int kd = r % g.Pd, c = r / g.Pd;
        const int d = d0 + kd, yy = y0 + py, xx = x0 + px;
        float v = 0.0f;
        if (d >= 0 && d < g.D && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W)
            v = x[((((size_t)n * g.C + c) * g.D + d) * g.H + yy) * g.W + xx];
        patch[i] = v;
    }
    __syncthreads();

    const int taps = g.Pd * g.Ph * PW, wrow = g.C * taps;
    const float *pbase = patch + (ly * g.sh) * PWp + lx * PXT * SW;
    const int m_lo = blockIdx.z * mper, m_hi = min(g.M, m_lo + mper);   // channel slice of this workgroup
    for (int m0 = m_lo; m0 < m_hi; m0 += MCH) {
        float acc[MCH][PXT];
#pragma unroll
        for (int j = 0; j < MCH; ++j)
#pragma unroll
            for (int p = 0; p < PXT; ++p) acc[j][p] = 0.0f;
        for (int c = 0; c < g.C; ++c)
            for (int kd = 0; kd < g.Pd; ++kd)
                for (int ki = 0; ki < g.Ph; ++ki) {
                    const float *prow = pbase + ((c * g.Pd + kd) * PH + ki) * PWp;
                    float win[WL];
#pragma unroll
                    for (int i = 0; i < WL; ++i) win[i] = prow[i];
                    const int wofs = c * taps + (kd * g.Ph + ki) * PW;
#pragma unroll
                    for (int j = 0; j < MCH; ++j) {
                        const float *wr = w + (size_t)min(m0 + j, g.M - 1) * wrow + wofs;   // wave-uniform
#pragma unroll
                        for (int kj = 0; kj < PW; ++kj) {
                            const float wv = wr[kj];
#pragma unroll
                            for (int p = 0; p < PXT; ++p) acc[j][p] = fmaf(win[p * SW + kj], wv, acc[j][p]);
                        }
                    }
                }
        if (zy < Hz) {
#pragma unroll
            for (int j = 0; j < MCH; ++j) {
                const int m = m0 + j;
                if (m >= g.M) continue;
                const float t = tau ? tau[n * g.M + m] : 0.0f;
                const size_t rowi = ((((size_t)n * g.M + m) * Dz + zd) * Hz + zy) * Wz;
#pragma unroll
                for (int p = 0; p < PXT; ++p) {
                    const int zx = zx0 + p;
                    if (zx >= Wz) continue;
                    float base = 0.0f;
                    if (zin) {
                        base = zin[rowi + zx];
                        if (gate && gate[rowi + zx] == 0.0f) base = 0.0f;
                    }
                    const float u = fmaf(alpha, acc[j][p], base);
                    out[rowi + zx] = px.zp ? cdl_prox_apply(px, u, rowi + zx, n * g.M + m) : (tau ? cdl_shrink(u, t) : u);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// synthesis: thread owns out[c0..c0+CC)[d][y][x0..x0+3]; code channels pass through LDS MCHS (template:
// 8 in 2-D, 2 in 3-D where the Pd-fold more rounds per workgroup want more workgroups per CU) at a
// time and one code depth slice at a time; rows of one wave share the stride phase.
template <int PW, int SW, int CC, int MCHS>
__global__ __launch_bounds__(256) void k_synthesis_t(cdl_geom g, const float *__restrict__ z,
                                                     const float *__restrict__ gate,
                                                     const float *__restrict__ w, float alpha,
                                                     const float *__restrict__ mask,
                                                     const float *__restrict__ sub,
                                                     float *__restrict__ out, int tilesX, int tilesY,
                                                     int PZH, int PZW, int mper, float *__restrict__ partial)
{
    extern __shared__ float patch[];                       // [MCHS][PZH][PZW]
    const int Dz = g.D / g.sd, Hz = g.H / g.sh, Wz = g.W / g.sw;
    int b = blockIdx.x;
    const int tx = b % tilesX; b /= tilesX;
    const int ty = b % tilesY; b /= tilesY;
    const int d = b;
    const int n = blockIdx.y;
    const int lx = threadIdx.x & 15, lw = threadIdx.x >> 6, lr = (threadIdx.x >> 4) & 3;
    const int ly = lw + 4 * lr;                            // rows of a wave are congruent mod 4
    const int y = ty * TY + ly, x0 = tx * TX + lx * PXT;
    const int pw = g.pw;
    // code-patch origin of this tile (floor division, may be negative: zero filled)
    const int zy_lo = cdl_floordiv(ty * TY + g.ph - (g.Ph - 1), g.sh);
    const int zx_lo = cdl_floordiv(tx * TX + pw - (PW - 1), SW);
    const int plane = PZH * PZW;
    const int taps = g.Pd * g.Ph * PW;
    // Row window of one thread.  With B0 = x0 + pw - (PW-1) (numerator of p = 0, kj = PW-1) the tap
    // (p, kj) has numerator B0 + t, t = p + PW-1 - kj, and is live when SW divides it; its code column
    // is zx_first + idx(t).  x0 is a multiple of 4 and pw = PW/2, so B0's parity is a constant of PW.
    constexpr int B0ODD = (SW == 2) ? (((PW - 1) / 2) & 1) : 0;
    constexpr int WLS = (PXT - 1 + PW - 1 + B0ODD) / SW + 1;
    float acc[CC][PXT];
    for (int c0 = 0; c0 < g.C; c0 += CC) {
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
#pragma unroll
            for (int p = 0; p < PXT; ++p) acc[cc][p] = 0.0f;
        const int m_lo = blockIdx.z * mper, m_hi = min(g.M, m_lo + mper);   // channel slice (split launches)
        for (int m0 = m_lo; m0 < m_hi; m0 += MCHS) {
            for (int kd = 0; kd < g.Pd; ++kd) {
                const int td = d + g.pd - kd;
                if (td < 0 || td % g.sd) continue;          // uniform
                const int zd = td / g.sd;
                if (zd >= Dz) continue;
                __syncthreads();
                for (int i = threadIdx.x; i < MCHS * plane; i += 256) {
                    const int px = i % PZW;
                    int r = i / PZW;
                    const int py = r % PZH, mm = r / PZH;
                    const int m = m0 + mm, zy = zy_lo + py, zx = zx_lo + px;
                    float v = 0.0f;
                    if (m < m_hi && zy >= 0 && zy < Hz && zx >= 0 && zx < Wz) {
                        const size_t idx = ((((size_t)n * g.M + m) * Dz + zd) * Hz + zy) * Wz + zx;
                        v = z[idx];
                        if (gate && gate[idx] == 0.0f) v = 0.0f;
                    }
                    patch[i] = v;
                }
                __syncthreads();
                const int mlim = min(MCHS, m_hi - m0);
                for (int ki = 0; ki < g.Ph; ++ki) {
                    const int tyy = y + g.ph - ki + g.sh * g.Ph;      // shifted positive
                    if (tyy % g.sh) continue;                          // same for every row of the wave
                    const int zyl = tyy / g.sh - g.Ph - zy_lo;
                    // first code column of the window, relative to the patch
                    const int zx_first = cdl_floordiv(x0 + pw - (PW - 1), SW) - zx_lo;
                    const int wbase = (kd * g.Ph + ki) * PW;
                    for (int mm = 0; mm < mlim; ++mm) {
                        const float *prow = patch + mm * plane + zyl * PZW + zx_first;
                        float win[WLS];
#pragma unroll
                        for (int i = 0; i < WLS; ++i) win[i] = prow[i];
#pragma unroll
                        for (int cc = 0; cc < CC; ++cc) {
                            if (c0 + cc >= g.C) continue;
                            const float *wr = w + ((size_t)(m0 + mm) * g.C + c0 + cc) * taps + wbase;   // uniform
#pragma unroll
                            for (int kj = 0; kj < PW; ++kj) {
                                const float wv = wr[kj];
#pragma unroll
                                for (int p = 0; p < PXT; ++p) {
                                    constexpr int dummy = 0; (void)dummy;
                                    const int t = p + PW - 1 - kj;                  // compile-time after unrolling
                                    if (SW == 1 || ((t & 1) == B0ODD))
                                        acc[cc][p] = fmaf(win[SW == 1 ? t : (t + B0ODD) / 2], wv, acc[cc][p]);
                                }
                            }
                        }
                    }
                }
            }
        }
        if (y < g.H) {
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                const int c = c0 + cc;
                if (c >= g.C) continue;
                const size_t rowi = ((((size_t)n * g.C + c) * g.D + d) * g.H + y) * g.W;
#pragma unroll
                for (int p = 0; p < PXT; ++p) {
                    const int xo = x0 + p;
                    if (xo >= g.W) continue;
                    if (partial) {          // split launch: raw channel-slice sum, folded by k_synth_fold
                        partial[(size_t)blockIdx.z * ((size_t)g.N * g.C * g.D * g.H * g.W) + rowi + xo] = acc[cc][p];
                        continue;
                    }
                    float v = alpha * acc[cc][p];
                    if (mask) v *= mask[rowi + xo];
                    if (sub) v -= sub[rowi + xo];
                    out[rowi + xo] = v;
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// synthesis, software-pipelined: same thread mapping and arithmetic as k_synthesis_t, but the code patch of
// the NEXT (channel pair, kd) round is fetched into registers while the current round is multiplied, and
// the two divisions per patch element are done once per thread instead of once per round (the element ->
// (channel, row, column) map does not change between rounds).  k_synthesis_t runs ~30-170 rounds per
// workgroup, each an exposed global -> LDS fill between two barriers: that, not the FMAs, was its time.
constexpr int MQ = 2, QNE = 16;       // channels per round; patch elements per thread (register prefetch)

template <int PW, int SW, int CC>
__global__ __launch_bounds__(256) void k_synthesis_q(cdl_geom g, const float *__restrict__ z,
                                                     const float *__restrict__ gate,
                                                     const float *__restrict__ w, float alpha,
                                                     const float *__restrict__ mask,
                                                     const float *__restrict__ sub,
                                                     float *__restrict__ out, int tilesX, int tilesY,
                                                     int PZH, int PZW, int mper, float *__restrict__ partial)
{
    extern __shared__ float patch[];                       // [MQ][PZH][PZW]
    const int Dz = g.D / g.sd, Hz = g.H / g.sh, Wz = g.W / g.sw;
    int b = blockIdx.x;
    const int tx = b % tilesX; b /= tilesX;
    const int ty = b % tilesY; b /= tilesY;
    const int d = b;
    const int n = blockIdx.y;
    const int lx = threadIdx.x & 15, lw = threadIdx.x >> 6, lr = (threadIdx.x >> 4) & 3;
    const int ly = lw + 4 * lr;                            // rows of a wave are congruent mod 4
    const int y = ty * TY + ly, x0 = tx * TX + lx * PXT;
    const int pw = g.pw;
    const int zy_lo = cdl_floordiv(ty * TY + g.ph - (g.Ph - 1), g.sh);
    const int zx_lo = cdl_floordiv(tx * TX + pw - (PW - 1), SW);
    const int plane = PZH * PZW, nel = MQ * plane;
    const int taps = g.Pd * g.Ph * PW;
    const int slab = Dz * Hz * Wz;                         // one channel of one sample (host checks MQ * slab < 2^31)
    constexpr int B0ODD = (SW == 2) ? (((PW - 1) / 2) & 1) : 0;
    constexpr int WLS = (PXT - 1 + PW - 1 + B0ODD) / SW + 1;
    const int zx_first = cdl_floordiv(x0 + pw - (PW - 1), SW) - zx_lo;

    // element table of this thread: offset inside the (channel pair, depth) slab, or -1 outside the code image
    int off[QNE];
    unsigned ch1 = 0;                                      // bit e: element e is in the second channel of the pair
#pragma unroll
    for (int e = 0; e < QNE; ++e) {
        const int i = threadIdx.x + 256 * e;
        off[e] = -1;
        if (i < nel) {
            const int px = i % PZW;
            const int r = i / PZW;
            const int py = r % PZH, mm = r / PZH;
            const int zy = zy_lo + py, zx = zx_lo + px;
            if (zy >= 0 && zy < Hz && zx >= 0 && zx < Wz) off[e] = mm * slab + zy * Wz + zx;
            if (mm) ch1 |= 1u << e;
        }
    }
    const int m_lo = blockIdx.z * mper, m_hi = min(g.M, m_lo + mper);   // channel slice (split launches)
    auto advance = [&](int &m0, int &kd, int &zd) -> bool {             // next (pair, kd) round with a live depth tap
        for (;;) {
            if (++kd >= g.Pd) {
                kd = 0;
                m0 += MQ;
            }
            if (m0 >= m_hi) return false;
            const int td = d + g.pd - kd;
            if (td < 0 || td % g.sd) continue;
            zd = td / g.sd;
            if (zd < Dz) return true;
        }
    };
    auto issue = [&](int m0, int zd, float (&pf)[QNE]) {
        const size_t base = (((size_t)n * g.M + m0) * Dz + zd) * (size_t)Hz * Wz;
        const bool two = m_hi - m0 > 1;
#pragma unroll
        for (int e = 0; e < QNE; ++e) {
            const bool ok = off[e] >= 0 && (two || !((ch1 >> e) & 1u));
            const size_t idx = base + (ok ? off[e] : 0);
            float v = z[idx];
            if (gate && gate[idx] == 0.0f) v = 0.0f;
            pf[e] = ok ? v : 0.0f;
        }
    };

    float acc[CC][PXT];
    float pf[QNE];
    for (int c0 = 0; c0 < g.C; c0 += CC) {
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
#pragma unroll
            for (int p = 0; p < PXT; ++p) acc[cc][p] = 0.0f;
        int m0 = m_lo, kd = -1, zd = 0;
        bool have = advance(m0, kd, zd);
        if (have) issue(m0, zd, pf);
        while (have) {
            __syncthreads();                               // the previous round's readers are done
#pragma unroll
            for (int e = 0; e < QNE; ++e) {
                const int i = threadIdx.x + 256 * e;
                if (i < nel) patch[i] = pf[e];
            }
            __syncthreads();
            int nm0 = m0, nkd = kd, nzd = zd;
            const bool nhave = advance(nm0, nkd, nzd);
            if (nhave) issue(nm0, nzd, pf);                // in flight during the multiply-adds below
            const int mlim = min(MQ, m_hi - m0);
            for (int ki = 0; ki < g.Ph; ++ki) {
                const int tyy = y + g.ph - ki + g.sh * g.Ph;      // shifted positive
                if (tyy % g.sh) continue;                          // same for every row of the wave
                const int zyl = tyy / g.sh - g.Ph - zy_lo;
                const int wbase = (kd * g.Ph + ki) * PW;
                for (int mm = 0; mm < mlim; ++mm) {
                    const float *prow = patch + mm * plane + zyl * PZW + zx_first;
                    float win[WLS];
#pragma unroll
                    for (int i = 0; i < WLS; ++i) win[i] = prow[i];
#pragma unroll
                    for (int cc = 0; cc < CC; ++cc) {
                        if (c0 + cc >= g.C) continue;
                        const float *wr = w + ((size_t)(m0 + mm) * g.C + c0 + cc) * taps + wbase;   // uniform
#pragma unroll
                        for (int kj = 0; kj < PW; ++kj) {
                            const float wv = wr[kj];
#pragma unroll
                            for (int p = 0; p < PXT; ++p) {
                                const int t = p + PW - 1 - kj;                  // compile-time after unrolling
                                if (SW == 1 || ((t & 1) == B0ODD))
                                    acc[cc][p] = fmaf(win[SW == 1 ? t : (t + B0ODD) / 2], wv, acc[cc][p]);
                            }
                        }
                    }
                }
            }
            m0 = nm0; kd = nkd; zd = nzd; have = nhave;
        }
        if (y < g.H) {
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                const int c = c0 + cc;
                if (c >= g.C) continue;
                const size_t rowi = ((((size_t)n * g.C + c) * g.D + d) * g.H + y) * g.W;
#pragma unroll
                for (int p = 0; p < PXT; ++p) {
                    const int xo = x0 + p;
                    if (xo >= g.W) continue;
                    if (partial) {          // split launch: raw channel-slice sum, folded by k_synth_fold
                        partial[(size_t)blockIdx.z * ((size_t)g.N * g.C * g.D * g.H * g.W) + rowi + xo] = acc[cc][p];
                        continue;
                    }
                    float v = alpha * acc[cc][p];
                    if (mask) v *= mask[rowi + xo];
                    if (sub) v -= sub[rowi + xo];
                    out[rowi + xo] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// filter gradient: one workgroup per (m, c, kd, ki) filter row as in the untiled kernel, but every
// thread takes 4 consecutive code columns per step (one row window of the image feeds 4*PW FMAs)
// and all-zero quads of the sparse / gated code are skipped.  Fixed reduction order.
template <int PW, int SW>
__global__ __launch_bounds__(256) void k_wgrad_t(cdl_geom g, const float *__restrict__ z,
                                                 const float *__restrict__ gate,
                                                 const float *__restrict__ x, float alpha,
                                                 float *__restrict__ dw)
{
    __shared__ float red[4][PW];
    constexpr int WL = (PXT - 1) * SW + PW;
    const int Dz = g.D / g.sd, Hz = g.H / g.sh, Wz = g.W / g.sw;
    const int m = blockIdx.x;
    int r = blockIdx.y;
    const int ki = r % g.Ph; r /= g.Ph;
    const int kd = r % g.Pd;
    const int c = r / g.Pd;
    float acc[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) acc[j] = 0.0f;

    const int W4 = (Wz + PXT - 1) / PXT;
    const int rows = g.N * Dz * Hz;
    const long items = (long)rows * W4;
    for (long it = threadIdx.x; it < items; it += 256) {
        const int row = (int)(it / W4), zx0 = (int)(it % W4) * PXT;
        const int zy = row % Hz, t = row / Hz;
        const int zd = t % Dz, n = t / Dz;
        const int d = zd * g.sd - g.pd + kd, y = zy * g.sh - g.ph + ki;
        if (d < 0 || d >= g.D || y < 0 || y >= g.H) continue;
        const size_t zoff = ((((size_t)n * g.M + m) * Dz + zd) * Hz + zy) * Wz + zx0;
        float zv[PXT];
        bool any = false;
#pragma unroll
        for (int p = 0; p < PXT; ++p) {
            float v = (zx0 + p < Wz) ? z[zoff + p] : 0.0f;
            if (gate && v != 0.0f && gate[zoff + p] == 0.0f) v = 0.0f;
            zv[p] = v;
            any |= v != 0.0f;
        }
        if (!any) continue;
        const float *xr = x + ((((size_t)n * g.C + c) * g.D + d) * g.H + y) * g.W;
        const int xb = zx0 * SW - g.pw;
        float win[WL];
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            const int xx = xb + i;
            win[i] = (xx >= 0 && xx < g.W) ? xr[xx] : 0.0f;
        }
#pragma unroll
        for (int kj = 0; kj < PW; ++kj)
#pragma unroll
            for (int p = 0; p < PXT; ++p) acc[kj] = fmaf(zv[p], win[p * SW + kj], acc[kj]);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int kj = 0; kj < PW; ++kj) {
        float v = acc[kj];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wv][kj] = v;
    }
    __syncthreads();
    if (threadIdx.x < PW) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        dw[((((size_t)m * g.C + c) * g.Pd + kd) * g.Ph + ki) * PW + threadIdx.x] = alpha * v;
    }
}

// ------------------------------------------------------------------------------------------
// filter gradient, second tier: one workgroup per (m, c, kd) and per chunk of code rows, all
// PH x PW taps of the filter plane in registers, so the code is read C*Pd times in total instead
// of C*Pd*Ph times.  Partial sums per row chunk go to `part`; k_wgrad_fold adds them in order.
constexpr int ZR = 2;             // code rows per thread in k_wgrad_p: their image-row windows overlap

template <int PH, int PW, int SW>
__global__ __launch_bounds__(256) void k_wgrad_p(cdl_geom g, const float *__restrict__ z,
                                                 const float *__restrict__ gate,
                                                 const float *__restrict__ x, float *__restrict__ part,
                                                 int groups_per_chunk, int lpr_shift)
{
    __shared__ float red[4][PH * PW];
    constexpr int WL = (PXT - 1) * SW + PW;
    constexpr int NXR = (ZR - 1) * SW + PH;               // image rows under ZR consecutive code rows
    const int Dz = g.D / g.sd, Hz = g.H / g.sh, Wz = g.W / g.sw;
    const int m = blockIdx.x;
    const int kd = blockIdx.y % g.Pd, c = blockIdx.y / g.Pd;
    float acc[PH][PW];
#pragma unroll
    for (int i = 0; i < PH; ++i)
#pragma unroll
        for (int j = 0; j < PW; ++j) acc[i][j] = 0.0f;

    // work item = (group of ZR code rows of one (n, zd) plane, quad of 4 code columns); a power-of-two
    // number of lanes walks the quads of a row group, the other lanes take further row groups:
    // 32-bit index arithmetic only, one division per row group
    const int W4 = (Wz + PXT - 1) / PXT, HG = (Hz + ZR - 1) / ZR;
    const int groups = g.N * Dz * HG;
    const int g0 = blockIdx.z * groups_per_chunk, g1 = min(groups, g0 + groups_per_chunk);
    const int lpr = 1 << lpr_shift, q0 = threadIdx.x & (lpr - 1), gstep = 256 >> lpr_shift;
    for (int grp = g0 + (threadIdx.x >> lpr_shift); grp < g1; grp += gstep) {
        const int zyg = grp % HG, t = grp / HG;
        const int zd = t % Dz, n = t / Dz;
        const int d = zd * g.sd - g.pd + kd;
        if (d < 0 || d >= g.D) continue;
        const int zy0 = zyg * ZR, yb = zy0 * g.sh - g.ph;
        const float *xplane = x + (((size_t)n * g.C + c) * g.D + d) * g.H * g.W;
        const size_t zrow0 = ((((size_t)n * g.M + m) * Dz + zd) * Hz + zy0) * Wz;
        for (int q = q0; q < W4; q += lpr) {
            const int zx0 = q * PXT;
            float zv[ZR][PXT];
            bool any = false;
#pragma unroll
            for (int r = 0; r < ZR; ++r)
#pragma unroll
                for (int p = 0; p < PXT; ++p) {
                    const size_t zi = zrow0 + (size_t)r * Wz + zx0 + p;
                    float v = (zy0 + r < Hz && zx0 + p < Wz) ? z[zi] : 0.0f;
                    if (gate && v != 0.0f && gate[zi] == 0.0f) v = 0.0f;
                    zv[r][p] = v;
                    any |= v != 0.0f;
                }
            if (!any) continue;
            const int xb = zx0 * SW - g.pw;
#pragma unroll
            for (int xr = 0; xr < NXR; ++xr) {
                const int y = yb + xr;
                if (y < 0 || y >= g.H) continue;
                const float *xrow = xplane + (size_t)y * g.W;
                float win[WL];
#pragma unroll
                for (int i = 0; i < WL; ++i) {
                    const int xx = xb + i;
                    win[i] = (xx >= 0 && xx < g.W) ? xrow[xx] : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < ZR; ++r) {
                    const int ki = xr - r * SW;                        // compile-time after unrolling
                    if (ki < 0 || ki >= PH) continue;
#pragma unroll
                    for (int kj = 0; kj < PW; ++kj)
#pragma unroll
                        for (int p = 0; p < PXT; ++p) acc[ki][kj] = fmaf(zv[r][p], win[p * SW + kj], acc[ki][kj]);
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < PH; ++i)
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            float v = acc[i][j];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) red[wv][i * PW + j] = v;
        }
    __syncthreads();
    if (threadIdx.x < PH * PW) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        // part[chunk][m][c][kd][PH*PW]
        part[(((size_t)blockIdx.z * g.M + m) * g.C * g.Pd + blockIdx.y) * (PH * PW) + threadIdx.x] = v;
    }
}

// ------------------------------------------------------------------------------------------
// filter gradient, image tile in LDS: a workgroup owns a 64 x 32 tile of code pixels of one (n, zd) and one
// (c, kd); the image rows under it (thin) are staged in LDS ONCE and every code channel m is then taken by
// one of the 4 waves in turn: z is read exactly C*Pd times in total (k_wgrad_p re-fetches the image windows
// from L1/L2 for every m: M x more image traffic and 3 FMAs per global load).  A lane accumulates its
// PH x PW plane over 4 items of 2 code rows x 4 columns, the wave reduces it by a butterfly and writes the
// per-(tile, c, kd, m) partial; k_wgrad_tfold adds the tiles in a fixed order.
constexpr int WLX = 64, WLY = 32;     // code-pixel tile of k_wgrad_l

template <int PH, int PW, int SW>
__global__ __launch_bounds__(256) void k_wgrad_l(cdl_geom g, const float *__restrict__ z,
                                                 const float *__restrict__ gate,
                                                 const float *__restrict__ x, float *__restrict__ part,
                                                 int tilesX, int tilesY, int XW)
{
    extern __shared__ float xt[];                          // [XH][XW] image rows under the tile (zero outside)
    constexpr int WL = (PXT - 1) * SW + PW;
    constexpr int NXR = (ZR - 1) * SW + PH;
    constexpr int XH = (WLY - 1) * SW + PH;
    const int Dz = g.D / g.sd, Hz = g.H / g.sh, Wz = g.W / g.sw;
    int b = blockIdx.x;
    const int tx = b % tilesX; b /= tilesX;
    const int ty = b % tilesY; b /= tilesY;
    const int zd = b % Dz, n = b / Dz;
    const int kd = blockIdx.y % g.Pd, c = blockIdx.y / g.Pd;
    const int d = zd * g.sd - g.pd + kd;
    const bool dok = d >= 0 && d < g.D;                    // uniform: a plane outside the image contributes zeros
    const int ybase = ty * WLY * SW - g.ph, xbase = tx * WLX * SW - g.pw;
    if (dok) {
        const float *xplane = x + (((size_t)n * g.C + c) * g.D + d) * g.H * g.W;
        for (int i = threadIdx.x; i < XH * XW; i += 256) {
            const int col = i % XW, row = i / XW;
            const int yy = ybase + row, xx = xbase + col;
            xt[i] = (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? xplane[(size_t)yy * g.W + xx] : 0.0f;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t pbase = ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * g.M;      // [tile][c,kd][m][PH*PW]
    constexpr int NIT = (WLX / PXT) * (WLY / ZR) / 64;    // items (2 code rows x 4 columns) per lane and channel
    // the code values of ALL items of a channel are loaded up front, and the next channel's while this one
    // is multiplied: with one item's loads at a time the kernel sat on global-load latency (15 % of VALU peak)
    auto load_items = [&](int m, float (&dst)[NIT][ZR][PXT]) {
        const size_t zplane = (((size_t)n * g.M + m) * Dz + zd) * Hz;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = lane + 64 * it;
            const int q = item & (WLX / PXT - 1), rp = item / (WLX / PXT);
            const int zy0 = ty * WLY + rp * ZR, zx0 = tx * WLX + q * PXT;
#pragma unroll
            for (int r = 0; r < ZR; ++r)
#pragma unroll
                for (int p = 0; p < PXT; ++p) {
                    const bool ok = dok && zy0 + r < Hz && zx0 + p < Wz;
                    const size_t zi = ok ? (zplane + zy0 + r) * Wz + zx0 + p : 0;
                    float v = z[zi];
                    if (gate && gate[zi] == 0.0f) v = 0.0f;
                    dst[it][r][p] = ok ? v : 0.0f;
                }
        }
    };
    float zv[NIT][ZR][PXT], zn[NIT][ZR][PXT];
    if (wv < g.M) load_items(wv, zv);
    for (int m = wv; m < g.M; m += 4) {
        if (m + 4 < g.M) load_items(m + 4, zn);
        float acc[PH][PW];
#pragma unroll
        for (int i = 0; i < PH; ++i)
#pragma unroll
            for (int j = 0; j < PW; ++j) acc[i][j] = 0.0f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = lane + 64 * it;
            const int q = item & (WLX / PXT - 1), rp = item / (WLX / PXT);
            bool any = false;
#pragma unroll
            for (int r = 0; r < ZR; ++r)
#pragma unroll
                for (int p = 0; p < PXT; ++p) any |= zv[it][r][p] != 0.0f;
            if (!any) continue;
            const float *xrow = xt + (rp * ZR * SW) * XW + q * PXT * SW;
#pragma unroll
            for (int xr = 0; xr < NXR; ++xr) {
                // 16-byte LDS reads (XW and the column offset are multiples of 4 words)
                constexpr int WL4 = (WL + 3) / 4;
                float win[WL4 * 4];
                const float4 *x4 = reinterpret_cast<const float4 *>(xrow + xr * XW);
#pragma unroll
                for (int i = 0; i < WL4; ++i) {
                    const float4 v4 = x4[i];
                    win[4 * i] = v4.x; win[4 * i + 1] = v4.y; win[4 * i + 2] = v4.z; win[4 * i + 3] = v4.w;
                }
#pragma unroll
                for (int r = 0; r < ZR; ++r) {
                    const int ki = xr - r * SW;                    // compile-time after unrolling
                    if (ki < 0 || ki >= PH) continue;
#pragma unroll
                    for (int kj = 0; kj < PW; ++kj)
#pragma unroll
                        for (int p = 0; p < PXT; ++p) acc[ki][kj] = fmaf(zv[it][r][p], win[p * SW + kj], acc[ki][kj]);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int r = 0; r < ZR; ++r)
#pragma unroll
                for (int p = 0; p < PXT; ++p) zv[it][r][p] = zn[it][r][p];
        // butterfly over the 64 lanes (fixed order), then lanes 0..PH*PW-1 store one tap each
        float *dst = part + (pbase + m) * (PH * PW);
#pragma unroll
        for (int i = 0; i < PH; ++i)
#pragma unroll
            for (int j = 0; j < PW; ++j) {
                float v = acc[i][j];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                if (lane == ((i * PW + j) & 63)) dst[i * PW + j] = v;
            }
    }
}

// dw[m][c][kd][tap] = alpha * sum_tiles part[tile][c,kd][m][tap].  16 outputs x 16 tile-strided partial sums
// per workgroup, combined through LDS in a fixed order (serial per-output loops over ~500 tiles cost 0.2 ms).
__global__ __launch_bounds__(256) void k_wgrad_tfold(const float *__restrict__ part, float *__restrict__ dw,
                                                     float alpha, int tiles, int CPd, int M, int taps)
{
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, ps = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o, total = M * CPd * taps;
    float s = 0.0f;
    if (i < total) {
        const int tap = i % taps;
        const int r = i / taps;
        const int cy = r % CPd, m = r / CPd;
        const size_t off = ((size_t)cy * M + m) * taps + tap, tstride = (size_t)CPd * M * taps;
        for (int t = ps; t < tiles; t += 16) s += part[(size_t)t * tstride + off];
    }
    red[ps][o] = s;
    __syncthreads();
    if (ps == 0 && i < total) {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][o];
        dw[i] = alpha * v;                                  // dw layout (M, C, Pd, Ph, Pw) = [m][cy][tap]
    }
}

__global__ void k_wgrad_fold(const float *__restrict__ part, float *__restrict__ dw, float alpha, int chunks,
                             int total)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float s = 0.0f;
    for (int k = 0; k < chunks; ++k) s += part[(size_t)k * total + i];
    dw[i] = alpha * s;
}

// out = mask * alpha * sum_chunks partial - sub, chunks added in a fixed order
__global__ void k_synth_fold(const float *__restrict__ part, const float *__restrict__ mask,
                             const float *__restrict__ sub, float alpha, float *__restrict__ out,
                             int chunks, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float s = 0.0f;
    for (int k = 0; k < chunks; ++k) s += part[(size_t)k * total + i];
    float v = alpha * s;
    if (mask) v *= mask[i];
    if (sub) v -= sub[i];
    out[i] = v;
}

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

// A launch with few workgroups (single frames, small crops) leaves most of the 256 CUs idle: split the
// code channels over blockIdx.z until there are about 1024 workgroups.  Returns channels per slice.
inline int channel_split(int M, long blocks, int *chunks)
{
    const int groups = (M + MCH - 1) / MCH;
    long want = blocks >= 512 ? 1 : (1024 + blocks - 1) / blocks;
    if (want > groups) want = groups;
    const int mper = (int)((groups + want - 1) / want) * MCH;
    *chunks = (M + mper - 1) / mper;
    return mper;
}

template <int PW, int SW>
int launch_analysis(const cdl_geom *g, const float *x, const float *w, float alpha, const float *zin,
                    const float *gate, const float *tau, float *out, const cdl_prox_args &px, void *stream)
{
    const int Dz = g->D / g->sd, Hz = g->H / g->sh, Wz = g->W / g->sw;
    const int tilesX = (Wz + TX - 1) / TX, tilesY = (Hz + TY - 1) / TY;
    const int PH = (TY - 1) * g->sh + g->Ph;
    const int PWp = (((TX - 1) * SW + PW) + 3) & ~3;
    const size_t lds = (size_t)g->C * g->Pd * PH * PWp * sizeof(float);
    if (lds > 96 * 1024) return CDL_EUNSUPPORTED;
    if (lds > 64 * 1024) {
        const int rc_ = cdl_ensure_dynamic_lds((const void *)k_analysis_t<PW, SW>, (int)lds);
        if (rc_) return rc_;
    }
    int chunks;
    const int mper = channel_split(g->M, (long)tilesX * tilesY * Dz * g->N, &chunks);
    dim3 grid((unsigned)(tilesX * tilesY * Dz), (unsigned)g->N, (unsigned)chunks);
    k_analysis_t<PW, SW><<<grid, 256, lds, S(stream)>>>(*g, x, w, alpha, zin, gate, tau, out, tilesX, tilesY, PH, PWp,
                                                        mper, px);
    CDL_LAUNCH_CHECK();
    return 0;
}

template <int PW, int SW, int CC, int MCHS>
int launch_synthesis_m(const cdl_geom *g, const float *z, const float *gate, const float *w, float alpha,
                     const float *mask, const float *sub, float *out, float *ws, size_t ws_floats, void *stream)
{
    const int tilesX = (g->W + TX - 1) / TX, tilesY = (g->H + TY - 1) / TY;
    const int PZH = (TY - 1 + g->Ph - 1) / g->sh + 2;
    const int PZW = (TX - 1 + PW - 1) / SW + 2 + 4;          // + slack for the fixed-length row windows
    const size_t lds = (size_t)MCHS * PZH * PZW * sizeof(float);
    if (lds > 64 * 1024) return CDL_EUNSUPPORTED;
    int chunks;
    int mper = channel_split(g->M, (long)tilesX * tilesY * g->D * g->N, &chunks);
    const size_t total = (size_t)g->N * g->C * g->D * g->H * g->W;
    if (chunks > 1 && (!ws || ws_floats < (size_t)chunks * total)) {      // no room for the partial sums
        chunks = 1;
        mper = g->M;
    }
    dim3 grid((unsigned)(tilesX * tilesY * g->D), (unsigned)g->N, (unsigned)chunks);
    k_synthesis_t<PW, SW, CC, MCHS><<<grid, 256, lds, S(stream)>>>(*g, z, gate, w, alpha, mask, sub, out, tilesX,
                                                          tilesY, PZH, PZW, mper, chunks > 1 ? ws : nullptr);
    CDL_LAUNCH_CHECK();
    if (chunks > 1) {
        k_synth_fold<<<(unsigned)((total + 255) / 256), 256, 0, S(stream)>>>(ws, mask, sub, alpha, out, chunks, total);
        CDL_LAUNCH_CHECK();
    }
    return 0;
}

template <int PW, int SW, int CC>
int launch_synthesis(const cdl_geom *g, const float *z, const float *gate, const float *w, float alpha,
                     const float *mask, const float *sub, float *out, float *ws, size_t ws_floats, void *stream)
{    {   // pipelined kernel when a thread can hold its share of a two-channel patch in registers
        const int PZH = (TY - 1 + g->Ph - 1) / g->sh + 2;
        const int PZW = (TX - 1 + PW - 1) / SW + 2 + 4;
        const size_t slab = (size_t)(g->D / g->sd) * (g->H / g->sh) * (g->W / g->sw);
        // measured (rocprofv3, per launch): cfg3 3-D 1.34 -> 1.11 ms, cfg4 C=3 0.50 -> 0.46 ms, but the 2-D C=1
        // M=169 stride-2 shape 1.02 -> 1.89 ms (4x the rounds and barriers of the 8-channel kernel for little
        // work per round): pipelined only where a round carries Pd or C times more work
        if ((g->Pd > 1 || g->C > 1) && MQ * PZH * PZW <= QNE * 256 && MQ * slab < ((size_t)1 << 31) &&
            !cdl_opts().no_pipelined_synthesis) {
            const int tilesX = (g->W + TX - 1) / TX, tilesY = (g->H + TY - 1) / TY;
            const size_t lds = (size_t)MQ * PZH * PZW * sizeof(float);
            int chunks;
            int mper = channel_split(g->M, (long)tilesX * tilesY * g->D * g->N, &chunks);
            const size_t total = (size_t)g->N * g->C * g->D * g->H * g->W;
            if (chunks > 1 && (!ws || ws_floats < (size_t)chunks * total)) {
                chunks = 1;
                mper = g->M;
            }
            dim3 grid((unsigned)(tilesX * tilesY * g->D), (unsigned)g->N, (unsigned)chunks);
            k_synthesis_q<PW, SW, CC><<<grid, 256, lds, S(stream)>>>(*g, z, gate, w, alpha, mask, sub, out, tilesX,
                                                                  tilesY, PZH, PZW, mper, chunks > 1 ? ws : nullptr);
            CDL_LAUNCH_CHECK();
            if (chunks > 1) {
                k_synth_fold<<<(unsigned)((total + 255) / 256), 256, 0, S(stream)>>>(ws, mask, sub, alpha, out, chunks, total);
                CDL_LAUNCH_CHECK();
            }
            return 0;
        }
    }

    // measured on cfg3 (3-D, 5x5x5): 2 channels per LDS round 30.7 ms forward, 4: 33.5, 8: 37.5; no effect in 2-D
    if (g->Pd > 1) return launch_synthesis_m<PW, SW, CC, 2>(g, z, gate, w, alpha, mask, sub, out, ws, ws_floats, stream);
    return launch_synthesis_m<PW, SW, CC, 8>(g, z, gate, w, alpha, mask, sub, out, ws, ws_floats, stream);
}

}  // namespace

// Returns CDL_EUNSUPPORTED when the shape has no tiled instantiation (the caller falls back).
int cdl_tiled_analysis(const cdl_geom *g, const float *x, const float *w, float alpha, const float *zin,
                       const float *gate, const float *tau, float *out, const cdl_prox_args &px, void *stream)
{
    if (g->sw != g->sh) return CDL_EUNSUPPORTED;
#define CDL_A(PW_, SW_) if (g->Pw == PW_ && g->sw == SW_) return launch_analysis<PW_, SW_>(g, x, w, alpha, zin, gate, tau, out, px, stream)
    CDL_A(3, 1); CDL_A(5, 1); CDL_A(7, 1); CDL_A(9, 1);
    CDL_A(3, 2); CDL_A(5, 2); CDL_A(7, 2); CDL_A(9, 2);
#undef CDL_A
    return CDL_EUNSUPPORTED;
}

size_t cdl_tiled_synthesis_ws_floats(const cdl_geom *g)
{
    const int tilesX = (g->W + TX - 1) / TX, tilesY = (g->H + TY - 1) / TY;
    int chunks;
    channel_split(g->M, (long)tilesX * tilesY * g->D * g->N, &chunks);
    return chunks > 1 ? (size_t)chunks * g->N * g->C * g->D * g->H * g->W : 0;
}

int cdl_tiled_synthesis(const cdl_geom *g, const float *z, const float *gate, const float *w, float alpha,
                        const float *mask, const float *sub, float *out, float *ws, size_t ws_floats,
                        void *stream)
{
    if (g->sw != g->sh || g->pw != g->Pw / 2) return CDL_EUNSUPPORTED;
    if (g->C != 1 && g->C != 3) return CDL_EUNSUPPORTED;
#define CDL_S(PW_, SW_)                                                                              \
    if (g->Pw == PW_ && g->sw == SW_)                                                                \
        return g->C == 1 ? launch_synthesis<PW_, SW_, 1>(g, z, gate, w, alpha, mask, sub, out, ws, ws_floats, stream) \
                         : launch_synthesis<PW_, SW_, 3>(g, z, gate, w, alpha, mask, sub, out, ws, ws_floats, stream)
    CDL_S(3, 1); CDL_S(5, 1); CDL_S(7, 1); CDL_S(9, 1);
    CDL_S(3, 2); CDL_S(5, 2); CDL_S(7, 2); CDL_S(9, 2);
#undef CDL_S
    return CDL_EUNSUPPORTED;
}

int cdl_tiled_wgrad(const cdl_geom *g, const float *z, const float *gate, const float *x, float alpha,
                    float *dw, float *workspace, size_t workspace_floats, void *stream)
{
    if (g->sw != g->sh) return CDL_EUNSUPPORTED;
    if (workspace) {
        // first choice: image tile in LDS, every code channel inside the workgroup (z read C*Pd times in total)
        const int Dz_ = g->D / g->sd, Hz_ = g->H / g->sh, Wz_ = g->W / g->sw;
        const int tX = (Wz_ + WLX - 1) / WLX, tY = (Hz_ + WLY - 1) / WLY;
        const long tiles = (long)g->N * Dz_ * tX * tY;
        const int CPd = g->C * g->Pd, tapsl = g->Ph * g->Pw;
        const int XW = (((WLX - 1) * g->sw + g->Pw + 3) + 3) & ~3;     // multiple of 4 words + room for the 16-byte reads
        const size_t ldsl = (size_t)((WLY - 1) * g->sh + g->Ph) * XW * sizeof(float);
        if (tiles * CPd >= 128 && (size_t)tiles * CPd * g->M * tapsl <= workspace_floats && ldsl <= 64 * 1024 &&
            tiles < (1L << 31)) {
            dim3 gridl((unsigned)tiles, (unsigned)CPd);
            const int totall = g->M * CPd * tapsl;
#define CDL_L(PH_, PW_, SW_)                                                                              \
            if (g->Ph == PH_ && g->Pw == PW_ && g->sw == SW_) {                                            \
                k_wgrad_l<PH_, PW_, SW_><<<gridl, 256, ldsl, S(stream)>>>(*g, z, gate, x, workspace, tX, tY, XW); \
                CDL_LAUNCH_CHECK();                                                                        \
                k_wgrad_tfold<<<(totall + 15) / 16, 256, 0, S(stream)>>>(workspace, dw, alpha, (int)tiles, CPd, g->M, tapsl); \
                CDL_LAUNCH_CHECK();                                                                        \
                return 0;                                                                                  \
            }
            CDL_L(3, 3, 1) CDL_L(5, 5, 1) CDL_L(7, 7, 1) CDL_L(9, 9, 1) CDL_L(3, 5, 1) CDL_L(7, 5, 1) CDL_L(9, 5, 1)
            CDL_L(3, 3, 2) CDL_L(5, 5, 2) CDL_L(7, 7, 2) CDL_L(9, 9, 2) CDL_L(3, 5, 2) CDL_L(7, 5, 2) CDL_L(9, 5, 2)
#undef CDL_L
        }
        // second tier: whole filter planes in registers, code rows split into chunks
        const int total = g->M * g->C * g->Pd * g->Ph * g->Pw;
        const int Hz = g->H / g->sh, Wz = g->W / g->sw;
        const int rows = g->N * (g->D / g->sd) * ((Hz + ZR - 1) / ZR);      // row groups of ZR code rows
        int chunks = 2048 / (g->M * g->C * g->Pd);
        if (chunks < 1) chunks = 1;
        if (chunks > rows) chunks = rows;
        while (chunks > 1 && (size_t)chunks * total > workspace_floats) --chunks;
        const int rpc = (rows + chunks - 1) / chunks;
        chunks = (rows + rpc - 1) / rpc;
        int lpr_shift = 0;                                                   // lanes per row group: pow2 >= quads
        while ((1 << lpr_shift) < (Wz + PXT - 1) / PXT && lpr_shift < 8) ++lpr_shift;
        if ((size_t)chunks * total <= workspace_floats) {
            dim3 grid2((unsigned)g->M, (unsigned)(g->C * g->Pd), (unsigned)chunks);
#define CDL_P(PH_, PW_, SW_)                                                                              \
            if (g->Ph == PH_ && g->Pw == PW_ && g->sw == SW_) {                                            \
                k_wgrad_p<PH_, PW_, SW_><<<grid2, 256, 0, S(stream)>>>(*g, z, gate, x, workspace, rpc, lpr_shift); \
                CDL_LAUNCH_CHECK();                                                                        \
                k_wgrad_fold<<<(total + 255) / 256, 256, 0, S(stream)>>>(workspace, dw, alpha, chunks, total); \
                CDL_LAUNCH_CHECK();                                                                        \
                return 0;                                                                                  \
            }
            CDL_P(3, 3, 1) CDL_P(5, 5, 1) CDL_P(7, 7, 1) CDL_P(9, 9, 1) CDL_P(3, 5, 1) CDL_P(9, 5, 1)
            CDL_P(3, 3, 2) CDL_P(5, 5, 2) CDL_P(7, 7, 2) CDL_P(9, 9, 2) CDL_P(3, 5, 2) CDL_P(9, 5, 2)
#undef CDL_P
        }
    }
    dim3 grid((unsigned)g->M, (unsigned)(g->C * g->Pd * g->Ph));
#define CDL_W(PW_, SW_)                                                                     \
    if (g->Pw == PW_ && g->sw == SW_) {                                                     \
        k_wgrad_t<PW_, SW_><<<grid, 256, 0, S(stream)>>>(*g, z, gate, x, alpha, dw);        \
        CDL_LAUNCH_CHECK();                                                                 \
        return 0;                                                                           \
    }
    CDL_W(3, 1) CDL_W(5, 1) CDL_W(7, 1) CDL_W(9, 1)
    CDL_W(3, 2) CDL_W(5, 2) CDL_W(7, 2) CDL_W(9, 2)
#undef CDL_W
    return CDL_EUNSUPPORTED;
}
