// Filter gradients on the matrix cores for the shapes the fused 2-D kernel does not take (any C, 2-D / 3-D,
// stride 1 / 2, odd square planes up to 9 x 9): what autograd computes for the reference's conv / conv-transpose
// weights (train.py:98, train3d.py:113),
//     dw[m][c][kd][ki][kj] = alpha * sum_{n,zd,zy,zx} F[n][m][zd][zy][zx] * X[n][c][zd*sd-pd+kd][zy*s-ph+ki][zx*s-pw+kj]
// with F the code-like operand (optionally gated by the support of another code tensor).
//
// Per (c, kd) group one GEMM with the PIXEL index as k:   D[tap][m] = sum_px A[tap][px] * B[px][m]
//   A: im2col of the thin image, gathered from a bf16 hi/lo tile in LDS (lane = tap row, 8 consecutive pixels)
//   B: the fat operand straight from global memory (lane = channel column, 8 consecutive pixels of its row)
// split-bf16 x3 products, fp32 accumulation.  A workgroup owns a 64 x 32 tile of code pixels of one (n, zd); its 8
// waves split the code channels (two 32-channel tiles per wave) and then the pixels; every wave writes its own
// partial bank to the workspace and k_wgm_fold adds all partials in a fixed order: deterministic, no atomics.
#include "cdl_common.h"

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int GLX = 64, GLY = 32;          // code-pixel tile of a workgroup
constexpr int GNT = 512;                   // 8 waves
constexpr int KSTEPS = GLX * GLY / 16;     // 16-pixel k-steps per tile (4 per code row)
constexpr int LDS_MAX = 128 * 1024;

// Unit stride: the thin tile is kept as bf16 PAIRS, twice -- plane E holds (x[2c], x[2c+1]), plane O holds
// (x[2c+1], x[2c+2]) -- so that the 8 consecutive pixels of an im2col row are 4 whole dwords whatever the parity of the
// tap column (a 16-byte read at 2-byte alignment costs ~8x an aligned one: the first version spent 60-70 % of the
// launch in the LDS pipe).  The reads are ds_read2_b32: 32 banks, the 32 lanes (= taps) of a half-wave form a group.
// Tap (ki, kj) reads dword  plane(kj & 1) + ki * RD + (kj >> 1)  (+ a wave-uniform offset), so with the row pitch
// RD = P (mod 32) and the plane pitch PD = ceil(P / 2) (mod 32) the taps of a filter row fill P consecutive banks and
// the rows follow each other: no two of the 32 taps of an instruction share a bank (P <= 5; two banks 2-way for P = 7).
// (With the natural pitch 34 and the planes 16 apart, half of all LDS cycles of the k-loop were conflict cycles.)
constexpr int pad_to(int d, int r) { return d + ((r - d % 32) + 32) % 32; }       // smallest d' >= d with d' = r (mod 32)
constexpr int eo_row_dwords(int sw, int pw) { return pad_to(((GLX - 1) * sw + pw + 1) / 2, pw % 32); }
constexpr int eo_plane_dwords(int sw, int ph, int pw)
{
    return pad_to(((GLY - 1) * sw + ph) * eo_row_dwords(sw, pw), (pw + 1) / 2);
}
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ unsigned int pack_bf16(float a, float b)
{
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned int, v);
}

// NG = number of (c, kd) groups held at once: the fat operand is loaded and split into bf16 parts ONCE per
// k-step and multiplied against the thin tiles of all NG groups (their LDS tiles and accumulators live side by
// side); launches loop over ceil(G / NG) group batches.
// CT = 32-channel tiles per wave: 2 halves the LDS gathers per fat byte, 1 halves the accumulators (NG*RT*16 of them),
// which is what lets two workgroups share a CU (<= 128 registers per lane) when NG*RT <= 5.
template <int PH, int PW, int SW, int NG, int CT>
__global__ __launch_bounds__(GNT) void k_wgm(cdl_geom g, const float *__restrict__ F0,
                                             const float *__restrict__ gate, const float *__restrict__ x0,
                                             float *__restrict__ part0, int tilesX, int tilesY, int nct, int MP CDL_DBG_COMMA(int dbg),
                                             int ntiles, int tpw, const float *__restrict__ F1,
                                             const float *__restrict__ x1, int rsc, int gspan, int rs)
{
    // blockIdx.z: the (c, kd) groups [z gspan, (z + 1) gspan) -- the group passes of a tile are independent (each re-reads the
    // fat operand and owns its partial rows), so a launch with few tiles spreads them over workgroups.
    // blockIdx.y = 1: the second (fat, thin) operand pair of a paired launch (dA_k and dB_k of one iteration), its
    // partial banks behind the first pair's
    const float *__restrict__ F = blockIdx.y ? F1 : F0;
    const float *__restrict__ x = blockIdx.y ? x1 : x0;
    float *__restrict__ part = part0 + (size_t)blockIdx.y * gridDim.x * ((size_t)g.C * g.Pd * (((PH * PW + 31) / 32) * 32) * MP);
    constexpr int T = PH * PW, RT = (T + 31) / 32, TP = RT * 32;
    constexpr int XH = (GLY - 1) * SW + PH, XW = (GLX - 1) * SW + PW;
    constexpr int XE = ((XH * XW + 7) / 8) * 8;            // elements per bf16 plane (16-byte multiple)
    constexpr bool EO = SW == 1;                           // paired planes (see eo_plane_dwords)
    constexpr int RD = eo_row_dwords(SW, PW), PD = eo_plane_dwords(SW, PH, PW);
    extern __shared__ __align__(16) unsigned char smem[];
    __bf16 *xh = reinterpret_cast<__bf16 *>(smem);         // [NG][XE] hi parts of the image rows under the tile
    __bf16 *xl = xh + NG * XE;                             // [NG][XE] lo parts
    const unsigned int *xw = reinterpret_cast<const unsigned int *>(smem);   // [NG][Eh | Oh | El | Ol][PD]
    const int Dz = g.D / g.sd, Hz = g.H / g.sh, Wz = g.W / g.sw;
    // a workgroup accumulates tpw consecutive tiles (same image / depth plane mostly) before it reduces: one
    // cross-wave sum and one partial bank per workgroup
    int tx = 0, ty = 0, zd = 0, n = 0, ybase = 0, xbase = 0;
    size_t fbase = 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l32 = lane & 31, h = lane >> 5;
    const int G = g.C * g.Pd;
    const int npx = 8 / nct;                               // pixel parts (waves per channel group)
    const int cg = wv % nct, pp = wv / nct;
    const bool active = pp < npx;                          // 8 % nct waves idle when nct does not divide 8
    const int kpw = KSTEPS / (npx * rs);                   // k-steps per wave (rs: row parts of a tile, see the tile loop)
    const size_t slab = (size_t)Dz * Hz * Wz;
    // rsc != 0: the fat operands are in the strip kernel's row-strip channel-major layout
    // [n][code depth][code row][ceil(Wz/32)][M][32 columns] (include/cdlnet_hip.h, CDL_LAY_RSC): a lane's 8 consecutive pixels are
    // still 32 contiguous bytes (they never straddle a 32-column strip), the 32 channels of a half-wave are 128 bytes apart
    const int nsx = (Wz + 31) >> 5;
    auto fat_index = [&](int m, int cy, int cx) -> size_t {
        return rsc ? ((((size_t)n * Dz + zd) * Hz + cy) * nsx + (cx >> 5)) * ((size_t)g.M * 32) + (size_t)m * 32 + (cx & 31)
                   : fbase + (size_t)m * slab + (size_t)cy * Wz + cx;
    };
    // 16-byte loads of the fat operand need rows that start on 16-byte boundaries (the base pointers do)
    const bool vec4 = (Wz & 3) == 0 && ((reinterpret_cast<size_t>(F) | reinterpret_cast<size_t>(gate)) & 15) == 0;

    // tap of this lane's A rows (clamped into the plane for the padding rows: their outputs are never read)
    int tki[RT], tkj[RT], teo[RT];
#pragma unroll
    for (int R = 0; R < RT; ++R) {
        const int t = min(32 * R + l32, T - 1);
        tki[R] = t / PW;
        tkj[R] = t % PW;
        teo[R] = (tkj[R] & 1) * PD + tki[R] * RD + (tkj[R] >> 1);
    }
    // im2col row of tap tile R of group gi for the 8 pixels zx0 .. zx0+7 of code row zy
    auto gather = [&](int gi, int R, int zy, int zx0, bf16x8 &ah, bf16x8 &al) {
        if constexpr (EO) {
            const int o = gi * 4 * PD + teo[R] + zy * RD + (zx0 >> 1);
            u32x4 a, b;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = xw[o + i];
                b[i] = xw[o + 2 * PD + i];
            }
            ah = __builtin_bit_cast(bf16x8, a);
            al = __builtin_bit_cast(bf16x8, b);
        } else {
            const int o = gi * XE + (zy * SW + tki[R]) * XW + zx0 * SW + tkj[R];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                ah[i] = xh[o + i * SW];
                al[i] = xl[o + i * SW];
            }
        }
    };

    // thin-tile staging (unit stride): pair i of a group = dwords (x[2cp], x[2cp+1]) / (x[2cp+1], x[2cp+2]) of tile row
    // `row`; the loads of two groups are issued before any of them is converted (15 dependent rounds of load -> LDS
    // write took 12 of the first version's 100 us at cfg3)
    constexpr int NPAIR = XH * RD, ITG = (NPAIR + GNT - 1) / GNT;
    auto st_load = [&](int grp, float (&v)[ITG][3]) {
        const int kd = grp % g.Pd, c = grp / g.Pd;
        const int d = zd * g.sd - g.pd + kd;
        const bool dok = grp < G && d >= 0 && d < g.D;      // uniform; a plane outside the image is a zero tile
        const float *xplane = x + (((size_t)n * g.C + (dok ? c : 0)) * g.D + (dok ? d : 0)) * g.H * g.W;
#pragma unroll
        for (int it = 0; it < ITG; ++it) {
            const int i = threadIdx.x + it * GNT;
            const int cp = i % RD, row = i / RD;
            const int yy = ybase + row, xx = xbase + 2 * cp;
            const bool rok = dok && i < NPAIR && yy >= 0 && yy < g.H;
            const float *prow = xplane + (size_t)min(max(yy, 0), g.H - 1) * g.W;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const bool ok = rok && xx + e >= 0 && xx + e < g.W && 2 * cp + e < XW;
                const float t = prow[min(max(xx + e, 0), g.W - 1)];
                v[it][e] = ok ? t : 0.0f;
            }
        }
    };
    auto st_write = [&](int gi, const float (&v)[ITG][3]) {
        unsigned int *dstw = reinterpret_cast<unsigned int *>(smem) + gi * 4 * PD;
#pragma unroll
        for (int it = 0; it < ITG; ++it) {
            const int i = threadIdx.x + it * GNT;
            if (i < NPAIR) {
                float hi[3], lo[3];
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    hi[e] = (float)(__bf16)v[it][e];
                    lo[e] = v[it][e] - hi[e];
                }
                dstw[i] = pack_bf16(hi[0], hi[1]);
                dstw[PD + i] = pack_bf16(hi[1], hi[2]);
                dstw[2 * PD + i] = pack_bf16(lo[0], lo[1]);
                dstw[3 * PD + i] = pack_bf16(lo[1], lo[2]);
            }
        }
    };

    const int g_end = min(G, (int)(blockIdx.z + 1) * gspan);
    for (int g0 = blockIdx.z * gspan; g0 < g_end; g0 += NG) {
        f32x16 acc[NG][RT][CT];
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
            for (int R = 0; R < RT; ++R)
#pragma unroll
                for (int q = 0; q < CT; ++q)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[gi][R][q][v] = 0.0f;
#pragma unroll 1
        for (int tl = 0; tl < tpw; ++tl) {
        int b = blockIdx.x * tpw + tl;
        if (b >= ntiles) break;                            // uniform
        // rs > 1: a tile's k-steps (its 32 code rows) are split over rs consecutive work items -- launches with a handful of
        // tiles (one 128 x 128 crop of a stride-2 net: 2) reach more CUs; every item has its own partial bank as before
        const int kpart = b % rs;
        b /= rs;
        tx = b % tilesX; b /= tilesX;
        ty = b % tilesY; b /= tilesY;
        zd = b % Dz; n = b / Dz;
        fbase = (size_t)n * g.M * slab + (size_t)zd * Hz * Wz;
        ybase = ty * GLY * SW - g.ph; xbase = tx * GLX * SW - g.pw;
        // Fast path (rows are whole 32-byte segments: Wz % 8 == 0, 16-byte aligned bases): the fat operand of k-step
        // ks+1 is loaded -- branch-free, at a clamped address -- while k-step ks is converted, gathered and multiplied.
        // The first version issued the loads of a k-step and consumed them at once: every k-step exposed a global
        // load latency (62 % of the wave cycles were waits, 20 % matrix-core utilisation).  One k-step ahead keeps
        // 16 KB per CU in flight, which at ~1 us of loaded latency is the 4 TB/s the k-loop ran at; with NS = 4 slots
        // (small accumulator sets) three k-steps are in flight, and the first ones of a tile are issued BEFORE its
        // thin staging so they arrive under it.
        // (ungated operands only -- the sweeps gate their gradients in place upstream; a gate would double the
        //  prefetch registers and spill)
        constexpr bool PREFETCH = NG * RT * CT * 16 <= 160;   // 32 prefetch registers next to the accumulators
        const bool fast = PREFETCH && vec4 && (Wz & 7) == 0 && gate == nullptr;
        typedef __attribute__((ext_vector_type(4))) float f32x4;
        constexpr int NS = NG * RT * CT * 16 <= 96 ? 4 : 2;  // ring slots
        f32x4 fraw[NS][CT][2];                              // [slot][channel tile][half segment]

        auto fat_issue = [&](int ks, f32x4 (&fr)[CT][2]) {
            const int zy = ks >> 2, zx0 = (ks & 3) * 16 + 8 * h;
            const int cy = ty * GLY + zy, cx0 = tx * GLX + zx0;
#pragma unroll
            for (int q = 0; q < CT; ++q) {
                const int m = 32 * (CT * cg + q) + l32;
                const bool ok = m < g.M && cy < Hz && cx0 < Wz;
                const size_t idx = ok ? fat_index(m, cy, cx0) : fat_index(0, 0, 0);
                fr[q][0] = *reinterpret_cast<const f32x4 *>(F + idx);
                fr[q][1] = *reinterpret_cast<const f32x4 *>(F + idx + 4);
            }
        };
        auto kstep = [&](int ks, const f32x4 (&fr)[CT][2]) {
            const int zy = ks >> 2, zx0 = (ks & 3) * 16 + 8 * h;
            const int cy = ty * GLY + zy, cx0 = tx * GLX + zx0;
            bf16x8 bh[CT], bl[CT];
#pragma unroll
            for (int q = 0; q < CT; ++q) {
                const int m = 32 * (CT * cg + q) + l32;
                const bool ok = m < g.M && cy < Hz && cx0 < Wz;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float v = ok ? fr[q][i >> 2][i & 3] : 0.0f;
                    const __bf16 hh = (__bf16)v;
                    bh[q][i] = hh;
                    bl[q][i] = (__bf16)(v - (float)hh);
                }
            }
            // the im2col rows of tile t+1 are read from LDS while the products of tile t issue (a wait for every
            // group's reads right before its MFMAs left the matrix pipe idle half of the k-loop)
            bf16x8 ah[2], al[2];
            gather(0, 0, zy, zx0, ah[0], al[0]);
#pragma unroll
            for (int t = 0; t < NG * RT; ++t) {
                if (t + 1 < NG * RT) gather((t + 1) / RT, (t + 1) % RT, zy, zx0, ah[(t + 1) & 1], al[(t + 1) & 1]);
#pragma unroll
                for (int q = 0; q < CT; ++q) {
                    f32x16 &a = acc[t / RT][t % RT][q];
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t & 1], bh[q], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t & 1], bl[q], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t & 1], bh[q], a, 0, 0, 0);
                }
            }
        };
        const int k0 = (kpart * npx + pp) * kpw, k1 = k0 + kpw;   // kpw is a power of two >= 8
        const bool pipelined = PREFETCH && active && fast && !CDL_DBG(dbg, 2048);
        if constexpr (PREFETCH) {
            if (pipelined) {
#pragma unroll
                for (int sl = 0; sl + 1 < NS; ++sl) fat_issue(k0 + sl, fraw[sl]);
            }
        }
        __syncthreads();                                   // previous tile / previous batch's reduction has read the buffer
        if constexpr (EO) {
            if (!CDL_DBG(dbg, 1024)) {                           // (CDL_FUSED_DEBUG bits 1024 / 2048 / 4096: timing ablations)
                constexpr int GS = NG * RT * CT * 16 <= 160 ? 2 : 1;   // groups in flight (registers next to acc)
#pragma unroll
                for (int gi = 0; gi < NG; gi += GS) {
                    float va[ITG][3], vb[ITG][3];
                    st_load(g0 + gi, va);
                    if (GS == 2 && gi + 1 < NG) st_load(g0 + gi + 1, vb);
                    st_write(gi, va);
                    if (GS == 2 && gi + 1 < NG) st_write(gi + 1, vb);
                }
            }
        } else {
#pragma unroll 1
            for (int gi = 0; gi < (CDL_DBG(dbg, 1024) ? 0 : NG); ++gi) {
                const int grp = g0 + gi;
                const int kd = grp % g.Pd, c = grp / g.Pd;
                const int d = zd * g.sd - g.pd + kd;
                const bool dok = grp < G && d >= 0 && d < g.D;  // uniform; a plane outside the image is a zero tile
                const float *xplane = x + (((size_t)n * g.C + (dok ? c : 0)) * g.D + (dok ? d : 0)) * g.H * g.W;
                // SBW loads (clamped addresses, no branches) in flight per thread before any is converted: one element at a
                // time was a dependent global-load latency each, 18 per plane under the 9 x 5 filter
                constexpr int SBW = 6;
                for (int i0 = threadIdx.x; i0 < XH * XW; i0 += GNT * SBW) {
                    float v[SBW];
#pragma unroll
                    for (int u = 0; u < SBW; ++u) {
                        const int i = i0 + u * GNT;
                        const int col = i % XW, row = i / XW;
                        const int yy = ybase + row, xx = xbase + col;
                        const float t = xplane[(size_t)min(max(yy, 0), g.H - 1) * g.W + min(max(xx, 0), g.W - 1)];
                        v[u] = (dok && i < XH * XW && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? t : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < SBW; ++u) {
                        const int i = i0 + u * GNT;
                        if (i < XH * XW) {
                            const __bf16 hh = (__bf16)v[u];
                            xh[gi * XE + i] = hh;
                            xl[gi * XE + i] = (__bf16)(v[u] - (float)hh);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if constexpr (PREFETCH) {
            if (pipelined) {
#pragma unroll 1
                for (int ks = k0; ks < k1; ks += NS) {
#pragma unroll
                    for (int sl = 0; sl < NS; ++sl) {       // (past the end: the last k-step again, never consumed)
                        fat_issue(min(ks + sl + NS - 1, k1 - 1), fraw[(sl + NS - 1) % NS]);
                        kstep(ks + sl, fraw[sl]);
                    }
                }
            }
        }
        if (active && !pipelined && !CDL_DBG(dbg, 2048)) {
#pragma unroll 1
            for (int ks = k0; ks < k1; ++ks) {
                const int zy = ks >> 2, zx0 = (ks & 3) * 16 + 8 * h;       // tile-local pixels zx0 .. zx0+7 of row zy
                const int cy = ty * GLY + zy, cx0 = tx * GLX + zx0;
                // ---- B: the fat operand, 8 consecutive pixels of this lane's channel(s); shared by all NG groups
                bf16x8 bh[CT], bl[CT];
#pragma unroll
                for (int q = 0; q < CT; ++q) {
                    const int m = 32 * (CT * cg + q) + l32;
                    const bool mok = m < g.M && cy < Hz;
                    // (index of pixel cx0 of this lane's row; the 8 pixels of a k-step follow contiguously in both layouts)
                    const size_t rowi = fat_index(mok ? m : 0, mok ? cy : 0, cx0 < Wz ? cx0 : 0) - (cx0 < Wz ? cx0 : 0);
                    float fv[8];
                    if (vec4 && mok && cx0 + 8 <= Wz) {            // whole 32-byte segment inside the row: two 16-byte loads
                        const float4 a0 = *reinterpret_cast<const float4 *>(F + rowi + cx0);
                        const float4 a1 = *reinterpret_cast<const float4 *>(F + rowi + cx0 + 4);
                        fv[0] = a0.x; fv[1] = a0.y; fv[2] = a0.z; fv[3] = a0.w;
                        fv[4] = a1.x; fv[5] = a1.y; fv[6] = a1.z; fv[7] = a1.w;
                        if (gate) {
                            const float4 g0 = *reinterpret_cast<const float4 *>(gate + rowi + cx0);
                            const float4 g1 = *reinterpret_cast<const float4 *>(gate + rowi + cx0 + 4);
                            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                            for (int i = 0; i < 8; ++i) fv[i] = gg[i] == 0.0f ? 0.0f : fv[i];
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const bool ok = mok && cx0 + i < Wz;
                            const size_t idx = rowi + (ok ? cx0 + i : 0);
                            float v = F[idx];
                            if (gate && gate[idx] == 0.0f) v = 0.0f;
                            fv[i] = ok ? v : 0.0f;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const __bf16 hh = (__bf16)fv[i];
                        bh[q][i] = hh;
                        bl[q][i] = (__bf16)(fv[i] - (float)hh);
                    }
                }
                // ---- A: im2col rows of this lane's taps, gathered from LDS, and the products
#pragma unroll
                for (int gi = 0; gi < NG; ++gi)
#pragma unroll
                    for (int R = 0; R < RT; ++R) {
                        bf16x8 ah, al;
                        gather(gi, R, zy, zx0, ah, al);
#pragma unroll
                        for (int q = 0; q < CT; ++q) {
                            acc[gi][R][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[q], acc[gi][R][q], 0, 0, 0);
                            acc[gi][R][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[q], acc[gi][R][q], 0, 0, 0);
                            acc[gi][R][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[q], acc[gi][R][q], 0, 0, 0);
                        }
                    }
            }
        }
        }   // tiles of this workgroup
        // ---- sum the pixel parts of every channel group: a binary tree over the waves through LDS (the upper half
        //      of the parts writes, the lower half adds; 16-byte accesses, as many accumulator tiles per round as the
        //      buffer holds for 4 writers), then the waves of part 0 write ONE partial bank per tile,
        //      partial[tile][grp][tap][m]; register v of tile (R, q) is tap 32R + 8(v>>2) + 4h + (v&3) of channel
        //      32(CT cg + q) + l32.  Fixed order: deterministic.  (The first version made 2 barriers and a strided
        //      8-term sum per accumulator tile: 25 of the 100 us of a cfg3 launch.)
        if (!CDL_DBG(dbg, 4096)) {
            constexpr int NT = NG * RT * CT;
            constexpr int LDS_BYTES = NG * (EO ? 4 * PD * 4 : XE * 4) > 32768 ? NG * (EO ? 4 * PD * 4 : XE * 4) : 32768;
            constexpr int TPR = LDS_BYTES / (4 * 4096) < NT ? LDS_BYTES / (4 * 4096) : NT;   // tiles per round
            typedef __attribute__((ext_vector_type(4))) float f32x4;
            f32x4 *red4 = reinterpret_cast<f32x4 *>(smem);     // [4 writers][TPR][4][64 lanes] x 16 bytes
            for (int half = npx >> 1; half >= 1; half >>= 1) {
                const bool writer = active && pp >= half && pp < 2 * half;
                const bool reader = active && pp < half;
                const int slot = ((writer ? pp - half : pp) * nct + cg) & 3;
#pragma unroll
                for (int t0 = 0; t0 < NT; t0 += TPR) {
                    __syncthreads();                       // the tiles (first round) / the previous round are consumed
                    if (writer) {
#pragma unroll
                        for (int t = t0; t < (t0 + TPR < NT ? t0 + TPR : NT); ++t)
#pragma unroll
                            for (int v4 = 0; v4 < 4; ++v4) {
                                const f32x16 &a = acc[t / (RT * CT)][(t / CT) % RT][t % CT];
                                red4[((slot * TPR + (t - t0)) * 4 + v4) * 64 + lane] =
                                    f32x4{a[4 * v4], a[4 * v4 + 1], a[4 * v4 + 2], a[4 * v4 + 3]};
                            }
                    }
                    __syncthreads();
                    if (reader) {
#pragma unroll
                        for (int t = t0; t < (t0 + TPR < NT ? t0 + TPR : NT); ++t) {
#pragma unroll
                            for (int v4 = 0; v4 < 4; ++v4) {
                                const f32x4 r = red4[((slot * TPR + (t - t0)) * 4 + v4) * 64 + lane];
                                f32x16 &a = acc[t / (RT * CT)][(t / CT) % RT][t % CT];
                                a[4 * v4] += r[0]; a[4 * v4 + 1] += r[1]; a[4 * v4 + 2] += r[2]; a[4 * v4 + 3] += r[3];
                            }
                            __builtin_amdgcn_sched_barrier(0);     // one tile's 4 reads in flight, not all of them
                        }
                    }
                }
            }
            if (active && pp == 0) {
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    if (g0 + gi >= G) continue;            // uniform
                    float *dst = part + ((size_t)blockIdx.x * G + g0 + gi) * ((size_t)TP * MP);
#pragma unroll
                    for (int R = 0; R < RT; ++R)
#pragma unroll
                        for (int q = 0; q < CT; ++q) {
                            const int m = 32 * (CT * cg + q) + l32;
#pragma unroll
                            for (int v = 0; v < 16; ++v) {
                                const int tap = 32 * R + 8 * (v >> 2) + 4 * h + (v & 3);
                                if (tap < T && m < g.M) dst[(size_t)tap * MP + m] = acc[gi][R][q][v];
                            }
                        }
                }
            }
        }
    }
}

// dw[m][grp][tap] = alpha * sum over the tiles of part[tile][grp][tap][m]; 16 outputs x 16 strided partial
// sums per workgroup, combined in a fixed order.  The 16 outputs of a workgroup are 16 consecutive channels of one
// (grp, tap): every partial read is one 64-byte segment (with consecutive taps per workgroup the same reads were
// 256 bytes apart and fetched 4x the bytes).
__global__ __launch_bounds__(256) void k_wgm_fold(const float *__restrict__ part0, float *__restrict__ dw0, float alpha0,
                                                  int nparts, int G, int M, int T, int TP, int MP,
                                                  float *__restrict__ dw1, float alpha1)
{
    const float *__restrict__ part = part0 + (size_t)blockIdx.y * nparts * ((size_t)G * TP * MP);
    float *__restrict__ dw = blockIdx.y ? dw1 : dw0;
    const float alpha = blockIdx.y ? alpha1 : alpha0;
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, ps = threadIdx.x >> 4;
    const int mblocks = (M + 15) / 16;
    const int m = (blockIdx.x % mblocks) * 16 + o, r = blockIdx.x / mblocks;
    const int tap = r % T, grp = r / T;
    const int i = (m * G + grp) * T + tap;
    const bool valid = m < M;
    float s = 0.0f;
    if (valid) {
        const size_t off = ((size_t)grp * TP + tap) * MP + m, stride = (size_t)G * TP * MP;
        for (int t = ps; t < nparts; t += 16) s += part[(size_t)t * stride + off];
    }
    red[ps][o] = s;
    __syncthreads();
    if (ps == 0 && valid) {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][o];
        dw[i] = alpha * v;                                  // (M, C, Pd, Ph, Pw) = [m][grp][tap]
    }
}

struct Plan {
    int tilesX, tilesY, nct, npx, MP, TP, ng, ct, tpw, rs;
    size_t tiles, blocks, part_floats, lds;
};

bool plan_for(const cdl_geom *g, Plan *p)
{
    if (g->sw != g->sh || (g->sw != 1 && g->sw != 2)) return false;
    if (g->Pw != 3 && g->Pw != 5 && g->Pw != 7 && g->Pw != 9) return false;
    if (g->Ph != g->Pw && !(g->Ph == 9 && g->Pw == 5)) return false;      // rectangular planes: the shipped 9 x 9 x 5 net
    if (g->pw != g->Pw / 2 || g->ph != g->Ph / 2) return false;
    const int Dz = g->D / g->sd, Hz = g->H / g->sh, Wz = g->W / g->sw;
    const int MT = (g->M + 31) / 32;
    // one channel tile per wave whenever the waves divide evenly: half the accumulators (room for the fat
    // prefetch and the tree reduction without spills) for twice the -- now cheap -- LDS gathers; measured at cfg3 / cfg4:
    // 88.6 / 56.4 us against 94 / 70 with two tiles per wave
    const int CT = (MT == 1 || MT == 2 || MT == 4 || MT == 8) ? 1 : 2;
    p->ct = CT;
    p->nct = (MT + CT - 1) / CT;
    if (p->nct > 8) return false;                          // M <= 512
    p->npx = 8 / p->nct;
    if (KSTEPS % p->npx) return false;
    p->MP = p->nct * CT * 32;
    p->TP = ((g->Ph * g->Pw + 31) / 32) * 32;
    p->tilesX = (Wz + GLX - 1) / GLX;
    p->tilesY = (Hz + GLY - 1) / GLY;
    p->tiles = (size_t)g->N * Dz * p->tilesX * p->tilesY;
    p->tpw = 1;
    const size_t XH = (size_t)(GLY - 1) * g->sh + g->Ph, XW = (size_t)(GLX - 1) * g->sw + g->Pw;
    p->ng = 1;                                             // groups held at once: registers (NG*RT*CT*16 <= 192) and LDS permitting
    const int G = g->C * g->Pd, RT = p->TP / 32;
    const size_t plane = g->sw == 1 ? (size_t)eo_plane_dwords(1, g->Ph, g->Pw) * 4 * 4    // E/O x hi/lo dword planes
                                    : ((XH * XW + 7) / 8) * 8 * 2 * 2;                    // hi + lo bf16 planes of one group
    if (G >= 5 && RT == 1 && 5 * plane <= LDS_MAX) p->ng = 5;
    else if (G >= 3 && RT <= 2 && 3 * plane <= LDS_MAX) p->ng = 3;
    // a launch that leaves most CUs idle even with its group passes spread (launch_ct: blockIdx.z) does better with one group
    // per pass: three times the workgroups, a third of the accumulators each (no spills at two channel tiles per wave)
    if (p->ng == 3 && p->tiles * 2 * ((G + 2) / 3) * 2 <= (size_t)cdl_cu_count()) p->ng = 1;
    // row parts per tile (k_wgm): doubled while a paired launch, its group passes spread, would still leave half the CUs idle
    p->rs = 1;
    const size_t spread = p->tiles * 2 * ((G + p->ng - 1) / p->ng);
    while (p->rs < 8 && spread * (p->rs * 2) * 2 <= (size_t)cdl_cu_count() && KSTEPS / (p->npx * p->rs * 2) >= 8) p->rs *= 2;
    p->blocks = p->tiles * p->rs;                          // upper bound (the launch groups work items by the CU count)
    p->part_floats = p->blocks * g->C * g->Pd * p->TP * p->MP;
    p->lds = p->ng * plane;
    if (p->lds < 8 * 16 * 64 * 4) p->lds = 8 * 16 * 64 * 4;  // the cross-wave reduction buffer reuses it
    if (p->lds > LDS_MAX) return false;
    // too few workgroups: the VALU kernels do better -- except under deep filters (C Pd Ph Pw >= 256 taps, the 9 x 9 x 5
    // net at batch 1: 16 tiles), where k_wgrad_l takes 0.76 ms a launch
    // -- and under many channels (M > 64: `k_wgrad_p` takes 85 us for one 128 x 128 crop at M = 169), where the row parts
    // above spread the few tiles
    const size_t min_tiles = MT >= 3 ? 2 : (size_t)G * g->Ph * g->Pw >= 256 ? 8 : 64;
    if (p->tiles < min_tiles || p->tiles >= ((size_t)1 << 31)) return false;
    if (p->part_floats > ((size_t)1 << 27)) return false;               // 512 MiB of partials: not worth it
    return true;
}

template <int PH, int PW, int SW, int NG, int CT>
int launch_ct(const cdl_geom *g, const Plan &p, const float *F, const float *gate, const float *x, float *ws,
              hipStream_t st, const float *F1, const float *x1, int rsc)
{
    if (int rc = cdl_ensure_dynamic_lds((const void *)k_wgm<PH, PW, SW, NG, CT>, LDS_MAX)) return rc;
    // group passes over workgroups while the launch would leave CUs idle (gspan a multiple of NG)
    const int G = g->C * g->Pd, passes = (G + NG - 1) / NG;
    const size_t wgs = p.blocks * (F1 ? 2 : 1), cus = (size_t)cdl_cu_count();
    int gz = 1;
    while (gz < passes && wgs * (gz + 1) <= cus) ++gz;
    const int gspan = ((passes + gz - 1) / gz) * NG;
    gz = (G + gspan - 1) / gspan;
    k_wgm<PH, PW, SW, NG, CT><<<dim3((unsigned)p.blocks, F1 ? 2 : 1, (unsigned)gz), GNT, p.lds, st>>>(
        *g, F, gate, x, ws, p.tilesX, p.tilesY, p.nct, p.MP CDL_DBG_COMMA(cdl_opts().fused_debug & (1024 | 2048 | 4096)), (int)(p.tiles * p.rs),
        p.tpw, F1, x1, rsc, gspan, p.rs);
    CDL_LAUNCH_CHECK();
    return 0;
}

template <int PH, int PW, int SW, int NG>
int launch_ng(const cdl_geom *g, const Plan &p, const float *F, const float *gate, const float *x, float *ws,
              hipStream_t st, const float *F1, const float *x1, int rsc)
{
    return p.ct == 1 ? launch_ct<PH, PW, SW, NG, 1>(g, p, F, gate, x, ws, st, F1, x1, rsc)
                     : launch_ct<PH, PW, SW, NG, 2>(g, p, F, gate, x, ws, st, F1, x1, rsc);
}

template <int PH, int PW, int SW>
int launch(const cdl_geom *g, const Plan &p, const float *F, const float *gate, const float *x, float alpha,
           float *dw, float *ws, hipStream_t st, const float *F1, const float *x1, float alpha1, float *dw1, int rsc)
{
    constexpr int RT = (PH * PW + 31) / 32;
    int rc;
    if (RT == 1 && p.ng == 5) rc = launch_ng<PH, PW, SW, (RT == 1 ? 5 : 1)>(g, p, F, gate, x, ws, st, F1, x1, rsc);
    else if (RT <= 2 && p.ng == 3) rc = launch_ng<PH, PW, SW, (RT <= 2 ? 3 : 1)>(g, p, F, gate, x, ws, st, F1, x1, rsc);
    else rc = launch_ng<PH, PW, SW, 1>(g, p, F, gate, x, ws, st, F1, x1, rsc);
    if (rc) return rc;
    const int G = g->C * g->Pd, T = g->Ph * g->Pw;
    k_wgm_fold<<<dim3(G * T * ((g->M + 15) / 16), F1 ? 2 : 1), 256, 0, st>>>(ws, dw, alpha, (int)p.blocks, G, g->M, T,
                                                                             p.TP, p.MP, dw1, alpha1);
    CDL_LAUNCH_CHECK();
    return 0;
}

}  // namespace

size_t cdl_mfma_wgrad_ws_floats(const cdl_geom *g)
{
    Plan p;
    return plan_for(g, &p) ? p.part_floats : 0;
}

// CDL_EUNSUPPORTED: the caller falls back to the VALU kernels
static int wgrad_entry(const cdl_geom *g, const float *F, const float *gate, const float *x, float alpha, float *dw,
                       const float *F1, const float *x1, float alpha1, float *dw1, float *ws, size_t ws_floats,
                       void *stream, int rsc = 0)
{
    Plan p;
    if (!plan_for(g, &p) || !ws) return CDL_EUNSUPPORTED;
    if (rsc && gate) return CDL_EUNSUPPORTED;
    const size_t jobs = F1 ? 2 : 1;
    const size_t cus = (size_t)cdl_cu_count();              // one workgroup per CU at a time (registers): tiles per
    const size_t items = p.tiles * p.rs;                    // (tile, row part) work items
    p.tpw = (int)((jobs * items + cus - 1) / cus);          // workgroup = the number of rounds an item-per-workgroup grid takes
    p.blocks = (items + p.tpw - 1) / p.tpw;
    if (ws_floats < jobs * p.blocks * ((size_t)g->C * g->Pd * p.TP * p.MP)) return CDL_EUNSUPPORTED;
#define CDL_M(PH_, P_, S_)                                   \
    if (g->Ph == PH_ && g->Pw == P_ && g->sw == S_)          \
        return launch<PH_, P_, S_>(g, p, F, gate, x, alpha, dw, ws, S(stream), F1, x1, alpha1, dw1, rsc)
    CDL_M(3, 3, 1); CDL_M(5, 5, 1); CDL_M(7, 7, 1); CDL_M(9, 9, 1);
    CDL_M(3, 3, 2); CDL_M(5, 5, 2); CDL_M(7, 7, 2); CDL_M(9, 9, 2);
    CDL_M(9, 5, 1); CDL_M(9, 5, 2);
#undef CDL_M
    return CDL_EUNSUPPORTED;
}

int cdl_mfma_wgrad(const cdl_geom *g, const float *F, const float *gate, const float *x, float alpha, float *dw,
                   float *ws, size_t ws_floats, void *stream)
{
    return wgrad_entry(g, F, gate, x, alpha, dw, nullptr, nullptr, 0.0f, nullptr, ws, ws_floats, stream);
}

// Two ungated filter gradients of the same geometry (dA_k = a0 F0 (x) x0, dB_k = a1 F1 (x) x1) in ONE launch + ONE fold
int cdl_mfma_wgrad_pair(const cdl_geom *g, const float *F0, const float *x0, float alpha0, float *dw0, const float *F1,
                        const float *x1, float alpha1, float *dw1, float *ws, size_t ws_floats, void *stream)
{
    return wgrad_entry(g, F0, nullptr, x0, alpha0, dw0, F1, x1, alpha1, dw1, ws, ws_floats, stream);
}

// The same with the fat operands in the strip kernel's row-strip channel-major layout (cdl_strip.hip; rsc != 0), for the
// reverse sweep of the strip shapes.  cdl_mfma_wgrad_takes: this kernel (and not a VALU fallback, which reads the
// reference layout only) is what cdl_wgrad would run for the geometry.
bool cdl_mfma_wgrad_takes(const cdl_geom *g)
{
    Plan p;
    return cdl_opts().mfma_wgrad && plan_for(g, &p);
}

int cdl_mfma_wgrad_lay(const cdl_geom *g, const float *F, const float *x, float alpha, float *dw, float *ws,
                       size_t ws_floats, int rsc, void *stream)
{
    return wgrad_entry(g, F, nullptr, x, alpha, dw, nullptr, nullptr, 0.0f, nullptr, ws, ws_floats, stream, rsc);
}

int cdl_mfma_wgrad_pair_lay(const cdl_geom *g, const float *F0, const float *x0, float alpha0, float *dw0,
                            const float *F1, const float *x1, float alpha1, float *dw1, float *ws, size_t ws_floats,
                            int rsc, void *stream)
{
    return wgrad_entry(g, F0, nullptr, x0, alpha0, dw0, F1, x1, alpha1, dw1, ws, ws_floats, stream, rsc);
}
