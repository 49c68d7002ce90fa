// Strip kernel for the grouped shapes: any number of image channels C, 2-D or 3-D (depth taps Pd), unit stride, square
// filter planes P in {3, 5, 7}, M <= 64 -- BASELINE configs[2] (CDLNetVideo K=20 M=48 P=5x5x5: G = C * Pd = 5 groups) and
// configs[3] (JDD: C = 3, M = 64, P = 7, Bayer mask: G = 3).  Reference: the loop bodies model/net.py:87 and :205,
// z = ST(z - A_k(mask * B_k z - yp), tau_k).
//
// Same launch contract as the tile kernel of cdl_fusedg.hip (k_stage_g), which it replaces behind the cdl_fusedg_* entry
// points, and the same arithmetic per 32-pixel code row (analysis-like GEMM over the G*P*P taps gathered from bf16 hi|lo
// planes in LDS, epilogue in registers, synthesis-like GEMM per group fed from the accumulators, col2im by DPP Horner
// chains and a P-row register ring per group).  What differs is the WORK DECOMPOSITION of cdl_strip.hip: a wave is an
// autonomous worker walking down a strip of 32 code columns of one (n, depth) plane,
//   * one barrier per launch (weights), none per tile: k_stage_g spent 44 % of a cfg3 launch in per-tile phases (thin
//     staging, patch combine) that all 8 waves of a workgroup went through in lock-step with no fat traffic in flight
//     (profiles/r02_g_fusedg_ablation_cfg3.jsonl);
//   * thin input in a per-wave circular row buffer (G planes x (8 + P - 1) rows x (32 + P - 1) columns of (hi | lo) dwords):
//     every image row is loaded once per strip, one new row per code row;
//   * finished patch rows go straight from the ring registers to the item's patch in global memory (no LDS patch, no
//     combine); k_assemble_sg sums the <= 2 x 2 overlapping patches and the depth taps in a fixed order.
// The fat input of a code row is loaded INTO the accumulator registers (C-in of the matrix cores; the analysis-like
// fragments carry the iteration's sign), one row ahead.  Deterministic, no atomics.
#include <type_traits>

#include "cdl_strip.h"

namespace {

#include "cdl_strip_dev.h"

struct GSParams {
    const float *r;          // (N,C,D,H,W)
    const float *zin;        // (N,M,D,H,W) / CDL_LAY_RSC, or nullptr
    unsigned *map;           // (N,4,D,H,W) words: plane 2h = support, 2h+1 = sign, bit 16R + v (as k_stage_g)
    float *zout;
    const float *tau;        // (N,M)
    float *dtau;             // (items, M)
    const uint4 *frags;      // cdl_fusedg.hip's prepared pair: [A hi | A lo | B hi | B lo]
    float *patches;          // (items, G, prows, PXW)
    float sgn;
    int do_synth;
    int N, C, M, D, H, W, Pd, KQ, nsx, nsy, SEG, prows, rev, items, lay_in, lay_out;
    CDL_DBG_FIELD(int dbg;)  // probe build only (CDL_FUSED_DEBUG; results are wrong): 1 no fat loads, 2 no fat stores,
                             // 4 no analysis-like MFMAs, 8 no synthesis-like MFMAs, 16 no col2im, 32 no gather
};

#ifndef CDL_STRIPG_SYNTH_PIPE
#define CDL_STRIPG_SYNTH_PIPE 1
#endif
constexpr bool SYNTH_PIPE = CDL_STRIPG_SYNTH_PIPE != 0;

struct GCarve {
    int wb, koff, wave0, wave_bytes, tau_off, total;
};
template <int P>
__host__ __device__ inline GCarve gcarve(int MT, int KS, int KQ, int G)
{
    constexpr int RT = Shape<P>::RT;
    GCarve c;
    c.wb = MT * KS * 2 * 1024;
    c.koff = c.wb + G * RT * KQ * 2 * 1024;
    c.wave0 = c.koff + KS * 16 * 4;
    c.tau_off = G * Shape<P>::RBUF * Strip<P, 1>::XWP * 4;
    c.wave_bytes = c.tau_off + 64 * 4;
    c.total = c.wave0 + NWV * c.wave_bytes;
    return c;
}

template <int P, int G, int MT, int MODE, bool MAPPED>
__global__ __launch_bounds__(NTS) void k_stripg(GSParams p)
{
    using SH = Shape<P>;
    using ST_ = Strip<P, 1>;
    constexpr int RT = SH::RT, RC = SH::RC, RBUF = SH::RBUF, HALO = P / 2;
    constexpr int XW = ST_::XW, XWP = ST_::XWP, PXW = ST_::PXW;
    constexpr int K = G * P * P, KS = (K + 15) / 16;
    constexpr int NLD = (G * XW + 63) / 64;                          // thin loads per lane and image row (all G planes)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = p.M, KQ = p.KQ;
    const GCarve cv = gcarve<P>(MT, KS, KQ, G);
    const uint4 *wa = reinterpret_cast<const uint4 *>(smem);
    const uint4 *wb = reinterpret_cast<const uint4 *>(smem + cv.wb);
    int *koff = reinterpret_cast<int *>(smem + cv.koff);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);          // (provably wave-uniform: see cdl_strip.hip)
    const int c = lane & 31, h = lane >> 5;
    const int FA = MT * KS, FB = G * RT * KQ;
    unsigned char *wbase = smem + cv.wave0 + wid * cv.wave_bytes;
    unsigned *thin = reinterpret_cast<unsigned *>(wbase);               // [G][RBUF][XWP] dwords: bf16 hi | bf16 lo
    float *tau_s = reinterpret_cast<float *>(wbase + cv.tau_off);       // [64]

    // ---- once per workgroup: weight fragments (the analysis-like ones with the iteration's sign) and the tap table
    const bool negA = MODE != MODE_BWD && p.sgn < 0.0f;
    {
        uint4 *wdst = reinterpret_cast<uint4 *>(smem);
        const int nfr = 2 * (FA + FB) * 64;
        for (int i = tid; i < nfr; i += 4 * NTS) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p.frags[min(i + u * NTS, nfr - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * NTS < nfr) {
                    if (negA && i + u * NTS < 2 * FA * 64) {
                        v[u].x ^= 0x80008000u; v[u].y ^= 0x80008000u; v[u].z ^= 0x80008000u; v[u].w ^= 0x80008000u;
                    }
                    wdst[i + u * NTS] = v[u];
                }
        }
        for (int k = tid; k < KS * 16; k += NTS) {
            int o = 0;
            if (k < K) {
                const int kj = k % P, q = k / P;
                o = ((q / P) * RBUF + (q % P)) * XWP + kj;              // plane g = q / P, filter row ki = q % P
            }
            koff[k] = o * 4;
        }
    }
    __syncthreads();
    auto afrag = [&](int f) { return __builtin_bit_cast(bf16x8, wa[f * 64 + lane]); };

    const int HW = p.H * p.W, DHW = p.D * HW;
    const int up_addr = ((lane & 31) + 32) * 4;
    const int nwaves = gridDim.x * NWV;
    const bool has_base = (MODE == MODE_FWD) || (MODE == MODE_BWD && p.zin != nullptr);
    const int thin_base = (int)(wbase - smem);

#pragma unroll 1
    for (int it = blockIdx.x * NWV + wid; it < p.items; it += nwaves) {
        int bid = p.rev ? p.items - 1 - it : it;
        const int item = bid;
        const int sx = bid % p.nsx; bid /= p.nsx;
        const int sy = bid % p.nsy; bid /= p.nsy;
        const int zd = bid % p.D, n = bid / p.D;
        const int xz0 = sx * 32, yz0 = sy * p.SEG;
        const int x = xz0 + c;
        const bool xok = x < p.W;
        const int nblk = min(p.SEG, p.H - yz0);
        const bool edge = __ballot(!xok) != 0ull;                        // uniform

        if (MODE != MODE_BWD) tau_s[lane] = lane < M ? p.tau[(size_t)n * M + lane] : 0.0f;

        // thin row group bb: image row yz0 + bb + HALO of the G planes (plane g = (c, kd): depth zd - Pd/2 + kd of channel c).
        // Per item and lane: the byte offset of its element's (plane, column) in r (out of range where the plane or the
        // column does not exist: the buffer load then returns 0) and its dword in the wave's LDS planes; per row only the
        // image row moves (a scalar offset).  (Recomputing plane / column per row cost ~150 vector instructions, a third
        // of them quarter-rate integer multiplies.)
        const __amdgpu_buffer_rsrc_t rs_thin = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(p.r), 0, (int)((size_t)p.N * p.C * DHW * 4), 0x00020000);
        int t_off[NLD], t_lds[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = lane + 64 * k;
            const int g = min(idx / XW, G - 1), col = idx - (idx / XW) * XW;
            const int kd = g % p.Pd, cc = g / p.Pd;
            const int d = zd - p.Pd / 2 + kd;
            const int xx = xz0 - HALO + col;
            const bool ok = idx < G * XW && d >= 0 && d < p.D && xx >= 0 && xx < p.W;
            t_off[k] = ok ? (((n * p.C + cc) * p.D + d) * HW + xx) * 4 : OOB;
            t_lds[k] = idx < G * XW ? (g * RBUF) * XWP + col : -1;
        }
        float tn[NLD];
        auto thin_issue = [&](int bb, float (&t)[NLD]) __attribute__((always_inline)) {
            const int yy = yz0 + bb + HALO;
            const bool rok = yy >= 0 && yy < p.H;                        // uniform
            const int srow = __builtin_amdgcn_readfirstlane(min(max(yy, 0), p.H - 1) * p.W * 4);
#pragma unroll
            for (int k = 0; k < NLD; ++k)
                t[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_thin, rok ? t_off[k] : OOB, srow, 0));
        };
        auto thin_commit = [&](int bb, const float (&t)[NLD]) __attribute__((always_inline)) {
            const int slot = (yz0 + bb + HALO + 8 * RC) & (RC - 1);
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const float v = t[k];
                const __bf16 hh = (__bf16)v;
                const __bf16 ll = (__bf16)(v - (float)hh);
                const unsigned w = (unsigned)__builtin_bit_cast(unsigned short, hh) |
                                   ((unsigned)__builtin_bit_cast(unsigned short, ll) << 16);
                if (t_lds[k] >= 0) {
                    thin[t_lds[k] + slot * XWP] = w;
                    if (slot < P - 1) thin[t_lds[k] + (slot + RC) * XWP] = w;
                }
            }
        };
        {
            float tp[P][NLD];
#pragma unroll
            for (int gI = 0; gI < P; ++gI) thin_issue(gI - (P - 1), tp[gI]);
            if (nblk > 1) thin_issue(1, tn);
#pragma unroll
            for (int gI = 0; gI < P; ++gI) thin_commit(gI - (P - 1), tp[gI]);
        }

        // fat addressing (cdl_strip.hip): layout 0 = (N,M,D,H,W), 1 = row-strip channel-major [n][d][y][strip][M][32 px]
        const int bytes_in = p.lay_in ? M * p.D * p.H * p.nsx * 128 : M * DHW * 4;
        const int bytes_out = p.lay_out ? M * p.D * p.H * p.nsx * 128 : M * DHW * 4;
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
            has_base ? const_cast<float *>(p.zin) + (size_t)n * (bytes_in / 4) : p.zout, 0, has_base ? bytes_in : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
            p.zout + (size_t)n * (bytes_out / 4), 0, bytes_out, 0x00020000);
        const int estr_in = p.lay_in ? 128 : DHW * 4, estr_out = p.lay_out ? 128 : DHW * 4;
        const int vlane_in = xok ? (p.lay_in ? (4 * h * 32 + c) * 4 : (4 * h * DHW + x) * 4) : OOB;
        const int vlane_out = xok ? (p.lay_out ? (4 * h * 32 + c) * 4 : (4 * h * DHW + x) * 4) : OOB;
        auto sbase_in = [&](int yz, int R) {
            return __builtin_amdgcn_readfirstlane(p.lay_in ? (((zd * p.H + yz) * p.nsx + sx) * M + 32 * R) * 128
                                                           : (zd * HW + yz * p.W + 32 * R * DHW) * 4);
        };
        auto sbase_out = [&](int yz, int R) {
            return __builtin_amdgcn_readfirstlane(p.lay_out ? (((zd * p.H + yz) * p.nsx + sx) * M + 32 * R) * 128
                                                            : (zd * HW + yz * p.W + 32 * R * DHW) * 4);
        };
        unsigned *const map_n = p.map ? p.map + (((size_t)n * 4 + 2 * h) * p.D + zd) * HW : nullptr;
        // (M is a multiple of 8 here, as for the tile kernel: a register quad of 4 channels x 2 lane halves is all real or
        //  all padding, padding quads are skipped by uniform branches and never reach an address)
        const int vo_none[4] = {OOB, OOB, OOB, OOB};

        // fat input of a code row (all MT channel tiles) into THE accumulator registers: the loads of row b+1 are issued as
        // soon as row b's codes have been split into their bf16 operands (the registers are free then) and fly during its
        // synthesis-like GEMM and col2im -- a second register set for a whole row ahead spilled (and every spill reload
        // waits on the prefetch in front of it)
        f32x16 acc[MT];
        unsigned sup = 0, sgb = 0;                                      // reverse: the row's map words (both tiles)
        auto fat_issue = [&](int yz, f32x16 (&dst)[MT], unsigned &sup, unsigned &sgb) __attribute__((always_inline)) {
            const bool loads = !(MODE == MODE_FIRST || !has_base || CDL_DBG(p.dbg, 1));       // uniform
            // (all 16 * MT registers are zeroed, also the ones a load is about to fill: zeroing only the padding quads
            //  left the sets partially defined across the row loop and the allocator spilled 80 registers)
#pragma unroll
            for (int R = 0; R < MT; ++R)
#pragma unroll
                for (int v = 0; v < 16; ++v) dst[R][v] = 0.0f;
            if (MODE == MODE_BWD) {
                sup = 0; sgb = 0;
                if (xok) {
                    sup = map_n[(size_t)yz * p.W + x];
                    sgb = map_n[(size_t)DHW + (size_t)yz * p.W + x];
                }
            }
            if (!loads) return;
#pragma unroll
            for (int R = 0; R < MT; ++R) {
                const int so0 = sbase_in(yz, R);
#pragma unroll
                for (int qv = 0; qv < 4; ++qv) {
                    const int cb = 32 * R + 8 * qv;
                    if (cb >= M) continue;                                // uniform
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        dst[R][4 * qv + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rs_in, vlane_in, so0 + (8 * qv + e) * estr_in, 0));
                }
            }
        };

        float ring[G][P];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < P; ++i) ring[g][i] = 0.0f;
        float tacc[MT];
#pragma unroll
        for (int R = 0; R < MT; ++R) tacc[R] = 0.0f;
        float *patch = p.patches + (size_t)item * G * p.prows * PXW;
        const bool tau_neg = MODE != MODE_BWD && __ballot(!(tau_s[lane] >= 0.0f)) != 0ull;

        fat_issue(yz0, acc, sup, sgb);

#pragma unroll 1
        for (int b = 0; b < nblk; ++b) {
            const int yz = yz0 + b;
            const unsigned csup = sup, csgb = sgb;
            // ---- analysis-like GEMM: im2col gathered through the tap table, depth-2 software pipeline over the k-steps
            const int s0 = (yz - HALO + 8 * RC) & (RC - 1);
            const int pixbase = thin_base + (s0 * XWP + c) * 4;
            auto gather = [&](int ks, bf16x8 &gh, bf16x8 &gl) __attribute__((always_inline)) {
                if (CDL_DBG(p.dbg, 32)) { const u32x4 c4 = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
                                          gh = __builtin_bit_cast(bf16x8, c4); gl = gh; return; }
                const int4 o0 = *reinterpret_cast<const int4 *>(koff + 16 * ks + 8 * h);
                const int4 o1 = *reinterpret_cast<const int4 *>(koff + 16 * ks + 8 * h + 4);
                const int oo[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
                unsigned w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i] = *reinterpret_cast<const unsigned *>(smem + (oo[i] + pixbase));
                u32x4 hv, lv;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hv[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x05040100u);   // low halves
                    lv[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x07060302u);   // high halves
                }
                // code columns beyond the plane (last strip only: uniform branch) get a zero operand: their accumulators
                // and (the fat loads being out of range) their codes stay exactly zero
                if (edge) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { hv[i] = xok ? hv[i] : 0u; lv[i] = xok ? lv[i] : 0u; }
                }
                gh = __builtin_bit_cast(bf16x8, hv);
                gl = __builtin_bit_cast(bf16x8, lv);
            };
            {
                bf16x8 bh[2], bl[2];
                gather(0, bh[0], bl[0]);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int cu = ks & 1;
                    if (ks + 1 < KS) gather(ks + 1, bh[cu ^ 1], bl[cu ^ 1]);
                    if (CDL_DBG(p.dbg, 4)) continue;
                    bf16x8 ah[MT], al[MT];
#pragma unroll
                    for (int R = 0; R < MT; ++R) { ah[R] = afrag(R * KS + ks); al[R] = afrag(FA + R * KS + ks); }
#pragma unroll
                    for (int R = 0; R < MT; ++R) acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[R], bh[cu], acc[R], 0, 0, 0);
#pragma unroll
                    for (int R = 0; R < MT; ++R) acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[R], bl[cu], acc[R], 0, 0, 0);
#pragma unroll
                    for (int R = 0; R < MT; ++R) acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[R], bh[cu], acc[R], 0, 0, 0);
                }
            }
            // ---- thin rows: commit the row of the next code row (loaded one row ago), fetch the one after
            if (b + 1 < nblk) thin_commit(b + 1, tn);
            if (b + 2 < nblk) thin_issue(b + 2, tn);
            __builtin_amdgcn_sched_barrier(0);
            // ---- epilogue per channel tile
            unsigned wsw = 0, wgw = 0;
            const bool nostore = CDL_DBG(p.dbg, 2);
#pragma unroll
            for (int R = 0; R < MT; ++R) {
                float tsum[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) tsum[v] = 0.0f;
                unsigned ws = 0, wg = 0;
                const int so0 = sbase_out(yz, R);
                if (tau_neg)
                    strip_epilogue<MODE, MAPPED, true, false>(acc[R], tsum, ws, wg, tau_s, R, M, h, -8, vo_none, vlane_out,
                                                              rs_out, so0, estr_out, csup >> (16 * R), csgb >> (16 * R), nostore);
                else
                    strip_epilogue<MODE, MAPPED, false, false>(acc[R], tsum, ws, wg, tau_s, R, M, h, -8, vo_none, vlane_out,
                                                               rs_out, so0, estr_out, csup >> (16 * R), csgb >> (16 * R), nostore);
                wsw |= ws << (16 * R);
                wgw |= (wg & ws) << (16 * R);
                if (MODE == MODE_BWD) tacc[R] += LaneTransposeSum<16>::run(tsum, c);
            }
            if (MODE != MODE_BWD && MAPPED && xok) {
                map_n[(size_t)yz * p.W + x] = wsw;
                map_n[(size_t)DHW + (size_t)yz * p.W + x] = wgw;
            }
            if (MODE == MODE_BWD && !p.do_synth) {
                if (b + 1 < nblk) fat_issue(yz + 1, acc, sup, sgb);
                continue;
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- synthesis-like GEMM per group: the accumulator tiles are the B operand (k = channel), split once
            bf16x8 zh[2 * MT], zl[2 * MT];
#pragma unroll
            for (int q = 0; q < 2 * MT; ++q) {
                u32x4 zhw, zlw;
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const float v0 = acc[q >> 1][8 * (q & 1) + 2 * e2], v1 = acc[q >> 1][8 * (q & 1) + 2 * e2 + 1];
                    const bf16x2 hh = __builtin_convertvector(f32x2{v0, v1}, bf16x2);
                    const unsigned hb = __builtin_bit_cast(unsigned, hh);
                    const float r0 = v0 - __builtin_bit_cast(float, hb << 16);
                    const float r1 = v1 - __builtin_bit_cast(float, hb & 0xffff0000u);
                    const bf16x2 ll = __builtin_convertvector(f32x2{r0, r1}, bf16x2);
                    zhw[e2] = hb;
                    zlw[e2] = __builtin_bit_cast(unsigned, ll);
                }
                zh[q] = __builtin_bit_cast(bf16x8, zhw);
                zl[q] = __builtin_bit_cast(bf16x8, zlw);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float keep0 = acc[0][0], keep1 = acc[MT - 1][15];     // (probe build: bits 8 / 16 read them)
            (void)keep0; (void)keep1;
            if (b + 1 < nblk) fat_issue(yz + 1, acc, sup, sgb);          // the accumulators are free: next row's fat input
            // tap tiles t = g * RT + Rt as a depth-2 pipeline: the products of tile t+1 are issued before the col2im of
            // tile t, so the matrix pipe works under the DPP chains
            // (fragment addresses: one running per-lane base per tap tile + compile-time offsets for the k-steps.  KQ is a
            //  run-time value, and with  bfrag(t * KQ + q)  the compiler precomputed all 2 * G * RT * KQ addresses outside
            //  the row loop and spilled them)
            int fb_hi = lane * 16, fb_lo = lane * 16 + FB * 1024;
            asm volatile("" : "+v"(fb_hi), "+v"(fb_lo));
            auto mm = [&](int t) __attribute__((always_inline)) {
                // (the first product takes the zero CONSTANT as its C operand: no 16 v_mov per tap tile)
                const f32x16 zero16 = f32x16{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                f32x16 Dt = zero16;
                const unsigned char *wbb = reinterpret_cast<const unsigned char *>(wb);
#pragma unroll
                for (int q = 0; q < 2 * MT; ++q) {
                    if (q >= KQ) continue;                               // uniform
                    if (CDL_DBG(p.dbg, 8)) { Dt[0] += keep0; continue; }
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(wbb + fb_hi + q * 1024));
                    const bf16x8 wl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(wbb + fb_lo + q * 1024));
                    Dt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, zh[q], Dt, 0, 0, 0);
                    Dt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, zl[q], Dt, 0, 0, 0);
                    Dt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, zh[q], Dt, 0, 0, 0);
                }
                fb_hi += KQ * 1024;                                      // next tap tile (the calls come in ascending t)
                fb_lo += KQ * 1024;
                return Dt;
            };
            auto c2i = [&](int t, const f32x16 &Dt) __attribute__((always_inline)) {
                const int g = t / RT, Rt = t % RT;
                if (CDL_DBG(p.dbg, 16)) { ring[g][0] += Dt[0] + Dt[15]; return; }
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    if ((tap_slot<P>(i, 0) >> 5) != Rt) continue;        // compile time: this filter row is in the other tile
                    float sr = 0.0f;
#pragma unroll
                    for (int j = P - 1; j >= 0; --j) {
                        const int slot = tap_slot<P>(i, j);
                        const int v = 4 * ((slot >> 3) & 3) + (slot & 3), hh = (slot >> 2) & 1;
                        float val = Dt[v];
                        if (hh) val = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(up_addr, __builtin_bit_cast(int, val)));
                        sr = wave_shr1(sr) + (h == 0 ? val : 0.0f);
                    }
                    ring[g][i] += sr;
                }
                if (Rt == RT - 1) {
                    // ring slot 0 is complete: row b of the item's patch, straight to global memory
                    if (lane < PXW) patch[((size_t)g * p.prows + b) * PXW + lane] = ring[g][0];
#pragma unroll
                    for (int i = 0; i + 1 < P; ++i) ring[g][i] = ring[g][i + 1];
                    ring[g][P - 1] = 0.0f;
                }
            };
            constexpr int NTT = G * RT;
            // (the reverse mode keeps the threshold partial sums besides the rings: the second accumulator tile spills there)
            constexpr bool SPIPE = SYNTH_PIPE && !(MODE == MODE_BWD);
            f32x16 Dd[2];
            Dd[0] = mm(0);
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                if (SPIPE) {
                    if (t + 1 < NTT) Dd[(t + 1) & 1] = mm(t + 1);
                    c2i(t, Dd[t & 1]);
                } else {
                    c2i(t, Dd[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 1 < NTT) Dd[0] = mm(t + 1);
                }
            }
        }
        // ---- the P-1 rows below the item's last code row are still in the rings
        if (MODE != MODE_BWD || p.do_synth) {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int i = 0; i + 1 < P; ++i)
                    if (lane < PXW) patch[((size_t)g * p.prows + nblk + i) * PXW + lane] = ring[g][i];
        }
        if (MODE == MODE_BWD) {
#pragma unroll
            for (int R = 0; R < MT; ++R) {
                const float tot = tacc[R] + __shfl_xor(tacc[R], 16, 64);
                const int ch = 32 * R + 8 * (c >> 2) + 4 * h + (c & 3);
                if (c < 16 && ch < M) p.dtau[(size_t)item * M + ch] = tot;
            }
        }
    }
}

// out[n,c,d,Y,X] = (mask ? mask : 1) * alpha * (sum over depth taps kd and covering items of the patches) - (sub ? sub : 0);
// patch position (Y + HALO, X + HALO) of item (n, zd = d + Pd/2 - kd, segment, strip), group c * Pd + kd.  Fixed order.
template <int P>
__global__ __launch_bounds__(256) void k_assemble_sg(const float *__restrict__ patches, const float *__restrict__ mask,
                                                     const float *__restrict__ sub, float alpha, float *__restrict__ out,
                                                     int N, int C, int D, int H, int W, int Pd, int nsx, int nsy, int SEG,
                                                     int prows)
{
    constexpr int HALO = P / 2, PXW = Strip<P, 1>::PXW;
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
    if (X >= W) return;
    int r = blockIdx.z;
    const int d = r % D; r /= D;
    const int c = r % C, n = r / C;
    const int G = C * Pd;
    const int nrow = Y + HALO, m = X + HALO;
    const int sy_hi = min(nsy - 1, nrow / SEG), sx_hi = min(nsx - 1, m / 32);
    const bool y_lo = sy_hi > 0 && nrow - (sy_hi - 1) * SEG < SEG + P - 1;
    const bool x_lo = sx_hi > 0 && m - (sx_hi - 1) * 32 < 32 + P - 1;
    const size_t pplane = (size_t)prows * PXW;
    float sum = 0.0f;
    for (int kd = 0; kd < Pd; ++kd) {
        const int zd = d + Pd / 2 - kd;                                  // the code depth whose tap kd lands on d
        if (zd < 0 || zd >= D) continue;
        const int g = c * Pd + kd;
        auto at = [&](int sy, int sx) {
            const size_t item = (((size_t)n * D + zd) * nsy + sy) * nsx + sx;
            return patches[(item * G + g) * pplane + (size_t)(nrow - sy * SEG) * PXW + (m - sx * 32)];
        };
        if (y_lo) {
            if (x_lo) sum += at(sy_hi - 1, sx_hi - 1);
            sum += at(sy_hi - 1, sx_hi);
        }
        if (x_lo) sum += at(sy_hi, sx_hi - 1);
        sum += at(sy_hi, sx_hi);
    }
    const size_t i = ((((size_t)n * C + c) * D + d) * H + Y) * W + X;
    float v = alpha * sum;
    if (mask) v *= mask[i];
    if (sub) v -= sub[i];
    out[i] = v;
}

template <int P, int G, int MT, int MODE, bool MAPPED>
int launch_one(const GSParams &p, const cdl_stripg_plan &pl, hipStream_t st)
{
    const int lds = gcarve<P>(MT, pl.KS, pl.KQ, G).total;
    if (int rc = cdl_ensure_dynamic_lds((const void *)k_stripg<P, G, MT, MODE, MAPPED>, lds)) return rc;
    size_t cus = (size_t)cdl_cu_count();
    const int cap = cdl_opts().fused_grid;
    if (cap > 0 && (size_t)cap < cus) cus = (size_t)cap;
    const size_t wgs = (pl.items + NWV - 1) / NWV;
    const unsigned grid = (unsigned)(wgs < cus ? wgs : cus);
    k_stripg<P, G, MT, MODE, MAPPED><<<grid, NTS, lds, st>>>(p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <int P, int G, int MT>
int launch_mode(const GSParams &p, const cdl_stripg_plan &pl, int mode, hipStream_t st)
{
    if (mode == MODE_FWD)
        return p.map ? launch_one<P, G, MT, MODE_FWD, true>(p, pl, st) : launch_one<P, G, MT, MODE_FWD, false>(p, pl, st);
    if (mode == MODE_FIRST)
        return p.map ? launch_one<P, G, MT, MODE_FIRST, true>(p, pl, st) : launch_one<P, G, MT, MODE_FIRST, false>(p, pl, st);
    return launch_one<P, G, MT, MODE_BWD, true>(p, pl, st);
}

template <int P, int G>
int launch_g(const GSParams &p, const cdl_stripg_plan &pl, int mode, hipStream_t st)
{
    return pl.MT == 2 ? launch_mode<P, G, 2>(p, pl, mode, st) : launch_mode<P, G, 1>(p, pl, mode, st);
}

template <int P>
int launch_p(const GSParams &p, const cdl_stripg_plan &pl, int mode, hipStream_t st)
{
    if (pl.G == 1) return launch_g<P, 1>(p, pl, mode, st);
    if (pl.G == 3) return launch_g<P, 3>(p, pl, mode, st);
    if (pl.G == 5) return launch_g<P, 5>(p, pl, mode, st);
    return launch_g<P, 7>(p, pl, mode, st);
}

template <int P>
int lds_for(const cdl_stripg_plan &pl)
{
    return gcarve<P>(pl.MT, pl.KS, pl.KQ, pl.G).total;
}

}  // namespace

bool cdl_stripg_plan_for(const cdl_geom *g, cdl_stripg_plan *pl)
{
    if (!cdl_geom_ok(g)) return false;
    if (g->sd != 1 || g->sh != 1 || g->sw != 1) return false;
    if (g->Ph != g->Pw || (g->Ph != 3 && g->Ph != 5 && g->Ph != 7)) return false;
    if ((g->Pd & 1) == 0 || g->pd != g->Pd / 2 || g->ph != g->Ph / 2 || g->pw != g->Pw / 2) return false;
    if (g->M > 64 || g->M < 8 || (g->M & 7)) return false;             // whole register quads of channels (see k_stripg)
    pl->P = g->Ph;
    pl->G = g->C * g->Pd;
    if (pl->G != 1 && pl->G != 3 && pl->G != 5 && pl->G != 7) return false;
    if (g->Ph == 7 && pl->G == 7) return false;                          // 49 ring registers
    pl->MT = (g->M + 31) / 32;
    pl->KS = (pl->G * g->Ph * g->Pw + 15) / 16;
    pl->KQ = (g->M + 15) / 16;
    pl->nsx = (g->W + 31) / 32;
    // segment length from the SAMPLE's geometry only (never the batch size: cdl_strip.hip); 16 rows when that still gives
    // a sample 256 items, else 8 (ring tails of P - 1 rows must end inside the next segment: >= 8 for P = 7)
    const size_t ips16 = (size_t)g->D * pl->nsx * ((g->H + 15) / 16);
    int seg = ips16 >= 256 ? 16 : 8;
    if (g->Ph <= 5 && g->H < 32) seg = 4;
    pl->SEG = seg;
    pl->nsy = (g->H + seg - 1) / seg;
    pl->items = (size_t)g->N * g->D * pl->nsy * pl->nsx;
    pl->prows = seg + g->Ph - 1;
    pl->pxw = 32 + g->Ph - 1;
    pl->patch_floats = pl->items * pl->G * pl->prows * pl->pxw;
    const size_t vox = (size_t)g->D * g->H * g->W;
    if ((size_t)g->M * vox * 4 >= ((size_t)1 << 31) || (size_t)g->M * g->D * g->H * pl->nsx * 128 >= ((size_t)1 << 31)) return false;
    if (pl->items >= ((size_t)1 << 30) || g->H > 65535 || (size_t)g->N * g->C * g->D > 65535) return false;
    const int lds = pl->P == 3 ? lds_for<3>(*pl) : pl->P == 5 ? lds_for<5>(*pl) : lds_for<7>(*pl);
    return lds <= 160 * 1024;
}

int cdl_stripg_stage(const cdl_geom *g, const cdl_stripg_plan &pl, int mode, const float *r, const float *zin,
                     const float *tau, const void *frags, float sgn, float *zout, float *patches, unsigned *map,
                     float *dtau_partial, int do_synth, int rev, int lay_in, int lay_out, hipStream_t st)
{
    if (sgn != 1.0f && sgn != -1.0f) return CDL_EINVAL;
    GSParams p = {};
    p.r = r; p.zin = zin; p.map = map; p.zout = zout; p.tau = tau; p.dtau = dtau_partial;
    p.frags = reinterpret_cast<const uint4 *>(frags);
    p.patches = patches; p.sgn = sgn; p.do_synth = do_synth;
    p.N = g->N; p.C = g->C; p.M = g->M; p.D = g->D; p.H = g->H; p.W = g->W; p.Pd = g->Pd; p.KQ = pl.KQ;
    p.nsx = pl.nsx; p.nsy = pl.nsy; p.SEG = pl.SEG; p.prows = pl.prows; p.rev = rev; p.items = (int)pl.items;
    p.lay_in = lay_in; p.lay_out = lay_out;
    CDL_DBG_FIELD(p.dbg = cdl_opts().fused_debug;)
    if (pl.P == 3) return launch_p<3>(p, pl, mode, st);
    if (pl.P == 5) return launch_p<5>(p, pl, mode, st);
    return launch_p<7>(p, pl, mode, st);
}

int cdl_stripg_assemble(const cdl_geom *g, const cdl_stripg_plan &pl, const float *patches, const float *mask,
                        const float *sub, float alpha, float *out, hipStream_t st)
{
    dim3 grid((unsigned)((g->W + 255) / 256), (unsigned)g->H, (unsigned)(g->N * g->C * g->D));
#define CDL_ASM_G(P_) k_assemble_sg<P_><<<grid, 256, 0, st>>>(patches, mask, sub, alpha, out, g->N, g->C, g->D, g->H, g->W, g->Pd, pl.nsx, pl.nsy, pl.SEG, pl.prows)
    if (pl.P == 3) CDL_ASM_G(3); else if (pl.P == 5) CDL_ASM_G(5); else CDL_ASM_G(7);
#undef CDL_ASM_G
    CDL_LAUNCH_CHECK();
    return 0;
}

size_t cdl_stripg_rsc_floats(const cdl_geom *g, const cdl_stripg_plan &pl)
{
    return (size_t)g->N * g->M * g->D * g->H * pl.nsx * 32;
}
