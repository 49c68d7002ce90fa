// Fused ISTA iteration for one-channel 2-D nets with stride 1 or 2 and up to 192 subbands -- the shipped
// CDLNet-s2030 architecture (K=30 M=169 P=7 s=2; /root/reference/trained_nets/CDLNet-s2030/args.json:2-9).
// Reference: the loop body model/net.py:87,  z = ST(z - A_k(mask * B_k z - yp), tau_k), A_k = Conv2d(1, M, P, stride=s),
// B_k = ConvTranspose2d(M, 1, P, stride=s, output_padding=s-1) (net.py:32-33).
//
// Same fusion boundary as cdl_fused2d.hip / cdl_fusedg.hip:
//
//     launch k :  r_k (thin, full resolution) , z_k (fat, 1/s^2 resolution)  ->  z_{k+1} (fat) , patches of B_{k+1} z_{k+1} (thin)
//
// so the fat tensors cross HBM once in and once out per iteration (2 fat passes instead of the 3 of the analysis /
// synthesis / assemble launches this replaces).  What is new here is the WORK DECOMPOSITION: there is no tile that a
// workgroup's waves share.  A wave is an autonomous worker that walks DOWN a strip of 32 code columns, SEG code rows
// per work item:
//   * the only workgroup-wide state is read-only (weight fragments, tap table): ONE barrier per launch, none per tile;
//     the 8 waves of a workgroup drift apart, so one wave's matrix work overlaps its SIMD partner's loads, LDS traffic
//     and vector work instead of all waves meeting at staging / combine phases;
//   * the thin input lives in a per-wave circular row buffer in LDS (P rows live, s new rows per code row, each image
//     row loaded once per strip, converted once to (bf16 hi | bf16 lo) dwords);
//   * the im2col operand of a code row is gathered ONCE and reused by all M/32 channel tiles (channel-outer loop: 16
//     accumulator registers live per tile, the synthesis-like accumulators -- 32 slots x 32 pixels per tap tile -- are
//     summed over the channel tiles);
//   * col2im: for stride s the taps of a filter row split into s parity classes (kj = s j' + e lands on output column
//     s (x + j') + e - P/2), each a stride-1 Horner chain of DPP wave shifts over j'; the same in the row direction with
//     a register ring per (row parity, column parity).  A finished half-resolution output row (32 + ceil(P/s) - 1 lanes)
//     goes straight from registers to the work item's patch in global memory -- no patch in LDS, no combine phase;
//   * k_assemble_s interleaves the parity planes, sums the <= 2 x 2 overlapping patches in a fixed order and applies
//     alpha, mask and -yp.  Deterministic, no atomics.
// fp32-grade accuracy on the bf16 matrix cores by the hi/lo split of cdl_fused2d.hip (3 MFMAs per product).
#include <type_traits>

#include "cdl_strip.h"

namespace {

#include "cdl_strip_dev.h"

struct SParams {
    const float *r;          // (N,1,H,W)
    const float *zin;        // (N,M,Hz,Wz) or nullptr
    unsigned *map;           // (N, 4*MTP, Hz, Wz) words: plane (2*(R>>1) + h)*2 + {support, sign}, bit 16*(R&1) + v
    float *zout;
    const float *tau;        // (N,M)
    float *dtau;             // (items, M)
    const uint4 *frags;
    float *patches;          // (items, S*S, PROWS, PXW)
    float sgn;
    int do_synth;
    int N, M, H, W, Hz, Wz, MT, KQ, nsx, nsy, SEG, prows, rev;
    int items;
    int lay_in, lay_out;     // 0: (N,M,Hz,Wz); 1: row-strip channel-major [n][yz][strip][M][32 px] (CDL_LAY_RSC)
    CDL_DBG_FIELD(int dbg;)  // probe build only (CDL_FUSED_DEBUG; results are wrong): 1 no fat loads, 2 no fat stores,
                             // 4 no analysis-like MFMAs, 8 no synthesis-like MFMAs, 16 no col2im, 32 no gather
};

// ---- weight preparation (fragment order of cdl_fusedg.hip with one group) ---------------------------------------
//   [A hi | A lo] : (R, ks)       lane (row ch = 32R + (lane&31), h): taps k = 16ks + 8h + e of wA[ch] (k = ki*P + kj)
//   [B hi | B lo] : (Rt, q=2R+s)  lane (row t = 32Rt + (lane&31), h), element e: channel 32R + 16s + 8(e>>2) + 4h + (e&3)
//                                 of wB[.][tap of slot t]
template <int P>
__device__ __forceinline__ void prep_one(const float *__restrict__ wA, const float *__restrict__ wB,
                                         uint4 *__restrict__ out, int M, int MT, int KQ, int t)
{
    constexpr int RT = Shape<P>::RT, KS = Shape<P>::KS, K = P * P;
    const int FA = MT * KS, FB = RT * KQ;
    if (t >= (FA + FB) * 64) return;
    const int lane = t & 63, f = t >> 6;
    const int row = lane & 31, h = lane >> 5;
    float v[8];
    if (f < FA) {
        const int R = f / KS, ks = f % KS;
        const int ch = 32 * R + row;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * ks + 8 * h + e;
            v[e] = (ch < M && k < K) ? wA[(size_t)ch * K + k] : 0.0f;
        }
    } else {
        const int q = f - FA;
        const int kq = q % KQ, Rt = q / KQ;
        const int slot = 32 * Rt + row;
        int tap = -1;
        for (int i = 0; i < P; ++i)
            for (int j = 0; j < P; ++j)
                if (tap_slot<P>(i, j) == slot) tap = i * P + j;
        const int R = kq >> 1, s = kq & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = 32 * R + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3);
            v[e] = (tap >= 0 && ch < M) ? wB[(size_t)ch * K + tap] : 0.0f;
        }
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hh = (__bf16)v[e];
        hi[e] = hh;
        lo[e] = (__bf16)(v[e] - (float)hh);
    }
    uint4 *hd, *ld;
    if (f < FA) { hd = out + (size_t)f * 64; ld = out + (size_t)(FA + f) * 64; }
    else { hd = out + (size_t)(2 * FA + (f - FA)) * 64; ld = out + (size_t)(2 * FA + FB + (f - FA)) * 64; }
    hd[lane] = __builtin_bit_cast(uint4, hi);
    ld[lane] = __builtin_bit_cast(uint4, lo);
}

constexpr int PREP_BATCH = 32;
struct PrepBatch {
    const float *wA[PREP_BATCH];
    const float *wB[PREP_BATCH];
};
template <int P>
__global__ void k_prep_s(PrepBatch b, uint4 *__restrict__ out, int frag_uint4, int M, int MT, int KQ)
{
    prep_one<P>(b.wA[blockIdx.y], b.wB[blockIdx.y], out + (size_t)blockIdx.y * frag_uint4, M, MT, KQ,
                blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- LDS carve: [A frags][B frags][tap table][per wave: thin rows | thresholds] -----------------------------------
struct Carve {
    int wb, koff, wave0, wave_bytes, tau_off, total;
};
template <int P, int S>
__host__ __device__ inline Carve carve(int MT, int KQ)
{
    constexpr int RT = Shape<P>::RT, KS = Shape<P>::KS;
    Carve c;
    c.wb = MT * KS * 2 * 1024;
    c.koff = c.wb + RT * KQ * 2 * 1024;
    c.wave0 = c.koff + KS * 16 * 4;
    c.tau_off = Shape<P>::RBUF * Strip<P, S>::XWP * 4;
    c.wave_bytes = c.tau_off + 2 * MAXP * 32 * 4;
    c.total = c.wave0 + NWV * c.wave_bytes;
    return c;
}

// ---- the stage kernel -----------------------------------------------------------------------------------------------
// MAPPED: the forward modes also write the support / sign map (training); always read in the reverse mode
template <int P, int S, int MTP, int MODE, bool MAPPED>
__global__ __launch_bounds__(NTS) void k_strip(SParams p)
{
    using SH = Shape<P>;
    using ST_ = Strip<P, S>;
    constexpr int RT = SH::RT, KS = SH::KS, RC = SH::RC, HALO = P / 2;
    constexpr int XW = ST_::XW, XWP = ST_::XWP, PXW = ST_::PXW, NLD = ST_::NLD, NPRE = ST_::NPRE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = p.M, MT = p.MT, KQ = p.KQ;
    const Carve cv = carve<P, S>(MT, KQ);
    const uint4 *wa = reinterpret_cast<const uint4 *>(smem);
    const uint4 *wb = reinterpret_cast<const uint4 *>(smem + cv.wb);
    int *koff = reinterpret_cast<int *>(smem + cv.koff);

    const int tid = threadIdx.x, lane = tid & 63;
    // (the wave index is wave-uniform, but only readfirstlane makes that provable: everything a wave derives from it --
    //  its work item, sample, buffer descriptors, LDS base -- must live in scalar registers, or every buffer access
    //  gets a waterfall loop around its descriptor)
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int FA = MT * KS, FB = RT * KQ;
    unsigned char *wbase = smem + cv.wave0 + wid * cv.wave_bytes;       // this wave's private LDS
    unsigned *thin = reinterpret_cast<unsigned *>(wbase);               // [RBUF][XWP] dwords: bf16 hi (low half) | bf16 lo
    float *tau_s = reinterpret_cast<float *>(wbase + cv.tau_off);       // [2*MTP*32]

    // ---- once per workgroup: weight fragments and the tap -> byte offset table; the ONLY barrier of the launch
    const bool negA = MODE != MODE_BWD && p.sgn < 0.0f;                 // uniform
    {
        uint4 *wdst = reinterpret_cast<uint4 *>(smem);
        const int nfr = 2 * (FA + FB) * 64;
        for (int i = tid; i < nfr; i += 4 * NTS) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p.frags[min(i + u * NTS, nfr - 1)];
            // A forward iteration computes u = z - A r: the LDS copy of the analysis-like fragments (hi and lo: the first
            // 2 FA entries) carries that sign (bf16 sign bits flipped: exact), so the accumulator, started at z, IS u
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * NTS < nfr) {
                    if (negA && i + u * NTS < 2 * FA * 64) {
                        v[u].x ^= 0x80008000u; v[u].y ^= 0x80008000u; v[u].z ^= 0x80008000u; v[u].w ^= 0x80008000u;
                    }
                    wdst[i + u * NTS] = v[u];
                }
        }
        for (int k = tid; k < KS * 16; k += NTS) koff[k] = k < P * P ? ((k / P) * XWP + (k % P)) * 4 : 0;
    }
    __syncthreads();
    auto afrag = [&](int f) { return __builtin_bit_cast(bf16x8, wa[f * 64 + lane]); };
    auto bfrag = [&](int f) { return __builtin_bit_cast(bf16x8, wb[f * 64 + lane]); };

    const int plane = p.Hz * p.Wz;
    const int plane4 = plane * 4;
    const int up_addr = ((lane & 31) + 32) * 4;
    const int nwaves = gridDim.x * NWV;
    const bool has_base = (MODE == MODE_FWD) || (MODE == MODE_BWD && p.zin != nullptr);
    const int thin_base = (int)(wbase - smem);                          // byte address of the wave's thin rows in LDS

    // per-lane element k of a thin row group: flat index lane + 64 k -> (row within the group, column); recomputed where
    // needed (held in registers across the row loop they cost 3 per element)
    auto tl_in = [&](int k) { return lane + 64 * k < S * XW; };
    auto tl_rr = [&](int k) { return (S > 1 && lane + 64 * k >= XW) ? 1 : 0; };
    auto tl_col = [&](int k) { return lane + 64 * k - tl_rr(k) * XW; };

#pragma unroll 1
    for (int it = blockIdx.x * NWV + wid; it < p.items; it += nwaves) {
        int bid = p.rev ? p.items - 1 - it : it;
        const int item = bid;
        const int sx = bid % p.nsx; bid /= p.nsx;
        const int sy = bid % p.nsy;
        const int n = bid / p.nsy;
        const int xz0 = sx * 32, yz0 = sy * p.SEG;
        const int x = xz0 + c;
        const bool xok = x < p.Wz;
        const int nblk = min(p.SEG, p.Hz - yz0);                        // code rows of this item (uniform)

        if (MODE != MODE_BWD) {
#pragma unroll
            for (int k = 0; k < MTP; ++k) {
                const int ch = lane + 64 * k;
                tau_s[ch] = ch < M ? p.tau[(size_t)n * M + ch] : 0.0f;
            }
        }
        const float *rimg = p.r + (size_t)n * p.H * p.W;
        // thin row group of code row bb: the LAST S image rows of its window, rows S*(yz0+bb) - HALO + P - S + [0, S)
        float tn[NLD];
        auto thin_issue = [&](int bb, float (&tn)[NLD]) __attribute__((always_inline)) {
            const int row0 = S * (yz0 + bb) - HALO + P - S;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int yy = row0 + tl_rr(k), xx = S * xz0 - HALO + tl_col(k);
                const bool ok = tl_in(k) && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                const float v = rimg[(size_t)min(max(yy, 0), p.H - 1) * p.W + min(max(xx, 0), p.W - 1)];
                tn[k] = ok ? v : 0.0f;
            }
        };
        auto thin_commit = [&](int bb, const float (&tn)[NLD]) __attribute__((always_inline)) {
            const int row0 = S * (yz0 + bb) - HALO + P - S;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const float v = tn[k];
                const __bf16 hh = (__bf16)v;
                const __bf16 ll = (__bf16)(v - (float)hh);
                const unsigned w = (unsigned)__builtin_bit_cast(unsigned short, hh) |
                                   ((unsigned)__builtin_bit_cast(unsigned short, ll) << 16);
                const int slot = (row0 + tl_rr(k)) & (RC - 1);
                if (tl_in(k)) {
                    thin[slot * XWP + tl_col(k)] = w;
                    if (slot < P - 1) thin[(slot + RC) * XWP + tl_col(k)] = w;
                }
            }
        };
        // window of code row 0: groups -NPRE .. 0, all loads in flight before the first conversion
        {
            float tp[NPRE + 1][NLD];
#pragma unroll
            for (int gI = 0; gI <= NPRE; ++gI) thin_issue(gI - NPRE, tp[gI]);
            if (nblk > 1) thin_issue(1, tn);
#pragma unroll
            for (int gI = 0; gI <= NPRE; ++gI) thin_commit(gI - NPRE, tp[gI]);
        }

        // fat addressing.  A (channel, code row, code column) element of sample n sits at
        //   layout 0 (the reference's):  (ch * plane + yz * Wz + x) * 4
        //   layout 1 (row-strip channel-major, inside a sweep):  ((yz * nsx + sx) * M + ch) * 128 + c * 4  -- the 32 columns
        //     of a strip row are 128 contiguous bytes per channel and the channels follow each other, so everything a
        //     wave moves for one code row (M * 128 B) is ONE contiguous run (whole DRAM pages, whole cache lines)
        // per lane: channel half 4h and column; per element a scalar stride (estr); per step a scalar base.
        const int bytes_in = p.lay_in ? M * p.Hz * p.nsx * 128 : M * plane4;
        const int bytes_out = p.lay_out ? M * p.Hz * p.nsx * 128 : M * plane4;
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
            has_base ? const_cast<float *>(p.zin) + (size_t)n * (bytes_in / 4) : p.zout, 0, has_base ? bytes_in : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
            p.zout + (size_t)n * (bytes_out / 4), 0, bytes_out, 0x00020000);
        const int estr_in = p.lay_in ? 128 : plane4, estr_out = p.lay_out ? 128 : plane4;
        const int vlane_in = xok ? (p.lay_in ? (4 * h * 32 + c) * 4 : (4 * h * plane + x) * 4) : OOB;
        const int vlane_out = xok ? (p.lay_out ? (4 * h * 32 + c) * 4 : (4 * h * plane + x) * 4) : OOB;
        auto sbase_in = [&](int yz, int R) {
            return __builtin_amdgcn_readfirstlane(p.lay_in ? ((yz * p.nsx + sx) * M + 32 * R) * 128 : (yz * p.Wz + 32 * R * plane) * 4);
        };
        auto sbase_out = [&](int yz, int R) {
            return __builtin_amdgcn_readfirstlane(p.lay_out ? ((yz * p.nsx + sx) * M + 32 * R) * 128 : (yz * p.Wz + 32 * R * plane) * 4);
        };
        unsigned *const map_n = p.map ? p.map + (size_t)n * (4 * MTP) * plane : nullptr;

        // The fat input of a channel tile is loaded INTO the accumulator registers of that tile: the matrix cores then
        // accumulate (sgn A) r on top of it (C-in operand), so no separate copy of z_k is held and the epilogue has no
        // add.  THREE sets rotate over the (code row, channel tile) steps of the item: while step s is worked on, the loads
        // of steps s+1 and s+2 are in flight (16 dword loads each) -- with one tile in flight per wave the CU had ~32 KB
        // outstanding and the launch streamed at 3.8 TB/s even with the arithmetic switched off (tools/probe_strip.py).
        f32x16 accA, accB, accC;
        unsigned short supA = 0, supB = 0, supC = 0, sgnA = 0, sgnB = 0, sgnC = 0;         // reverse: the step's 16-bit halves of the map words
        // the register quad that straddles M (if any) addresses its missing channels out of range
        const int cb_part = (M & 7) ? (M & ~7) : -8;
        int vo_part_in[4], vo_part_out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            vo_part_in[e] = (cb_part + 4 * h + e < M) ? vlane_in : OOB;
            vo_part_out[e] = (cb_part + 4 * h + e < M) ? vlane_out : OOB;
        }
        // element v of tile R = channel 32R + 8(v>>2) + 4h + (v&3); register quads beyond M stay zero (uniform skip)
        auto fat_issue = [&](int yz, int R, f32x16 &dst, unsigned short &sup, unsigned short &sgb) __attribute__((always_inline)) {
#pragma unroll
            for (int v = 0; v < 16; ++v) dst[v] = 0.0f;
            if (MODE == MODE_BWD) {
                sup = 0; sgb = 0;
                if (xok) {
                    const unsigned short *mp = reinterpret_cast<const unsigned short *>(
                        map_n + (size_t)((2 * (R >> 1) + h) * 2) * plane + (size_t)yz * p.Wz + x) + (R & 1);
                    sup = mp[0];
                    sgb = mp[2 * plane];
                }
            }
            if (MODE == MODE_FIRST || !has_base || CDL_DBG(p.dbg, 1)) return;
            const int so0 = sbase_in(yz, R);
#pragma unroll
            for (int qv = 0; qv < 4; ++qv) {
                const int cb = 32 * R + 8 * qv;
                if (cb >= M) continue;                                    // uniform
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dst[4 * qv + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rs_in, cb == cb_part ? vo_part_in[e] : vlane_in, so0 + (8 * qv + e) * estr_in, 0));
            }
        };

        float ring[S][S][ST_::JM];                                       // [row parity][column parity][slot]
#pragma unroll
        for (int f = 0; f < S; ++f)
#pragma unroll
            for (int e = 0; e < S; ++e)
#pragma unroll
                for (int i = 0; i < ST_::JM; ++i) ring[f][e][i] = 0.0f;
        float tacc[2 * MTP];
#pragma unroll
        for (int i = 0; i < 2 * MTP; ++i) tacc[i] = 0.0f;
        float *patch = p.patches + (size_t)item * (S * S) * p.prows * PXW;
        const bool tau_neg = MODE != MODE_BWD &&
                             __ballot(!(tau_s[lane] >= 0.0f) || (MTP > 1 && !(tau_s[lane + 64] >= 0.0f)) ||
                                      (MTP > 2 && !(tau_s[lane + 128] >= 0.0f))) != 0ull;

        bf16x8 bh[KS], bl[KS];                                           // im2col operand of the current code row
        f32x16 D[RT];                                                    // synthesis-like accumulators of the current code row
        const int nsteps = nblk * MT;
        // (b, R) of the step being worked on and of the step whose loads are issued (two ahead)
        int b = 0, R = 0, pb = 0, pR = 0;
        auto advance = [&](int &bb, int &RR) { if (++RR == MT) { RR = 0; ++bb; } };
        fat_issue(yz0, 0, accA, supA, sgnA);
        advance(pb, pR);
        if (nsteps > 1) fat_issue(yz0 + pb, pR, accB, supB, sgnB);
        advance(pb, pR);

        // ---- one step: channel tile R of code row b, accumulators in `acc`
        auto step = [&](f32x16 &acc, unsigned short sup, unsigned short sgb) __attribute__((always_inline)) {
            const int yz = yz0 + b;
            if (R == 0) {
                // ---- im2col operand of this code row, gathered once for all channel tiles
                const int s0 = (S * yz - HALO) & (RC - 1);
                const int pixbase = thin_base + (s0 * XWP + S * c) * 4;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (CDL_DBG(p.dbg, 32)) { const u32x4 c4 = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
                                              bh[ks] = __builtin_bit_cast(bf16x8, c4); bl[ks] = bh[ks]; continue; }
                    const int4 o0 = *reinterpret_cast<const int4 *>(koff + 16 * ks + 8 * h);
                    const int4 o1 = *reinterpret_cast<const int4 *>(koff + 16 * ks + 8 * h + 4);
                    const int oo[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
                    unsigned w[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) w[i] = *reinterpret_cast<const unsigned *>(smem + (oo[i] + pixbase));
                    u32x4 hv, lv;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        hv[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x05040100u);     // low halves
                        lv[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x07060302u);     // high halves
                    }
                    bh[ks] = __builtin_bit_cast(bf16x8, hv);
                    bl[ks] = __builtin_bit_cast(bf16x8, lv);
                }
                // code columns beyond the plane (last strip only: uniform branch) get a zero operand: their accumulators
                // and (the fat loads being out of range) their codes stay exactly zero, so nothing downstream needs a select
                if (__ballot(!xok) != 0ull) {
                    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        if (!xok) { bh[ks] = __builtin_bit_cast(bf16x8, zero4); bl[ks] = __builtin_bit_cast(bf16x8, zero4); }
                }
                // ---- thin rows: commit the group of the next code row (loaded one row ago), fetch the one after
                if (b + 1 < nblk) thin_commit(b + 1, tn);
                if (b + 2 < nblk) thin_issue(b + 2, tn);
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) D[t][v] = 0.0f;
            }
            // ---- analysis-like GEMM on top of the loaded fat input
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (CDL_DBG(p.dbg, 4)) continue;
                const bf16x8 ah = afrag(R * KS + ks), al = afrag(FA + R * KS + ks);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ks], acc, 0, 0, 0);
            }
            // ---- epilogue
            const int so0 = sbase_out(yz, R);
            float tsum[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) tsum[v] = 0.0f;
            unsigned ws = 0, wg = 0;
            const bool nostore = CDL_DBG(p.dbg, 2);
            if (tau_neg)                                                 // wave-uniform
                strip_epilogue<MODE, MAPPED, true>(acc, tsum, ws, wg, tau_s, R, M, h, cb_part, vo_part_out, vlane_out, rs_out, so0,
                                                   estr_out, sup, sgb, nostore);
            else
                strip_epilogue<MODE, MAPPED, false>(acc, tsum, ws, wg, tau_s, R, M, h, cb_part, vo_part_out, vlane_out, rs_out, so0,
                                                    estr_out, sup, sgb, nostore);
            if (MODE != MODE_BWD && MAPPED && xok) {
                // this tile's 16-bit half of the pair's map words (a sign bit only where there is support)
                unsigned *mw = map_n + (size_t)((2 * (R >> 1) + h) * 2) * plane + (size_t)yz * p.Wz + x;
                if (R == MT - 1 && (R & 1) == 0) {                       // uniform: an odd tile count leaves the last words'
                    mw[0] = ws;                                          // upper halves without a tile: written as zero
                    mw[plane] = wg & ws;
                } else {
                    unsigned short *mp = reinterpret_cast<unsigned short *>(mw) + (R & 1);
                    mp[0] = (unsigned short)ws;
                    mp[2 * plane] = (unsigned short)(wg & ws);
                }
            }
            if (MODE == MODE_BWD) {
                const float ts = LaneTransposeSum<16>::run(tsum, c);
#pragma unroll
                for (int k = 0; k < 2 * MTP; ++k) tacc[k] += k == R ? ts : 0.0f;     // (branch-free: a branch made tacc a scratch array)
            }
            if (MODE == MODE_BWD && !p.do_synth) return;
            // ---- this tile's share of the synthesis-like GEMM: the accumulator tile is the B operand (k = channel) as it
            //      stands; hi / lo split two elements at a time (v_cvt_pk_bf16_f32)
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) {
                const int q = 2 * R + sI;
                if (q >= KQ) continue;                                   // uniform: padding channels only
                u32x4 zhw, zlw;
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const float v0 = acc[8 * sI + 2 * e2], v1 = acc[8 * sI + 2 * e2 + 1];
                    const bf16x2 hh = __builtin_convertvector(f32x2{v0, v1}, bf16x2);
                    const unsigned hb = __builtin_bit_cast(unsigned, hh);
                    const float r0 = v0 - __builtin_bit_cast(float, hb << 16);
                    const float r1 = v1 - __builtin_bit_cast(float, hb & 0xffff0000u);
                    const bf16x2 ll = __builtin_convertvector(f32x2{r0, r1}, bf16x2);
                    zhw[e2] = hb;
                    zlw[e2] = __builtin_bit_cast(unsigned, ll);
                }
                const bf16x8 zh = __builtin_bit_cast(bf16x8, zhw), zl = __builtin_bit_cast(bf16x8, zlw);
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    if (CDL_DBG(p.dbg, 8)) { D[t][0] += acc[8 * sI]; continue; }
                    const bf16x8 wh = bfrag(t * KQ + q), wl = bfrag(FB + t * KQ + q);
                    D[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, zh, D[t], 0, 0, 0);
                    D[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, zl, D[t], 0, 0, 0);
                    D[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, zh, D[t], 0, 0, 0);
                }
            }
            if (R != MT - 1) return;
            if (CDL_DBG(p.dbg, 16)) { ring[0][0][0] += D[0][0] + D[RT - 1][15]; return; }
            // ---- last tile of the code row: col2im.  Column direction: for filter row i and column parity e, Horner chain
            // over j' (kj = S j' + e): lane L ends with the contribution to half-resolution column L of the strip.  Row
            // direction: filter row i = S i' + f of code row yz lands on half-resolution row yz + i' of parity plane f:
            // ring slot i'.
#pragma unroll
            for (int i = 0; i < P; ++i) {
                float sr[S];
#pragma unroll
                for (int e = 0; e < S; ++e) sr[e] = 0.0f;
#pragma unroll
                for (int jj = ST_::JM - 1; jj >= 0; --jj)
#pragma unroll
                    for (int e = 0; e < S; ++e) {
                        const int kj = S * jj + e;
                        if (kj >= P) continue;
                        const int slot = tap_slot<P>(i, kj);
                        const int v = 4 * ((slot >> 3) & 3) + (slot & 3), hh = (slot >> 2) & 1;
                        float val = D[slot >> 5][v];
                        if (hh) val = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(up_addr, __builtin_bit_cast(int, val)));
                        sr[e] = wave_shr1(sr[e]) + (h == 0 ? val : 0.0f);
                    }
#pragma unroll
                for (int e = 0; e < S; ++e) ring[i % S][e][i / S] += sr[e];
            }
            // slot 0 of every ring is complete: half-resolution row b of the item's patch, straight to global memory
#pragma unroll
            for (int f = 0; f < S; ++f)
#pragma unroll
                for (int e = 0; e < S; ++e) {
                    if (lane < PXW) patch[((size_t)(f * S + e) * p.prows + b) * PXW + lane] = ring[f][e][0];
#pragma unroll
                    for (int i = 0; i + 1 < ST_::JM; ++i) ring[f][e][i] = ring[f][e][i + 1];
                    ring[f][e][ST_::JM - 1] = 0.0f;
                }
        };

#pragma unroll 1
        for (int s0 = 0; s0 < nsteps; s0 += 3) {
            // three steps with their own register sets (named, not an indexed array: the array went to scratch)
            auto sub = [&](int u, f32x16 &cur, unsigned short &csup, unsigned short &csgn, f32x16 &nxt, unsigned short &nsup,
                           unsigned short &nsgn) __attribute__((always_inline)) {
                if (s0 + u >= nsteps) return;                            // uniform
                // the set of step s-1 is free (its codes are stored, its bf16 split sits in the matrix pipe's operands):
                // it takes the loads of step s+2
                if (s0 + u + 2 < nsteps) fat_issue(yz0 + pb, pR, nxt, nsup, nsgn);
                advance(pb, pR);
                step(cur, csup, csgn);
                advance(b, R);
                __builtin_amdgcn_sched_barrier(0);
            };
            sub(0, accA, supA, sgnA, accC, supC, sgnC);
            sub(1, accB, supB, sgnB, accA, supA, sgnA);
            sub(2, accC, supC, sgnC, accB, supB, sgnB);
        }
        // ---- the rows below the item's last code row are still in the rings
        if (MODE != MODE_BWD || p.do_synth) {
#pragma unroll
            for (int f = 0; f < S; ++f)
#pragma unroll
                for (int e = 0; e < S; ++e)
#pragma unroll
                    for (int i = 0; i + 1 < ST_::JM; ++i)
                        if (lane < PXW) patch[((size_t)(f * S + e) * p.prows + nblk + i) * PXW + lane] = ring[f][e][i];
        }
        if (MODE == MODE_BWD) {
#pragma unroll
            for (int R = 0; R < 2 * MTP; ++R) {
                if (R >= MT) continue;
                const float tot = tacc[R] + __shfl_xor(tacc[R], 16, 64);
                const int ch = 32 * R + 8 * (c >> 2) + 4 * h + (c & 3);
                if (c < 16 && ch < M) p.dtau[(size_t)item * M + ch] = tot;
            }
        }
    }
}

// out[n,0,Y,X] = (mask ? mask : 1) * alpha * (sum of the patches covering (Y, X)) - (sub ? sub : 0).
// (Y + HALO) = S * nrow + f, (X + HALO) = S * m + e: parity plane (f, e), half-resolution position (nrow, m); covered by
// the segment / strip that contains it and, near their upper / left border, by the one before.  Fixed order.
// V pixels per thread (V = 4 when W % 4 == 0: 16-byte thin accesses; every pixel still sums its own <= 4 patch values in
// the same order, so both forms are bit-identical).
template <int P, int S, int V>
__global__ __launch_bounds__(256) void k_assemble_s(const float *__restrict__ patches, const float *__restrict__ mask,
                                                    const float *__restrict__ sub, float alpha, float *__restrict__ out,
                                                    int N, int H, int W, int nsx, int nsy, int SEG, int prows)
{
    constexpr int HALO = P / 2, PXW = Strip<P, S>::PXW;
    const int X0 = (blockIdx.x * 256 + threadIdx.x) * V, Y = blockIdx.y, n = blockIdx.z;
    if (X0 >= W) return;
    const int f = (Y + HALO) % S, nrow = (Y + HALO) / S;
    const int If = (P - f + S - 1) / S;                                  // taps in this row-parity class
    const int sy_hi = min(nsy - 1, nrow / SEG);
    const bool y_lo = sy_hi > 0 && nrow - (sy_hi - 1) * SEG < SEG + If - 1;
    const size_t pplane = (size_t)prows * PXW;
    float v[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int X = X0 + k;
        const int e = (X + HALO) % S, m = (X + HALO) / S;
        const int Je = (P - e + S - 1) / S;
        const int sx_hi = min(nsx - 1, m / 32);
        const bool x_lo = sx_hi > 0 && m - (sx_hi - 1) * 32 < 32 + Je - 1;
        auto at = [&](int sy, int sx) {
            const size_t item = ((size_t)n * nsy + sy) * nsx + sx;
            return patches[(item * (S * S) + (size_t)(f * S + e)) * pplane + (size_t)(nrow - sy * SEG) * PXW + (m - sx * 32)];
        };
        float sum = 0.0f;
        if (y_lo) {
            if (x_lo) sum += at(sy_hi - 1, sx_hi - 1);
            sum += at(sy_hi - 1, sx_hi);
        }
        if (x_lo) sum += at(sy_hi, sx_hi - 1);
        sum += at(sy_hi, sx_hi);
        v[k] = alpha * sum;
    }
    const size_t i = ((size_t)n * H + Y) * W + X0;
    if (V == 4) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (mask) { const float4 mm = *reinterpret_cast<const float4 *>(mask + i); o.x *= mm.x; o.y *= mm.y; o.z *= mm.z; o.w *= mm.w; }
        if (sub) { const float4 sv = *reinterpret_cast<const float4 *>(sub + i); o.x -= sv.x; o.y -= sv.y; o.z -= sv.z; o.w -= sv.w; }
        *reinterpret_cast<float4 *>(out + i) = o;
    } else {
        float o = v[0];
        if (mask) o *= mask[i];
        if (sub) o -= sub[i];
        out[i] = o;
    }
}

template <int P, int S, int MTP, int MODE, bool MAPPED>
int launch_one(const SParams &p, const cdl_strip_plan &pl, hipStream_t st)
{
    const int lds = carve<P, S>(pl.MT, pl.KQ).total;
    if (int rc = cdl_ensure_dynamic_lds((const void *)k_strip<P, S, MTP, MODE, MAPPED>, lds)) return rc;
    size_t cus = (size_t)cdl_cu_count();
    const int cap = cdl_opts().fused_grid;
    if (cap > 0 && (size_t)cap < cus) cus = (size_t)cap;
    const size_t wgs = (pl.items + NWV - 1) / NWV;
    const unsigned grid = (unsigned)(wgs < cus ? wgs : cus);
    k_strip<P, S, MTP, MODE, MAPPED><<<grid, NTS, lds, st>>>(p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <int P, int S, int MTP>
int launch_mode(const SParams &p, const cdl_strip_plan &pl, int mode, hipStream_t st)
{
    if (mode == MODE_FWD)
        return p.map ? launch_one<P, S, MTP, MODE_FWD, true>(p, pl, st) : launch_one<P, S, MTP, MODE_FWD, false>(p, pl, st);
    if (mode == MODE_FIRST)
        return p.map ? launch_one<P, S, MTP, MODE_FIRST, true>(p, pl, st) : launch_one<P, S, MTP, MODE_FIRST, false>(p, pl, st);
    return launch_one<P, S, MTP, MODE_BWD, true>(p, pl, st);
}

template <int P, int S>
int launch_mt(const SParams &p, const cdl_strip_plan &pl, int mode, hipStream_t st)
{
    const int mtp = (pl.MT + 1) / 2;
    if (mtp == 1) return launch_mode<P, S, 1>(p, pl, mode, st);
    if (mtp == 2) return launch_mode<P, S, 2>(p, pl, mode, st);
    return launch_mode<P, S, 3>(p, pl, mode, st);
}

template <int P>
int launch_s(const SParams &p, const cdl_strip_plan &pl, int mode, hipStream_t st)
{
    return pl.S == 2 ? launch_mt<P, 2>(p, pl, mode, st) : launch_mt<P, 1>(p, pl, mode, st);
}

template <int P>
int lds_for(const cdl_strip_plan &pl)
{
    return pl.S == 2 ? carve<P, 2>(pl.MT, pl.KQ).total : carve<P, 1>(pl.MT, pl.KQ).total;
}

}  // namespace

bool cdl_strip_plan_for(const cdl_geom *g, cdl_strip_plan *pl)
{
    if (!cdl_geom_ok(g)) return false;
    if (g->C != 1 || g->D != 1 || g->Pd != 1 || g->sd != 1 || g->pd != 0) return false;
    if (g->sh != g->sw || (g->sh != 1 && g->sh != 2)) return false;
    if (g->Ph != g->Pw || (g->Ph != 3 && g->Ph != 5 && g->Ph != 7)) return false;
    if (g->ph != g->Ph / 2 || g->pw != g->Pw / 2) return false;
    if (g->M < 8 || g->M > 64 * MAXP) return false;
    pl->P = g->Ph;
    pl->S = g->sh;
    pl->MT = (g->M + 31) / 32;
    pl->KS = (g->Ph * g->Pw + 15) / 16;
    pl->KQ = (g->M + 15) / 16;
    pl->RT = g->Ph == 7 ? 2 : 1;
    pl->Hz = g->H / pl->S;
    pl->Wz = g->W / pl->S;
    pl->nsx = (pl->Wz + 31) / 32;
    // segment length: a function of the SAMPLE's code plane only -- never of the batch size -- so that a sample's
    // result does not depend on what else is in the batch (the patch split fixes the order of the col2im row sums);
    // 8 segments per plane up to 32 rows each: the 64 x 256 x 256 batch of the benchmark gives every wave of the chip
    // exactly one 16-row item, longer segments re-load fewer halo rows and emit fewer ring tails
    // (at least ceil(P/S) - 1 rows: a segment's ring tail must end inside the NEXT segment -- k_assemble_s sums two
    //  candidates per direction)
    const int seg_min = (pl->P + pl->S - 1) / pl->S - 1 <= 4 ? 4 : 8;
    int seg = pl->Hz >= 256 ? 32 : pl->Hz >= 128 ? 16 : pl->Hz >= 64 ? 8 : 4;
    seg = seg < seg_min ? seg_min : seg;
    pl->SEG = seg;
    pl->nsy = (pl->Hz + seg - 1) / seg;
    pl->items = (size_t)g->N * pl->nsy * pl->nsx;
    const int JM = (pl->P + pl->S - 1) / pl->S;
    pl->prows = seg + JM - 1;
    pl->pxw = 32 + JM - 1;
    pl->frag_uint4 = (size_t)2 * (pl->MT * pl->KS + pl->RT * pl->KQ) * 64;
    pl->patch_floats = pl->items * pl->S * pl->S * pl->prows * pl->pxw;
    pl->map_words = (size_t)g->N * 4 * ((pl->MT + 1) / 2) * pl->Hz * pl->Wz;
    if ((size_t)g->M * pl->Hz * pl->Wz * 4 >= ((size_t)1 << 31)) return false;        // per-sample buffer descriptor range
    if ((size_t)g->H * g->W >= ((size_t)1 << 29) || pl->items >= ((size_t)1 << 30) || g->H > 65535 || g->N > 65535) return false;
    const int lds = pl->P == 3 ? lds_for<3>(*pl) : pl->P == 5 ? lds_for<5>(*pl) : lds_for<7>(*pl);
    return lds <= 160 * 1024;
}

int cdl_strip_prep_pairs(const cdl_geom *g, const cdl_strip_plan &pl, const float *const *w1, const float *const *w2,
                         int K, int shift2, void *frags, hipStream_t st)
{
    const int threads = (int)(pl.frag_uint4 / 2);
    for (int k0 = 0; k0 < K; k0 += PREP_BATCH) {
        const int nb = K - k0 < PREP_BATCH ? K - k0 : PREP_BATCH;
        PrepBatch b = {};
        for (int i = 0; i < nb; ++i) {
            const int k = k0 + i;
            b.wA[i] = shift2 == 0 ? w1[(k + 1) % K] : w1[k];
            b.wB[i] = shift2 == 0 ? w2[k] : w2[(k + 1) % K];
        }
        dim3 grid((unsigned)((threads + 255) / 256), (unsigned)nb);
        uint4 *out = reinterpret_cast<uint4 *>(frags) + (size_t)k0 * pl.frag_uint4;
        if (pl.P == 3) k_prep_s<3><<<grid, 256, 0, st>>>(b, out, (int)pl.frag_uint4, g->M, pl.MT, pl.KQ);
        else if (pl.P == 5) k_prep_s<5><<<grid, 256, 0, st>>>(b, out, (int)pl.frag_uint4, g->M, pl.MT, pl.KQ);
        else k_prep_s<7><<<grid, 256, 0, st>>>(b, out, (int)pl.frag_uint4, g->M, pl.MT, pl.KQ);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return -(int)e;
    }
    return 0;
}

int cdl_strip_stage(const cdl_geom *g, const cdl_strip_plan &pl, int mode, const float *r, const float *zin,
                    const float *tau, const void *frags, float sgn, float *zout, float *patches, unsigned *map,
                    float *dtau_partial, int do_synth, int rev, int lay_in, int lay_out, hipStream_t st)
{
    if (sgn != 1.0f && sgn != -1.0f) return CDL_EINVAL;              // the accumulator carries sgn * zin: |sgn| = 1 only
    SParams p = {};
    p.r = r; p.zin = zin; p.map = map; p.zout = zout; p.tau = tau; p.dtau = dtau_partial;
    p.frags = reinterpret_cast<const uint4 *>(frags);
    p.patches = patches; p.sgn = sgn; p.do_synth = do_synth;
    p.N = g->N; p.M = g->M; p.H = g->H; p.W = g->W; p.Hz = pl.Hz; p.Wz = pl.Wz; p.MT = pl.MT; p.KQ = pl.KQ;
    p.nsx = pl.nsx; p.nsy = pl.nsy; p.SEG = pl.SEG; p.prows = pl.prows; p.rev = rev; p.items = (int)pl.items;
    p.lay_in = lay_in; p.lay_out = lay_out;
    if ((size_t)g->M * pl.Hz * pl.nsx * 128 >= ((size_t)1 << 31)) return CDL_EUNSUPPORTED;
    CDL_DBG_FIELD(p.dbg = cdl_opts().fused_debug;)
    if (pl.P == 3) return launch_s<3>(p, pl, mode, st);
    if (pl.P == 5) return launch_s<5>(p, pl, mode, st);
    return launch_s<7>(p, pl, mode, st);
}

size_t cdl_strip_rsc_floats(const cdl_geom *g, const cdl_strip_plan &pl)
{
    return (size_t)g->N * g->M * pl.Hz * pl.nsx * 32;
}

int cdl_strip_assemble(const cdl_geom *g, const cdl_strip_plan &pl, const float *patches, const float *mask,
                       const float *sub, float alpha, float *out, hipStream_t st)
{
    const bool v4 = (g->W & 3) == 0 && !cdl_opts().scalar_assemble;      // (CDL_SCALAR_ASSEMBLE=1: the scalar form, for tests)
    dim3 grid((unsigned)(((v4 ? g->W / 4 : g->W) + 255) / 256), (unsigned)g->H, (unsigned)g->N);
#define CDL_ASM_S(P_, S_)                                                                                                   \
    do {                                                                                                                    \
        if (v4) k_assemble_s<P_, S_, 4><<<grid, 256, 0, st>>>(patches, mask, sub, alpha, out, g->N, g->H, g->W, pl.nsx, pl.nsy, pl.SEG, pl.prows); \
        else k_assemble_s<P_, S_, 1><<<grid, 256, 0, st>>>(patches, mask, sub, alpha, out, g->N, g->H, g->W, pl.nsx, pl.nsy, pl.SEG, pl.prows);    \
    } while (0)
    if (pl.S == 2) { if (pl.P == 3) CDL_ASM_S(3, 2); else if (pl.P == 5) CDL_ASM_S(5, 2); else CDL_ASM_S(7, 2); }
    else { if (pl.P == 3) CDL_ASM_S(3, 1); else if (pl.P == 5) CDL_ASM_S(5, 1); else CDL_ASM_S(7, 1); }
#undef CDL_ASM_S
    CDL_LAUNCH_CHECK();
    return 0;
}
