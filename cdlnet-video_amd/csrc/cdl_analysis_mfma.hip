// Analysis (correlation with the filter bank) on the matrix cores for the shapes the fused 2-D kernel does not
// take: any C, 2-D / 3-D, stride 1 / 2, odd square planes up to 9 x 9 and the 9 x 5 planes of the 3-D MRI net (reference F.conv2d / F.conv3d at
// model/net.py:85,87,200,205 and their use in the reverse sweep), with cdl_analysis' epilogues:
//     acc  = alpha * sum_k W[m][k] X[k(px)]            k = (c, kd, ki, kj) in the filter's own memory order
//     base = zin ? (gate ? (gate != 0 ? zin : 0) : zin) : 0
//     out  = prox(base + acc) | ST(base + acc, tau[n,m]) | base + acc
// One GEMM D[m][px] = sum_k A[m][k] B[k][px]: A = the filter rows (prepared once per launch as bf16 hi/lo MFMA
// fragments, resident in LDS), B = im2col of the thin image, gathered from bf16 hi/lo planes in LDS through a tap ->
// offset table (lane = pixel column, 8 consecutive taps), split-bf16 x3 products, fp32 accumulation.  A workgroup
// walks TPW consecutive 32 x 8 tiles of output pixels with the same weights; each of its 8 waves owns one pixel row.
//
// The epilogue goes through LDS one 32-channel tile at a time (a first version applied it straight from the
// accumulator registers: 16*MT unrolled fat accesses with 64-bit addresses per lane, 1.25 ms and scratch spills at
// M = 169 against 0.67 ms for the VALU kernel).
#include "cdl_common.h"

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int ALX = 32, ALY = 8;          // output-pixel tile: 8 rows of 32 pixels, one row per wave
constexpr int ANT = 64 * ALY;
constexpr int TPW = 4;                    // tiles (along x) per workgroup (the weight fragments are loaded once for them) in launches that
                                          // fill the chip anyway; fewer when the workgroups would not (Plan::tpw)
constexpr size_t LDS_MAX = 152 * 1024;    // dynamic LDS one workgroup may ask for (160 KB per CU)

// filters (M, K) with K = C*Pd*Ph*Pw -> A fragments frag[(R*KS + ks)*2 + hl][lane]: lane (row m = 32R + (lane&31),
// h = lane>>5) holds taps k = 16ks + 8h + i (zero beyond M or K)
__global__ void k_ana_prep(const float *__restrict__ w, uint4 *__restrict__ frags, int M, int K, int MT, int KS)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= MT * KS * 64) return;
    const int lane = t & 63, ks = (t >> 6) % KS, R = (t >> 6) / KS;
    const int m = 32 * R + (lane & 31), h = lane >> 5;
    bf16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = 16 * ks + 8 * h + i;
        const float v = (m < M && k < K) ? w[(size_t)m * K + k] : 0.0f;
        const __bf16 hh = (__bf16)v;
        hi[i] = hh;
        lo[i] = (__bf16)(v - (float)hh);
    }
    frags[((size_t)(R * KS + ks) * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    frags[((size_t)(R * KS + ks) * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
}

constexpr int OOB = 0x7fff0000;       // a vector offset beyond any descriptor: loads return 0, stores are dropped
// buffer descriptor over [base, base + bytes) with the base made provably wave-uniform (cdl_fused2d.hip)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const float *base, size_t bytes)
{
    const size_t a = reinterpret_cast<size_t>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}

template <int PH, int PW, int SW, int MT, bool PROX, bool REV>
__global__ __launch_bounds__(ANT) void k_ana_m(cdl_geom g, const float *__restrict__ x,
                                               const uint4 *__restrict__ frags, float alpha,
                                               const float *__restrict__ zin, const float *__restrict__ gate,
                                               const float *__restrict__ tau, float *__restrict__ out,
                                               cdl_prox_args px, int tilesX, int tilesY, int KS,
                                               const float *__restrict__ zsup, float *__restrict__ dtp, int tpw)
{
    constexpr int XH = (ALY - 1) * SW + PH, XW = (ALX - 1) * SW + PW;
    constexpr int PS = ((XH * XW + 7) / 8) * 8;            // elements per plane
    extern __shared__ __align__(16) unsigned char smem[];
    const int NP = g.C * g.Pd;                             // image planes under a tile
    uint4 *wl = reinterpret_cast<uint4 *>(smem);           // [MT][KS][2][64] weight fragments
    int *koff = reinterpret_cast<int *>(smem + (size_t)MT * KS * 2 * 64 * 16);        // [KS*16] tap -> offset
    __bf16 *xh = reinterpret_cast<__bf16 *>(koff + KS * 16);                          // [NP][PS] hi
    __bf16 *xl = xh + (size_t)NP * PS;                                                // [NP][PS] lo
    float *stage = reinterpret_cast<float *>(xh);          // [32][ALX*ALY] epilogue staging, reuses the planes (32 KB)
    const size_t plane_bytes = (size_t)NP * PS * 4 > (size_t)32 * ALX * ALY * 4 ? (size_t)NP * PS * 4 : (size_t)32 * ALX * ALY * 4;
    float *tau_s = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(xh) + plane_bytes);   // [32 MT] thresholds
    const int Dz = g.D / g.sd, Hz = g.H / g.sh, Wz = g.W / g.sw;
    const int K = NP * PH * PW;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l32 = lane & 31, h = lane >> 5;

    // once per workgroup: weight fragments and the tap table
    // blockIdx.y selects a group of MT 32-channel tiles: many code channels (M = 169) are split over workgroups, each
    // with its own small fragment set and epilogue (one workgroup holding all 6 tiles ran 6 epilogue rounds per
    // tile at one workgroup per CU: 0.90 ms against 0.67 ms for the VALU kernel)
    const int r0 = blockIdx.y * MT;                        // first 32-channel tile of this workgroup
    for (int i = threadIdx.x; i < MT * KS * 2 * 64; i += ANT) wl[i] = frags[(size_t)r0 * KS * 2 * 64 + i];
    for (int k = threadIdx.x; k < KS * 16; k += ANT) {
        int o = 0;
        if (k < K) {
            const int kj = k % PW;
            int r = k / PW;
            const int ki = r % PH, p = r / PH;             // p = c * Pd + kd
            o = p * PS + ki * XW + kj;
        }
        koff[k] = o;
    }

    int b = blockIdx.x;
    const int txg = b % ((tilesX + tpw - 1) / tpw); b /= (tilesX + tpw - 1) / tpw;
    const int ty = b % tilesY; b /= tilesY;
    const int zd = b % Dz, n = b / Dz;
    // thresholds of this workgroup's channels (one sample, MT channel tiles): read once into LDS -- the epilogue read
    // tau[row] from global memory per element (a third of its memory instructions) -- and a workgroup-uniform flag
    // selects the clamp form of the shrinkage when none is negative (cdl_fused2d.hip)
    if (tau && threadIdx.x < 32 * MT) {
        const int m = 32 * r0 + threadIdx.x;
        tau_s[threadIdx.x] = m < g.M ? tau[(size_t)n * g.M + m] : 0.0f;
    }
    __syncthreads();
    const bool tau_neg = tau && __syncthreads_or(threadIdx.x < 32 * MT && !(tau_s[threadIdx.x] >= 0.0f));   // negative OR NaN
    const size_t slab = (size_t)Dz * Hz * Wz;
    const int pixbase = (wv * SW) * XW + l32 * SW;

    for (int tt = 0; tt < tpw; ++tt) {
        const int tx = txg * tpw + tt;
        if (tx >= tilesX) break;                           // uniform
        const int ybase = ty * ALY * SW - g.ph, xbase = tx * ALX * SW - g.pw;
        __syncthreads();                                   // previous tile's readers are done (and the tables are in)
        // PB planes x ITS elements per thread are loaded (clamped addresses, no branches) before any of them is converted:
        // one plane element at a time was a dependent global-load latency per element -- 27 in a row under the 9 x 9 x 5
        // filter, most of that launch's 84 us
        constexpr int ITS = (XH * XW + ANT - 1) / ANT, PB = ITS <= 2 ? 4 : 3;
        int eoff[ITS];
        bool eok[ITS];
#pragma unroll
        for (int u = 0; u < ITS; ++u) {
            const int i = threadIdx.x + u * ANT;
            const int col = i % XW, row = i / XW;
            const int yy = ybase + row, xx = xbase + col;
            eok[u] = i < XH * XW && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
            eoff[u] = min(max(yy, 0), g.H - 1) * g.W + min(max(xx, 0), g.W - 1);
        }
        for (int p0 = 0; p0 < NP; p0 += PB) {
            float v[PB][ITS];
#pragma unroll
            for (int pp = 0; pp < PB; ++pp) {
                const int p = min(p0 + pp, NP - 1);
                const int kd = p % g.Pd, c = p / g.Pd;
                const int d = zd * g.sd - g.pd + kd;
                const bool dok = p0 + pp < NP && d >= 0 && d < g.D;      // uniform
                const float *xplane = x + (((size_t)n * g.C + c) * g.D + min(max(d, 0), g.D - 1)) * g.H * g.W;
#pragma unroll
                for (int u = 0; u < ITS; ++u) {
                    const float t = xplane[eoff[u]];
                    v[pp][u] = (dok && eok[u]) ? t : 0.0f;
                }
            }
#pragma unroll
            for (int pp = 0; pp < PB; ++pp) {
                if (p0 + pp >= NP) break;                                // uniform
#pragma unroll
                for (int u = 0; u < ITS; ++u) {
                    const int i = threadIdx.x + u * ANT;
                    if (i < XH * XW) {
                        const __bf16 hh = (__bf16)v[pp][u];
                        xh[(p0 + pp) * PS + i] = hh;
                        xl[(p0 + pp) * PS + i] = (__bf16)(v[pp][u] - (float)hh);
                    }
                }
            }
        }
        __syncthreads();
        f32x16 acc[MT];
#pragma unroll
        for (int R = 0; R < MT; ++R)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[R][v] = 0.0f;
        // depth-2 software pipeline over the k-steps: the im2col operand of k-step ks+1 is gathered into the other
        // register set before the products of ks are issued, and the products alternate between the accumulator tiles
        // (rolled with one register set every k-step was table read -> gathers -> fragments -> dependent MFMAs in
        // series; the same change took cdl_fusedg.hip's analysis from 940 to 650 cycles per k-step)
        auto gather_b = [&](int ks, bf16x8 &bh, bf16x8 &bl) {
            const int4 o0 = *reinterpret_cast<const int4 *>(koff + 16 * ks + 8 * h);
            const int4 o1 = *reinterpret_cast<const int4 *>(koff + 16 * ks + 8 * h + 4);
            const int oo[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bh[i] = xh[oo[i] + pixbase];
                bl[i] = xl[oo[i] + pixbase];
            }
        };
        auto products = [&](int ks, const bf16x8 &bh, const bf16x8 &bl) {
            bf16x8 ah[MT], al[MT];
#pragma unroll
            for (int R = 0; R < MT; ++R) {
                ah[R] = __builtin_bit_cast(bf16x8, wl[((R * KS + ks) * 2 + 0) * 64 + lane]);
                al[R] = __builtin_bit_cast(bf16x8, wl[((R * KS + ks) * 2 + 1) * 64 + lane]);
            }
#pragma unroll
            for (int R = 0; R < MT; ++R) acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[R], bh, acc[R], 0, 0, 0);
#pragma unroll
            for (int R = 0; R < MT; ++R) acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[R], bl, acc[R], 0, 0, 0);
#pragma unroll
            for (int R = 0; R < MT; ++R) acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[R], bh, acc[R], 0, 0, 0);
        };
        {
            bf16x8 bh0, bl0, bh1, bl1;
            gather_b(0, bh0, bl0);
#pragma unroll 1
            for (int ks = 0; ks < KS; ks += 2) {
                if (ks + 1 < KS) gather_b(ks + 1, bh1, bl1);      // uniform branches
                products(ks, bh0, bl0);
                if (ks + 2 < KS) gather_b(ks + 2, bh0, bl0);
                if (ks + 1 < KS) products(ks + 1, bh1, bl1);
            }
        }
        // epilogue through LDS, one 32-channel accumulator tile at a time: the waves park tile R as [channel][pixel]
        // (register v is channel 32R + 8(v>>2) + 4h + (v&3) of pixel column l32 of row wv), then all 512 threads walk
        // the 32 x 256 values with a rolled loop -- 32-bit index arithmetic, pixel-contiguous fat accesses, and none of
        // the 16*MT-fold unrolled 64-bit addressing that sent the M = 169 variant to scratch
        const size_t nbase = (size_t)n * g.M * slab + (size_t)zd * Hz * Wz;
#pragma unroll
        for (int R = 0; R < MT; ++R) {
            __syncthreads();                               // image planes (first round) / previous tile consumed
#pragma unroll
            for (int v = 0; v < 16; ++v)
                stage[(8 * (v >> 2) + 4 * h + (v & 3)) * (ALX * ALY) + wv * ALX + l32] = acc[R][v];
            __syncthreads();
            // 16 values per thread in two batches of 8 whose loads are issued together (predicated, not branched);
            // 32-bit offsets inside the sample's block.  A thread keeps its PIXEL (element e = tid + 512 jj is channel
            // 2 jj + (tid >> 8) of pixel tid & 255), so everything but the channel stride is computed once per tile.
            static_assert(ANT == 2 * ALX * ALY, "thread <-> (channel parity, pixel) mapping");
            constexpr int NB = 8;                            // (fallback for > 2 GB sample blocks and the CSR maps)
            const float *zin_n = zin ? zin + nbase : nullptr;
            const float *gate_n = (zin && gate) ? gate + nbase : nullptr;
            float *out_n = out + nbase;
            const int pxl = threadIdx.x & (ALX * ALY - 1), chh = threadIdx.x >> 8;
            const int oy = ty * ALY + pxl / ALX, ox = tx * ALX + pxl % ALX;
            const bool okp = oy < Hz && ox < Wz;
            const int m0 = 32 * (r0 + R) + chh;
            const int idx0 = m0 * (int)slab + oy * Wz + ox, step = 2 * (int)slab;
            if (!PROX && (size_t)g.M * slab * 4 < ((size_t)1 << 31)) {
                // buffer descriptors over the sample's code block: the thread's pixel and channel parity are its vector
                // offset, the channel pair 2 jj travels in the scalar offset -- no address arithmetic per element, and
                // out-of-range elements are dropped by the range check (only the vector offset is checked: channels
                // beyond M are masked per thread in the last channel tile)
                const size_t sb = (size_t)n * g.M * slab;
                const __amdgpu_buffer_rsrc_t rs_in = uniform_rsrc(zin ? zin + sb : out + sb, (size_t)g.M * slab * 4);
                const __amdgpu_buffer_rsrc_t rs_g = uniform_rsrc((zin && gate) ? gate + sb : out + sb, (size_t)g.M * slab * 4);
                const __amdgpu_buffer_rsrc_t rs_out = uniform_rsrc(out + sb, (size_t)g.M * slab * 4);
                // reverse mode (zsup): out = [zsup != 0] (zin + alpha A x), and the threshold-gradient partial of this
                // tile, sum over its pixels of -sign(zsup) * out, per channel -- what cdl_tau_grad_gate computes from
                // the tensor this launch writes (three fat passes of the generic reverse sweep)
                const __amdgpu_buffer_rsrc_t rs_sup = uniform_rsrc(zsup ? zsup + sb : out + sb, (size_t)g.M * slab * 4);
                const int voff = okp ? (int)(((size_t)chh * slab + (size_t)zd * Hz * Wz + (size_t)oy * Wz + ox) * 4) : OOB;
                const int slab4 = (int)slab * 4;
                const int soff0 = __builtin_amdgcn_readfirstlane(32 * (r0 + R) * slab4);
                const bool tailR = 32 * (r0 + R) + 32 > g.M;            // uniform
                constexpr int NE = 32 * ALX * ALY / ANT;                // 16 elements per thread
                // (keep this variant under 128 VGPRs: two workgroups per CU)
                float bv[NE], gv[NE];
#pragma unroll
                for (int jj = 0; jj < NE; ++jj) {
                    const int vo = (tailR && m0 + 2 * jj >= g.M) ? OOB : voff;
                    bv[jj] = zin ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, vo, soff0 + 2 * jj * slab4, 0)) : 0.0f;
                    // (the slot of the input gate carries the support tensor in reverse mode: the two never coincide)
                    gv[jj] = REV ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_sup, vo, soff0 + 2 * jj * slab4, 0))
                           : ((zin && gate) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, vo, soff0 + 2 * jj * slab4, 0)) : 1.0f);
                }
                if constexpr (REV) {
                    float tg[NE];
#pragma unroll
                    for (int jj = 0; jj < NE; ++jj) {
                        const float sv = gv[jj];
                        const float u = fmaf(alpha, stage[threadIdx.x + jj * ANT], bv[jj]);
                        const float val = sv != 0.0f ? u : 0.0f;
                        tg[jj] = sv > 0.0f ? -val : (sv < 0.0f ? val : 0.0f);
                        const int vo = (tailR && m0 + 2 * jj >= g.M) ? OOB : voff;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rs_out, vo, soff0 + 2 * jj * slab4, 0);
                    }
                    // per-channel sums over the tile's 256 pixels through the staging buffer (every thread has read its
                    // values): [32 channels][256 pixels], 16 threads per channel add 16 pixels each, then a fixed 4-step
                    // exchange inside their 16-lane group -- order-fixed, deterministic
                    __syncthreads();
#pragma unroll
                    for (int jj = 0; jj < NE; ++jj) stage[(2 * jj + chh) * (ALX * ALY) + pxl] = tg[jj];
                    __syncthreads();
                    const int ch = threadIdx.x >> 4, part = threadIdx.x & 15;
                    float a = 0.0f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) a += stage[ch * (ALX * ALY) + part * 16 + i];
                    a += __shfl_xor(a, 8, 16);
                    a += __shfl_xor(a, 4, 16);
                    a += __shfl_xor(a, 2, 16);
                    a += __shfl_xor(a, 1, 16);
                    const int mm = 32 * (r0 + R) + ch;
                    if (part == 0 && mm < g.M) {
                        const int S = Dz * tilesY * tilesX, kt = (zd * tilesY + ty) * tilesX + tx;
                        dtp[((size_t)n * g.M + mm) * S + kt] = a;
                    }
                    continue;
                }
#pragma unroll
                for (int jj = 0; jj < NE; ++jj) {
                    const float base = gv[jj] == 0.0f ? 0.0f : bv[jj];
                    const float u = fmaf(alpha, stage[threadIdx.x + jj * ANT], base);
                    const float ts = tau_s[32 * R + chh + 2 * jj];          // (unused when tau == nullptr)
                    const float val = tau ? (tau_neg ? cdl_shrink(u, ts) : u - __builtin_amdgcn_fmed3f(u, -ts, ts)) : u;
                    const int vo = (tailR && m0 + 2 * jj >= g.M) ? OOB : voff;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rs_out, vo, soff0 + 2 * jj * slab4, 0);
                }
                continue;
            }
#pragma unroll 1
            for (int j0 = 0; j0 < 32 * ALX * ALY / ANT; j0 += NB) {
                float bv[NB], gv[NB];
                int ix[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const bool ok = okp && m0 + 2 * (j0 + j) < g.M;
                    ix[j] = ok ? idx0 + (j0 + j) * step : -1;
                    bv[j] = zin_n ? zin_n[ok ? ix[j] : 0] : 0.0f;
                    gv[j] = gate_n ? gate_n[ok ? ix[j] : 0] : 1.0f;
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int e = threadIdx.x + (j0 + j) * ANT;
                    const int m = min(m0 + 2 * (j0 + j), g.M - 1);
                    const float base = gv[j] == 0.0f ? 0.0f : bv[j];
                    const float u = fmaf(alpha, stage[e], base);
                    const int row = n * g.M + m;
                    const float ts = tau_s[32 * R + chh + 2 * (j0 + j)];     // (unused when tau == nullptr)
                    if (ix[j] >= 0)
                        out_n[ix[j]] = PROX ? cdl_prox_apply(px, u, nbase + ix[j], row)
                                            : (tau ? (tau_neg ? cdl_shrink(u, ts) : u - __builtin_amdgcn_fmed3f(u, -ts, ts)) : u);
                }
            }
        }
    }
}

struct Plan {
    int tpw;                               // tiles along x per workgroup
    int tilesX, tilesY, MT, KS, MTW, ngy;
    size_t groups, frag_uint4, lds;
};

bool plan_for(const cdl_geom *g, Plan *p)
{
    if (g->sw != g->sh || (g->sw != 1 && g->sw != 2)) return false;
    if (g->Pw != 3 && g->Pw != 5 && g->Pw != 7 && g->Pw != 9) return false;
    if (g->Ph != g->Pw && !(g->Ph == 9 && g->Pw == 5)) return false;      // rectangular planes: the shipped 9 x 9 x 5 net
    if (g->pw != g->Pw / 2 || g->ph != g->Ph / 2) return false;
    const int Dz = g->D / g->sd, Hz = g->H / g->sh, Wz = g->W / g->sw;
    p->MT = (g->M + 31) / 32;
    if (p->MT > 16) return false;                          // M <= 512
    if ((size_t)g->M * Dz * Hz * Wz >= ((size_t)1 << 31)) return false;   // 32-bit offsets inside one sample's code block
    const int K = g->C * g->Pd * g->Ph * g->Pw;
    p->KS = (K + 15) / 16;
    p->tilesX = (Wz + ALX - 1) / ALX;
    p->tilesY = (Hz + ALY - 1) / ALY;
    p->tpw = TPW;
    p->groups = (size_t)g->N * Dz * p->tilesY * ((p->tilesX + TPW - 1) / TPW);
    const size_t XH = (size_t)(ALY - 1) * g->sh + g->Ph, XW = (size_t)(ALX - 1) * g->sw + g->Pw;
    const size_t PS = ((XH * XW + 7) / 8) * 8;
    size_t planes = (size_t)g->C * g->Pd * PS * 2 * 2;
    if (planes < 32 * ALX * ALY * 4) planes = 32 * ALX * ALY * 4;          // the epilogue staging reuses them
    // 32-channel tiles per workgroup: two where planes and fragments fit 96 KB (two workgroups per CU); one, with the
    // whole CU's LDS behind it, for the deep 3-D filters (K = 9*9*5 = 405 taps over 9 planes: 53 + 56 KB)
    for (p->MTW = p->MT <= 2 ? p->MT : 2; ; p->MTW = 1) {
        p->lds = (size_t)p->MTW * p->KS * 2 * 64 * 16 + (size_t)p->KS * 16 * 4 + planes + (size_t)32 * p->MTW * 4;
        if (p->lds <= 96 * 1024 || p->MTW == 1) break;
    }
    if (p->lds > 96 * 1024 && (p->MTW != 1 || p->lds > LDS_MAX)) return false;
    p->ngy = (p->MT + p->MTW - 1) / p->MTW;                // channel groups (grid.y)
    p->frag_uint4 = (size_t)p->ngy * p->MTW * p->KS * 2 * 64;   // padded to whole groups
    // small launches: the VALU kernels do better -- unless the filter is so deep (K >= 256) that they crawl, where the
    // channel groups count as workgroups too
    // M > 64 (three or more channel tiles, channel groups on grid.y): the VALU kernels are poor there too, so a launch that
    // would leave CUs idle takes one tile per workgroup (single frames of the M = 169 nets: 48 -> 192 workgroups) and
    // counts every workgroup
    if (p->MT >= 3 && p->groups * p->ngy < (size_t)cdl_cu_count()) {
        p->tpw = 1;
        p->groups = (size_t)g->N * Dz * p->tilesY * p->tilesX;
    }
    const size_t wgs = (K >= 256 || p->MT >= 3) ? p->groups * p->ngy : p->groups;
    if (wgs < (p->MT >= 3 ? 48 : 96) || p->groups >= ((size_t)1 << 31)) return false;
    return true;
}

template <int PH, int PW, int SW, int MT, bool PROX, bool REV>
int launch_mtp(const cdl_geom *g, const Plan &p, const float *x, const uint4 *frags, float alpha, const float *zin,
               const float *gate, const float *tau, float *out, const cdl_prox_args &px, hipStream_t st,
               const float *zsup, float *dtp)
{
    if (int rc = cdl_ensure_dynamic_lds((const void *)k_ana_m<PH, PW, SW, MT, PROX, REV>, p.lds > 96 * 1024 ? LDS_MAX : 96 * 1024))
        return rc;
    k_ana_m<PH, PW, SW, MT, PROX, REV><<<dim3((unsigned)p.groups, (unsigned)p.ngy), ANT, p.lds, st>>>(
        *g, x, frags, alpha, zin, gate, tau, out, px, p.tilesX, p.tilesY, p.KS, zsup, dtp, p.tpw);
    CDL_LAUNCH_CHECK();
    return 0;
}

template <int PH, int PW, int SW, int MT>
int launch_mt(const cdl_geom *g, const Plan &p, const float *x, const uint4 *frags, float alpha, const float *zin,
              const float *gate, const float *tau, float *out, const cdl_prox_args &px, hipStream_t st,
              const float *zsup, float *dtp)
{
    if (px.zp) return launch_mtp<PH, PW, SW, MT, true, false>(g, p, x, frags, alpha, zin, gate, tau, out, px, st, nullptr, nullptr);
    if (zsup) return launch_mtp<PH, PW, SW, MT, false, true>(g, p, x, frags, alpha, zin, gate, tau, out, px, st, zsup, dtp);
    return launch_mtp<PH, PW, SW, MT, false, false>(g, p, x, frags, alpha, zin, gate, tau, out, px, st, nullptr, nullptr);
}

template <int PH, int PW, int SW>
int launch(const cdl_geom *g, const Plan &p, const float *x, const float *w, float alpha, const float *zin,
           const float *gate, const float *tau, float *out, const cdl_prox_args &px, float *ws, hipStream_t st,
           const float *zsup = nullptr, float *dtp = nullptr)
{
    uint4 *frags = reinterpret_cast<uint4 *>(ws);
    const int ntile = p.ngy * p.MTW;
    const int nprep = ntile * p.KS * 64;
    k_ana_prep<<<(nprep + 255) / 256, 256, 0, st>>>(w, frags, g->M, g->C * g->Pd * g->Ph * g->Pw, ntile, p.KS);
    CDL_LAUNCH_CHECK();
    if (p.MTW == 1) return launch_mt<PH, PW, SW, 1>(g, p, x, frags, alpha, zin, gate, tau, out, px, st, zsup, dtp);
    return launch_mt<PH, PW, SW, 2>(g, p, x, frags, alpha, zin, gate, tau, out, px, st, zsup, dtp);
}

// dt0[m] = sum_n sum_k s[(n M + m) S + k], dt1[m] = sum_n c[n] (...): one wave per channel; the lanes split the S tile
// partials of a sample, a fixed exchange tree adds them, the 8 waves walk every 8th sample and are added in order --
// deterministic.  (One thread
// per channel walking N x S strided values took 0.40 ms at the s2030 shape: as long as the analysis itself.)
__global__ __launch_bounds__(512) void k_ana_tau_final(const float *__restrict__ s, const float *__restrict__ c,
                                                       float *__restrict__ dt0, float *__restrict__ dt1, int N, int M, int S)
{
    __shared__ float r0[8], r1[8];
    const int m = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float a0 = 0.0f, a1 = 0.0f;
    for (int n = wv; n < N; n += 8) {                      // wave wv takes samples wv, wv + 8, ...
        const float *row = s + (size_t)(n * M + m) * S;
        float v = 0.0f;
        for (int k = lane; k < S; k += 64) v += row[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        a0 += v;
        if (c) a1 = fmaf(c[n], v, a1);
    }
    if (lane == 0) { r0[wv] = a0; r1[wv] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { t0 += r0[w]; t1 += r1[w]; }
        dt0[m] = t0;
        dt1[m] = t1;
    }
}

}  // namespace

// Reverse-sweep form: out = [zsup != 0] (zin + alpha A x) and (dt0, dt1) = cdl_tau_grad(out, zsup, c), one fat launch.
// Workspace: cdl_mfma_analysis_rev_ws_floats(g) (fragments + N M S tile partials).  CDL_EUNSUPPORTED: the caller composes
// cdl_analysis_ws and cdl_tau_grad_gate.
size_t cdl_mfma_analysis_rev_ws_floats(const cdl_geom *g)
{
    Plan p;
    if (!plan_for(g, &p)) return 0;
    const size_t Dz = g->D / g->sd, Hz = g->H / g->sh, Wz = g->W / g->sw;
    if ((size_t)g->M * Dz * Hz * Wz * 4 >= ((size_t)1 << 31)) return 0;     // the descriptor epilogue only
    return p.frag_uint4 * 4 + (size_t)g->N * g->M * Dz * p.tilesY * p.tilesX;
}

int cdl_mfma_analysis_rev(const cdl_geom *g, const float *x, const float *w, float alpha, const float *zin,
                          const float *zsup, const float *c, float *dt0, float *dt1, float *out, float *ws,
                          size_t ws_floats, void *stream)
{
    Plan p;
    const size_t need = cdl_mfma_analysis_rev_ws_floats(g);
    if (!need || !plan_for(g, &p) || !ws || ws_floats < need || !zsup || !dt0 || !dt1) return CDL_EUNSUPPORTED;
    if ((reinterpret_cast<size_t>(ws) & 15) != 0) return CDL_EUNSUPPORTED;
    float *dtp = ws + p.frag_uint4 * 4;
    const int S_ = (g->D / g->sd) * p.tilesY * p.tilesX;
    int rc = CDL_EUNSUPPORTED;
#define CDL_M(PH_, P_, S2_) \
    if (g->Ph == PH_ && g->Pw == P_ && g->sw == S2_) \
        rc = launch<PH_, P_, S2_>(g, p, x, w, alpha, zin, nullptr, nullptr, out, cdl_prox_args{}, ws, S(stream), zsup, dtp)
    CDL_M(3, 3, 1); else CDL_M(5, 5, 1); else CDL_M(7, 7, 1); else CDL_M(9, 9, 1);
    else CDL_M(3, 3, 2); else CDL_M(5, 5, 2); else CDL_M(7, 7, 2); else CDL_M(9, 9, 2);
    else CDL_M(9, 5, 1); else CDL_M(9, 5, 2);
#undef CDL_M
    if (rc) return rc;
    k_ana_tau_final<<<g->M, 512, 0, S(stream)>>>(dtp, c, dt0, dt1, g->N, g->M, S_);
    CDL_LAUNCH_CHECK();
    return 0;
}

size_t cdl_mfma_analysis_ws_floats(const cdl_geom *g)
{
    Plan p;
    return plan_for(g, &p) ? p.frag_uint4 * 4 : 0;
}

// CDL_EUNSUPPORTED: the caller falls back to the VALU kernels
int cdl_mfma_analysis(const cdl_geom *g, const float *x, const float *w, float alpha, const float *zin,
                      const float *gate, const float *tau, float *out, const cdl_prox_args &px, float *ws,
                      size_t ws_floats, void *stream)
{
    Plan p;
    if (!plan_for(g, &p) || !ws || ws_floats < p.frag_uint4 * 4) return CDL_EUNSUPPORTED;
    if ((reinterpret_cast<size_t>(ws) & 15) != 0) return CDL_EUNSUPPORTED;
#define CDL_M(PH_, P_, S_) \
    if (g->Ph == PH_ && g->Pw == P_ && g->sw == S_) \
        return launch<PH_, P_, S_>(g, p, x, w, alpha, zin, gate, tau, out, px, ws, S(stream))
    CDL_M(3, 3, 1); CDL_M(5, 5, 1); CDL_M(7, 7, 1); CDL_M(9, 9, 1);
    CDL_M(3, 3, 2); CDL_M(5, 5, 2); CDL_M(7, 7, 2); CDL_M(9, 9, 2);
    CDL_M(9, 5, 1); CDL_M(9, 5, 2);
#undef CDL_M
    return CDL_EUNSUPPORTED;
}
