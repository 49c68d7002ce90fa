// Synthesis (transposed convolution) on the matrix cores for the shapes the fused 2-D kernel does not take:
// any C, 2-D or 3-D, stride 1 or 2, odd square filter planes up to 9 x 9 and 9 x 5 (reference F.conv_transpose2d/3d at
// model/net.py:87,90,205,210).
//
//   col[tap][px] = sum_m W^T[tap][m] * z[m][px]          one GEMM per (image channel c, depth tap kd):
//                                                        M_dim = Ph*Pw taps (no padding waste beyond 32-row tiles),
//                                                        K_dim = code channels, N_dim = 32 code pixels per block
//   out[c][d][Y][X] += col[(ki,kj)][(Y+ph-ki)/s, (X+pw-kj)/s]   (col2im)
//
// A workgroup owns a 32 x 8 tile of code pixels of one (n, zd).  Per (c, kd) group its 8 waves run the GEMM for
// one pixel row each (split-bf16 x3 like the fused kernel: fp32-grade), park the accumulator tiles in an LDS `col`
// tile and then gather them, in a fixed order, into the workgroup's output patch ((8-1)s+Ph rows x (32-1)s+Pw
// columns, thin).  Patches of neighbouring tiles and depth taps overlap; k_synth_assemble adds them in a
// fixed order (deterministic, no atomics) and applies alpha, mask and sub.  The fat tensor is read with no halo.
#include "cdl_common.h"

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TCX = 32, TCY = 8;      // code-pixel tile of a workgroup: 8 blocks of 32 pixels, one per wave
constexpr int SNT = 64 * TCY;         // 8 waves: with 64-108 KB of LDS only one workgroup fits a CU
constexpr int OOB = 0x7fff0000;       // a vector offset beyond any descriptor: the load returns 0
constexpr int KSC = 4;                // k-steps whose code values are loaded together (32 loads in flight per lane)

// ------------------------------------------------------------------------------------------
// filters (M, C, Pd, Ph, Pw) -> MFMA A-operand fragments of W^T per (c, kd) group:
// frag[((g*RT + R)*KS + ks)*2 + hl][lane] (16 B): lane (row = tap 32R + (lane & 31), h = lane >> 5) holds the 8
// channels m = 16ks + 8h + i of that tap (zero beyond M or Ph*Pw), hi or lo bf16 part.
__global__ void k_synth_prep(const float *__restrict__ w, uint4 *__restrict__ frags, int M, int C, int Pd,
                             int taps, int RT, int KS)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = C * Pd * RT * KS * 64;
    if (t >= total) return;
    const int lane = t & 63;
    int r = t >> 6;
    const int ks = r % KS; r /= KS;
    const int R = r % RT; r /= RT;
    const int g = r;                                        // c * Pd + kd
    const int row = lane & 31, h = lane >> 5;
    const int tap = 32 * R + row;
    bf16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = 16 * ks + 8 * h + i;
        float v = 0.0f;
        if (m < M && tap < taps) v = w[((size_t)m * C * Pd + g) * taps + tap];
        const __bf16 hh = (__bf16)v;
        hi[i] = hh;
        lo[i] = (__bf16)(v - (float)hh);
    }
    const size_t base = (((size_t)g * RT + R) * KS + ks) * 2;
    frags[(base + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    frags[(base + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
}

// buffer descriptor over [base, base + bytes) with the base made provably wave-uniform (otherwise every access gets a
// waterfall loop, cdl_fused2d.hip)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const float *base, size_t bytes)
{
    const size_t a = reinterpret_cast<size_t>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}

// ------------------------------------------------------------------------------------------
template <int PH, int PW, int SW, int KSM>
__global__ __launch_bounds__(SNT) void k_synth_m(cdl_geom g, const float *__restrict__ z,
                                                 const float *__restrict__ gate, const uint4 *__restrict__ frags,
                                                 float *__restrict__ patches, int tilesX, int tilesY, int KS, int KCH,
                                                 int ntiles, int gspan)
{
    constexpr int T = PH * PW, RT = (T + 31) / 32;
    constexpr int PY = (TCY - 1) * SW + PH, PX = (TCX - 1) * SW + PW;
    extern __shared__ __align__(16) unsigned char smem[];
    float *col = reinterpret_cast<float *>(smem);                              // [RT*32][TCX*TCY]
    uint4 *wl = reinterpret_cast<uint4 *>(smem + (size_t)RT * 32 * TCX * TCY * 4);   // [RT][KCH][2][64]: KCH k-steps of
                                                            // weight fragments at a time (all KS when they fit)
    const int Dz = g.D / g.sd, Hz = g.H / g.sh, Wz = g.W / g.sw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c32 = lane & 31, h = lane >> 5;
    const int G = g.C * g.Pd;
    const size_t slab = (size_t)Dz * Hz * Wz;                                  // one code channel of one sample
    const int nfr = RT * KS * 2 * 64;                                          // uint4 fragments per group
    // Persistent workgroups (grid = min(tiles, CUs)).  With ONE group whose fragments fit LDS whole (2-D nets: the
    // shipped M = 169 bank is 44 KB) they are loaded once per workgroup, not once per tile -- 4096 tiles x 44 KB of
    // L2 reads and their index arithmetic per launch at the s2030 shape.
    const bool resident = G == 1 && KCH >= KS;
    if (resident) {
        for (int i = threadIdx.x; i < nfr; i += SNT) {
            const int R = i / (KS * 2 * 64), rem = i % (KS * 2 * 64);
            wl[(size_t)R * KCH * 2 * 64 + rem] = frags[i];
        }
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int b = tile;
    const int tx = b % tilesX; b /= tilesX;
    const int ty = b % tilesY; b /= tilesY;
    const int zd = b % Dz, n = b / Dz;
    const int cx = tx * TCX + c32;
    // fat operand through buffer descriptors of the sample's code block (< 2 GB, plan_for): the channel travels in the
    // SCALAR offset (16 ks + i planes), the lane's part -- pixel and channel half 8h -- in the vector offset, so a
    // load costs no address arithmetic (64-bit pointer math per element was a quarter of the kernel's VALU work at
    // M = 169).  Only the vector offset is range-checked: channels beyond M are masked per lane in the last k-step.
    const size_t sbase = (size_t)n * g.M * slab;
    const __amdgpu_buffer_rsrc_t rz = uniform_rsrc(z + sbase, (size_t)g.M * slab * 4);
    const __amdgpu_buffer_rsrc_t rg = uniform_rsrc(gate ? gate + sbase : z + sbase, (size_t)g.M * slab * 4);
    const int slab4 = (int)slab * 4;

    // KSM > 0: the code values of this wave's pixel row are loaded and split ONCE (KS <= KSM k-steps in registers) and
    // reused by every (c, kd) group; KSM == 0 (many code channels): reloaded per group
    const int blk = wv;                                     // code row of this wave's block inside the tile
    const int cy = ty * TCY + blk;
    const bool ok = cy < Hz && cx < Wz;
    const int voff = ok ? (int)(((size_t)(8 * h) * slab + (size_t)zd * Hz * Wz + (size_t)cy * Wz + cx) * 4) : OOB;
    auto load8 = [&](int ks, float (&zv)[8]) {               // channels 16 ks + 8 h + (0..7) of this lane's pixel
        const int soff0 = __builtin_amdgcn_readfirstlane(16 * ks * slab4);
        const bool tail = 16 * ks + 16 > g.M;                // uniform
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int vo = (tail && 16 * ks + 8 * h + i >= g.M) ? OOB : voff;
            float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, vo, soff0 + i * slab4, 0));
            if (gate) {
                const float gv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, vo, soff0 + i * slab4, 0));
                if (gv == 0.0f) v = 0.0f;
            }
            zv[i] = v;
        }
    };
    bf16x8 cbh[KSM > 0 ? KSM : 1], cbl[KSM > 0 ? KSM : 1];
    if (KSM > 0) {
#pragma unroll
        for (int q = 0; q < KSM; ++q) {
            float zv[8];
            load8(q, zv);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const __bf16 hh = (__bf16)zv[i];
                cbh[q][i] = hh;
                cbl[q][i] = (__bf16)(zv[i] - (float)hh);
            }
        }
    }

    // blockIdx.y: the (c, kd) groups [y gspan, (y + 1) gspan) -- a group's patch is its own, so a launch with fewer tiles than
    // CUs (one clip of the 3-D nets) spreads them over workgroups
    const int grp_end = min(G, (int)(blockIdx.y + 1) * gspan);
    for (int grp = blockIdx.y * gspan; grp < grp_end; ++grp) {
        const int kd = grp % g.Pd;
        const int d = zd * g.sd - g.pd + kd;
        if (d < 0 || d >= g.D) continue;                    // uniform: this depth tap falls outside the image
        f32x16 acc[RT];
#pragma unroll
        for (int R = 0; R < RT; ++R)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[R][v] = 0.0f;
        for (int kc0 = 0; kc0 < KS; kc0 += KCH) {           // weight fragments of KCH k-steps at a time
            const int kcn = min(KCH, KS - kc0);
            if (!resident) {
                __syncthreads();                            // previous chunk's / group's readers are done
                for (int i = threadIdx.x; i < RT * kcn * 2 * 64; i += SNT) {
                    const int R = i / (kcn * 2 * 64), rem = i % (kcn * 2 * 64);
                    wl[(size_t)R * KCH * 2 * 64 + rem] = frags[(size_t)grp * nfr + ((size_t)R * KS + kc0) * 2 * 64 + rem];
                }
            }
            __syncthreads();                                // fragments in place (resident: the one-off load, first tile)
            if (KSM > 0) {                                  // KS <= KSM: one chunk, code fragments from registers
#pragma unroll
                for (int q = 0; q < KSM; ++q) {
                    if (q >= KS) break;
#pragma unroll
                    for (int R = 0; R < RT; ++R) {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, wl[((R * KCH + q) * 2 + 0) * 64 + lane]);
                        const bf16x8 al = __builtin_bit_cast(bf16x8, wl[((R * KCH + q) * 2 + 1) * 64 + lane]);
                        acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, cbh[q], acc[R], 0, 0, 0);
                        acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cbl[q], acc[R], 0, 0, 0);
                        acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cbh[q], acc[R], 0, 0, 0);
                    }
                }
            } else
            for (int ks0 = kc0; ks0 < kc0 + kcn; ks0 += KSC) {
                float zv[KSC][8];
#pragma unroll
                for (int q = 0; q < KSC; ++q) load8(ks0 + q, zv[q]);
#pragma unroll
                for (int q = 0; q < KSC; ++q) {
                    const int ks = ks0 + q;
                    if (ks >= kc0 + kcn) break;
                    bf16x8 bh, bl;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const __bf16 hh = (__bf16)zv[q][i];
                        bh[i] = hh;
                        bl[i] = (__bf16)(zv[q][i] - (float)hh);
                    }
#pragma unroll
                    for (int R = 0; R < RT; ++R) {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, wl[((R * KCH + ks - kc0) * 2 + 0) * 64 + lane]);
                        const bf16x8 al = __builtin_bit_cast(bf16x8, wl[((R * KCH + ks - kc0) * 2 + 1) * 64 + lane]);
                        acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[R], 0, 0, 0);
                        acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[R], 0, 0, 0);
                        acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[R], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();                                    // previous group's gather has read col (and the chunk loop is done)
        {
            // accumulator register v of tile R is tap 32R + 8(v>>2) + 4h + (v&3) of pixel column c32
#pragma unroll
            for (int R = 0; R < RT; ++R)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int tap = 32 * R + 8 * (v >> 2) + 4 * h + (v & 3);
                    col[tap * (TCX * TCY) + blk * TCX + c32] = acc[R][v];
                }
        }
        __syncthreads();
        // col2im into the patch of this (tile, group): fixed (ki, kj) order per output
        float *patch = patches + ((size_t)tile * G + grp) * (PY * PX);
        if constexpr (SW == 2) {
            // stride 2: output (Yl, Xl) collects tap (ki, kj) = (Yl % 2 + 2 j, Xl % 2 + 2 i) of code pixel (Yl / 2 - j,
            // Xl / 2 - i), j, i < ceil(P / 2): a static 4 x 4 (P = 7) candidate grid -- reads at clamped addresses issued
            // together, added in a fixed order under predicates -- instead of two data-dependent loops per output
            constexpr int NJ = (PH + 1) / 2, NI = (PW + 1) / 2;
            for (int o = threadIdx.x; o < PY * PX; o += SNT) {
                const int Yl = o / PX, Xl = o % PX;
                const int zy0 = Yl >> 1, zx0 = Xl >> 1, ky0 = Yl & 1, kx0 = Xl & 1;
                float v[NJ][NI];
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int zy = min(max(zy0 - j, 0), TCY - 1), zx = min(max(zx0 - i, 0), TCX - 1);
                        const int ki = min(ky0 + 2 * j, PH - 1), kj = min(kx0 + 2 * i, PW - 1);
                        v[j][i] = col[(ki * PW + kj) * (TCX * TCY) + zy * TCX + zx];
                    }
                float sum = 0.0f;
#pragma unroll
                for (int j = NJ - 1; j >= 0; --j)                  // ascending code row, then ascending code column
#pragma unroll
                    for (int i = NI - 1; i >= 0; --i) {
                        const bool ok = zy0 - j >= 0 && zy0 - j < TCY && ky0 + 2 * j < PH &&
                                        zx0 - i >= 0 && zx0 - i < TCX && kx0 + 2 * i < PW;
                        if (ok) sum += v[j][i];
                    }
                patch[o] = sum;
            }
        } else
        for (int o = threadIdx.x; o < PY * PX; o += SNT) {
            const int Yl = o / PX, Xl = o % PX;
            float sum = 0.0f;
            // ki = Yl - SW*zy with 0 <= zy < TCY, 0 <= ki < PH
            for (int zy = (Yl >= PH ? (Yl - PH + SW) / SW : 0); zy < TCY && SW * zy <= Yl; ++zy) {
                const int ki = Yl - SW * zy;
                for (int zx = (Xl >= PW ? (Xl - PW + SW) / SW : 0); zx < TCX && SW * zx <= Xl; ++zx) {
                    const int kj = Xl - SW * zx;
                    sum += col[(ki * PW + kj) * (TCX * TCY) + zy * TCX + zx];
                }
            }
            patch[o] = sum;
        }
    }
    }   // tiles of this workgroup
}

// out[n][c][d][Y][X] = mask * alpha * sum over depth taps and covering tiles of the patches - sub
template <int PH, int PW, int SW>
__global__ __launch_bounds__(256) void k_synth_assemble(cdl_geom g, const float *__restrict__ patches,
                                                        const float *__restrict__ mask,
                                                        const float *__restrict__ sub, float alpha,
                                                        float *__restrict__ out, int tilesX, int tilesY)
{
    constexpr int PY = (TCY - 1) * SW + PH, PX = (TCX - 1) * SW + PW;
    const int X = blockIdx.x * 256 + threadIdx.x;
    if (X >= g.W) return;
    const int Y = blockIdx.y;
    int r = blockIdx.z;
    const int d = r % g.D; r /= g.D;
    const int c = r % g.C, n = r / g.C;
    const int Dz = g.D / g.sd, G = g.C * g.Pd;
    // tiles whose patch covers Y: 0 <= Y + ph - ty*TCY*SW < PY   (same for X)
    const int ay = Y + g.ph, ax = X + g.pw;
    const int ty_hi = min(tilesY - 1, ay / (TCY * SW)), ty_lo = max(0, (ay - PY + TCY * SW) / (TCY * SW));
    const int tx_hi = min(tilesX - 1, ax / (TCX * SW)), tx_lo = max(0, (ax - PX + TCX * SW) / (TCX * SW));
    float sum = 0.0f;
    for (int kd = 0; kd < g.Pd; ++kd) {
        const int td = d + g.pd - kd;
        if (td < 0 || td % g.sd) continue;
        const int zd = td / g.sd;
        if (zd >= Dz) continue;
        const int grp = c * g.Pd + kd;
        for (int ty = ty_lo; ty <= ty_hi; ++ty) {
            const int Yl = ay - ty * TCY * SW;
            for (int tx = tx_lo; tx <= tx_hi; ++tx) {
                const int Xl = ax - tx * TCX * SW;
                const size_t tile = (((size_t)n * Dz + zd) * tilesY + ty) * tilesX + tx;
                sum += patches[((tile * G + grp) * PY + Yl) * PX + Xl];
            }
        }
    }
    const size_t i = ((((size_t)n * g.C + c) * g.D + d) * g.H + Y) * g.W + X;
    float v = alpha * sum;
    if (mask) v *= mask[i];
    if (sub) v -= sub[i];
    out[i] = v;
}

// W % 4 == 0: 4 consecutive pixels x one row per thread, 16-byte thin accesses; a pixel lies in at most 2 x 2 patches per
// depth tap (the patch halo P - 1 is smaller than a tile): the row candidates are wave-uniform, the column candidates
// are read side by side, and the terms are added in k_synth_assemble's order (depth tap, tile row, tile column).
template <int PH, int PW, int SW>
__global__ __launch_bounds__(256) void k_synth_assemble4(cdl_geom g, const float *__restrict__ patches,
                                                         const float *__restrict__ mask,
                                                         const float *__restrict__ sub, float alpha,
                                                         float *__restrict__ out, int tilesX, int tilesY)
{
    constexpr int PY = (TCY - 1) * SW + PH, PX = (TCX - 1) * SW + PW;
    constexpr int TWX = TCX * SW, TWY = TCY * SW;             // image pixels covered by a tile
    static_assert(PX - TWX < TWX, "at most two tile columns per pixel (the tile-row loop is general)");
    const int X0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X0 >= g.W || Y >= g.H) return;
    int r = blockIdx.z;
    const int d = r % g.D; r /= g.D;
    const int c = r % g.C, n = r / g.C;
    const int Dz = g.D / g.sd, G = g.C * g.Pd;
    const int ay = Y + g.ph;
    const int ty_hi = min(tilesY - 1, ay / TWY), ty_lo = max(0, (ay - PY + TWY) / TWY);
    int offh[4], offl[4];
    bool has_lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ax = X0 + e + g.pw;
        const int tx_hi = min(tilesX - 1, ax / TWX), tx_lo = max(0, (ax - PX + TWX) / TWX);
        has_lo[e] = tx_lo < tx_hi;
        offh[e] = tx_hi * (G * PY * PX) + (ax - tx_hi * TWX);
        offl[e] = tx_lo * (G * PY * PX) + (ax - tx_lo * TWX);
    }
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int kd = 0; kd < g.Pd; ++kd) {
        const int td = d + g.pd - kd;
        if (td < 0 || td % g.sd) continue;                    // uniform
        const int zd = td / g.sd;
        if (zd >= Dz) continue;
        const int grp = c * g.Pd + kd;
        for (int ty = ty_lo; ty <= ty_hi; ++ty) {             // uniform, 1 or 2 rows of tiles
            const float *row = patches + (((((size_t)n * Dz + zd) * tilesY + ty) * tilesX * G + grp) * PY + (ay - ty * TWY)) * PX;
            float lo[4], hi[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = row[offh[e]];
                lo[e] = has_lo[e] ? row[offl[e]] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (has_lo[e]) acc[e] += lo[e];
                acc[e] += hi[e];
            }
        }
    }
    const size_t i = ((((size_t)n * g.C + c) * g.D + d) * g.H + Y) * g.W + X0;
    float4 v = make_float4(alpha * acc[0], alpha * acc[1], alpha * acc[2], alpha * acc[3]);
    if (mask) {
        const float4 m = *reinterpret_cast<const float4 *>(mask + i);
        v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
    }
    if (sub) {
        const float4 sv = *reinterpret_cast<const float4 *>(sub + i);
        v.x -= sv.x; v.y -= sv.y; v.z -= sv.z; v.w -= sv.w;
    }
    *reinterpret_cast<float4 *>(out + i) = v;
}

struct Plan {
    int tilesX, tilesY, RT, KS, KCH;
    size_t tiles, frag_uint4, patch_floats, lds;
};

bool plan_for(const cdl_geom *g, Plan *p)
{
    if (g->sw != g->sh || (g->sw != 1 && g->sw != 2)) return false;
    if (g->Pw != 3 && g->Pw != 5 && g->Pw != 7 && g->Pw != 9) return false;
    if (g->Ph != g->Pw && !(g->Ph == 9 && g->Pw == 5)) return false;      // rectangular planes: the shipped 9 x 9 x 5 net
    if (g->pw != g->Pw / 2 || g->ph != g->Ph / 2) return false;
    const int Dz = g->D / g->sd, Hz = g->H / g->sh, Wz = g->W / g->sw;
    p->tilesX = (Wz + TCX - 1) / TCX;
    p->tilesY = (Hz + TCY - 1) / TCY;
    p->RT = (g->Ph * g->Pw + 31) / 32;
    p->KS = (g->M + 15) / 16;
    p->tiles = (size_t)g->N * Dz * p->tilesX * p->tilesY;
    p->frag_uint4 = (size_t)g->C * g->Pd * p->RT * p->KS * 2 * 64;
    const int PY = (TCY - 1) * g->sh + g->Ph, PX = (TCX - 1) * g->sw + g->Pw;
    p->patch_floats = p->tiles * g->C * g->Pd * PY * PX;
    p->KCH = p->KS;                                        // weight fragments resident per group, or KSC k-steps at a time
    p->lds = (size_t)p->RT * 32 * TCX * TCY * 4 + (size_t)p->RT * p->KCH * 2 * 64 * 16;
    if (p->lds > 150 * 1024) {
        p->KCH = KSC;
        p->lds = (size_t)p->RT * 32 * TCX * TCY * 4 + (size_t)p->RT * p->KCH * 2 * 64 * 16;
    }
    if (p->lds > 150 * 1024) return false;
    if (p->tiles >= ((size_t)1 << 31) || g->H > 65535 || (size_t)g->N * g->C * g->D > 65535) return false;
    if ((size_t)g->M * Dz * Hz * Wz * 4 >= ((size_t)1 << 31)) return false;   // one sample's code block behind a 32-bit buffer descriptor
    if (p->patch_floats > ((size_t)1 << 28)) return false;                     // 1 GiB of patches: not worth it
    return true;
}

template <int PH, int PW, int SW>
int launch(const cdl_geom *g, const Plan &p, const float *z, const float *gate, const float *w, float alpha,
           const float *mask, const float *sub, float *out, float *ws, hipStream_t st)
{
    uint4 *frags = reinterpret_cast<uint4 *>(ws);                               // 16-byte aligned: ws from hipMalloc
    float *patches = ws + p.frag_uint4 * 4;
    const int nprep = g->C * g->Pd * p.RT * p.KS * 64;
    k_synth_prep<<<(nprep + 255) / 256, 256, 0, st>>>(w, frags, g->M, g->C, g->Pd, g->Ph * g->Pw, p.RT, p.KS);
    CDL_LAUNCH_CHECK();
    if (int rc = cdl_ensure_dynamic_lds((const void *)k_synth_m<PH, PW, SW, 0>, 150 * 1024)) return rc;
    if (int rc = cdl_ensure_dynamic_lds((const void *)k_synth_m<PH, PW, SW, 4>, 150 * 1024)) return rc;
    const size_t cus = (size_t)cdl_cu_count();
    const unsigned grid_m = (unsigned)(p.tiles < cus ? p.tiles : cus);      // one workgroup per CU (64-108 KB of LDS each)
    const int G = g->C * g->Pd;
    int gy = 1;                                                             // group chunks while CUs would idle
    while (gy < G && (size_t)grid_m * (gy + 1) <= cus) ++gy;
    const int gspan = (G + gy - 1) / gy;
    const dim3 grid_s(grid_m, (unsigned)((G + gspan - 1) / gspan));
    if (p.KS <= 4 && G > 1)                  // several groups share the code values: keep their fragments in registers
        k_synth_m<PH, PW, SW, 4><<<grid_s, SNT, p.lds, st>>>(*g, z, gate, frags, patches, p.tilesX, p.tilesY, p.KS, p.KCH, (int)p.tiles, gspan);
    else
        k_synth_m<PH, PW, SW, 0><<<grid_s, SNT, p.lds, st>>>(*g, z, gate, frags, patches, p.tilesX, p.tilesY, p.KS, p.KCH, (int)p.tiles, gspan);
    CDL_LAUNCH_CHECK();
    const size_t al = reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(mask) | reinterpret_cast<size_t>(sub);
    if ((g->W & 3) == 0 && (al & 15) == 0 && !cdl_opts().scalar_assemble) {     // (CDL_SCALAR_ASSEMBLE=1: the scalar form, for tests)
        dim3 grid4((unsigned)((g->W + 255) / 256), (unsigned)((g->H + 3) / 4), (unsigned)(g->N * g->C * g->D));
        k_synth_assemble4<PH, PW, SW><<<grid4, 256, 0, st>>>(*g, patches, mask, sub, alpha, out, p.tilesX, p.tilesY);
        CDL_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((unsigned)((g->W + 255) / 256), (unsigned)g->H, (unsigned)(g->N * g->C * g->D));
    k_synth_assemble<PH, PW, SW><<<grid, 256, 0, st>>>(*g, patches, mask, sub, alpha, out, p.tilesX, p.tilesY);
    CDL_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// workspace (floats) of the MFMA synthesis for this geometry, 0 when it has no such kernel
size_t cdl_mfma_synthesis_ws_floats(const cdl_geom *g)
{
    Plan p;
    if (!plan_for(g, &p)) return 0;
    return p.frag_uint4 * 4 + p.patch_floats;
}

// CDL_EUNSUPPORTED: the caller falls back to the VALU kernels
int cdl_mfma_synthesis(const cdl_geom *g, const float *z, const float *gate, const float *w, float alpha,
                       const float *mask, const float *sub, float *out, float *ws, size_t ws_floats, void *stream)
{
    Plan p;
    if (!plan_for(g, &p) || !ws || ws_floats < p.frag_uint4 * 4 + p.patch_floats) return CDL_EUNSUPPORTED;
    if ((reinterpret_cast<size_t>(ws) & 15) != 0) return CDL_EUNSUPPORTED;
#define CDL_M(PH_, P_, S_)                                                  \
    if (g->Ph == PH_ && g->Pw == P_ && g->sw == S_)                         \
        return launch<PH_, P_, S_>(g, p, z, gate, w, alpha, mask, sub, out, ws, S(stream))
    CDL_M(3, 3, 1); CDL_M(5, 5, 1); CDL_M(7, 7, 1); CDL_M(9, 9, 1);
    CDL_M(3, 3, 2); CDL_M(5, 5, 2); CDL_M(7, 7, 2); CDL_M(9, 9, 2);
    CDL_M(9, 5, 1); CDL_M(9, 5, 2);
#undef CDL_M
    return CDL_EUNSUPPORTED;
}
