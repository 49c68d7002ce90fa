// Fused ISTA iteration for the flagship shape family (2-D, C = 1, stride 1, P <= 7,
// M = 32 or 64) on the gfx950 matrix cores.
//
// Fusion boundary.  The iteration z' = ST(z - A(mask*B z - yp), tau) is cut where the tensor is
// THIN (one channel), not where it is fat (M channels):
//
//     launch k :  r_k (thin) , z_k (fat)  ->  z_{k+1} (fat) , partial B_{k+1} z_{k+1} (thin patches)
//
// so every fat tensor crosses HBM exactly once in and once out per iteration and needs NO halo:
// z_k[pixel] is only used pointwise, the (P-1) halo lives on the one-channel residual r_k, which is
// tiny and L2 resident.  Inside the launch, per 32-pixel row block of a wave:
//
//   analysis   acc[ch][px]  = sum_k  W_A[ch][k] * im2col(r)[k][px]        MFMA 32x32x16 bf16, K = 64
//   epilogue   z'           = ST(z -/+ acc, tau)  -> global                (z loaded in the MFMA C/D layout)
//   synthesis  col[tap][px] = sum_ch W_B^T[tap][ch] * z'[ch][px]           MFMA, z' fed straight from the
//                                                                          accumulator registers (no LDS)
//   col2im     rsum[y-3+i][x-3+j] += col[(i,j)][y,x]                       row direction summed in registers
//                                                                          across the wave's 8 rows, column
//                                                                          direction through ds_add_f32
//
// fp32-grade accuracy on bf16 matrix cores: every operand is split v = hi + lo (two bf16) and each
// product is hi*hi + hi*lo + lo*hi with fp32 accumulation (relative error ~2^-16 per product, random
// sign; measured end-to-end parity is recorded in DESIGN.md).  PREC = 1 drops the lo terms (plain bf16).
//
// Work decomposition: workgroup = 256 threads = 4 waves = 64 x 16 pixel tile of one image
// (2 x 2 waves of 32 x 8); grid = N * tilesY * tilesX.  Each workgroup writes its (16+6) x (64+6)
// partial synthesis patch; k_assemble sums the <= 4 overlapping patches per pixel in a fixed order
// (deterministic, no atomics in HBM) and applies mask / -yp.
#include "cdl_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int WX = 2, WY = 2;               // waves per workgroup along x / y
constexpr int RB = 8;                       // row blocks (image rows) per wave
constexpr int TW = 32 * WX, TH = RB * WY;   // 64 x 16 tile
constexpr int HALO = 3;                     // filters are embedded in a 7 x 7 (padded 8 x 8) tap grid
constexpr int RTW = TW + 2 * HALO, RTH = TH + 2 * HALO;   // 70 x 22 residual tile / patch
constexpr int RTC = RTW + 2;                // columns kept in LDS (col 70 is the zero-weight pad tap)
constexpr int PITCH = 28;                   // bf16 elements per LDS column (56 B: conflict-free b64 reads)
constexpr int COPY = RTC * PITCH;           // elements per shifted copy
constexpr int LDS_RT = 2 * 4 * COPY * 2;    // bytes: {hi,lo} x 4 row-shifted copies
constexpr int LDS_RSUM = 4 * RTH * RTW * 4;  // one col2im slab per wave: summed in a fixed order (deterministic)
constexpr int LDS_TAU = 64 * 4;

struct FusedParams {
    const float *r;          // (N,H,W) thin input of the analysis-like half (r_k, yp, q_{k+1} or g_xp)
    const float *zin;        // (N,M,H,W) or nullptr: z_k (forward) / du_{k+1} (backward)
    const float *gate;       // backward only: z_{k+1}, whose support gates the gradient
    float *zout;             // (N,M,H,W): z_{k+1} (forward) / du_k (backward)
    const float *tau;        // forward: (N,M) thresholds
    float *dtau;             // backward: (numWG, M) per-workgroup partial sums of -sign(z_{k+1}) * du_k
    const uint4 *frags;      // prepared weights, see k_prep
    float *patches;          // (N,tilesY,tilesX,RTH,RTW)
    float sgn;               // u = zin + sgn * acc
    int do_synth;            // 0: skip the synthesis-like half (last backward stage)
    int N, H, W, tilesX, tilesY;
};

enum { MODE_FWD = 0, MODE_FIRST = 1, MODE_BWD = 2 };

// ------------------------------------------------------------------------------------------
// Weight preparation: fp32 filters -> bf16 hi/lo MFMA A-operand fragments.
//   analysis  frag f = R*4 + ks, lane (row = ch - 32R, h), element jj:  W_A[ch][i = jj][j = 2ks + h]
//             (k = 8*j + i, so that one lane's 8 k-values are 8 consecutive ROWS of the residual)
//   synthesis frag f = Rp*(2MT) + 2R + s, lane (row = tap - 32Rp, h), element jj:
//             W_B[ch = 32R + 16s + 8(jj>>2) + 4h + (jj&3)][tap], tap = 8*i + j -- the k order in which
//             a 32x32 accumulator tile presents itself as the next MFMA's B operand.
// Output: [A hi | A lo | B hi | B lo], each frag = 64 lanes x 16 B.
__device__ __forceinline__ float embed7(const float *w, int P, int i, int j)
{
    const int off = (7 - P) / 2;
    const int ii = i - off, jj = j - off;
    if (i > 6 || j > 6 || ii < 0 || jj < 0 || ii >= P || jj >= P) return 0.0f;
    return w[ii * P + jj];
}

__global__ void k_prep(const float *__restrict__ wA, const float *__restrict__ wB,
                       uint4 *__restrict__ out, int MT, int P)
{
    const int FA = MT * 4, FB = 4 * MT;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (FA + FB) * 64) return;
    const int lane = t & 63, f = t >> 6;
    const int row = lane & 31, h = lane >> 5;
    const int PP = P * P;
    float v[8];
    if (f < FA) {
        const int R = f / 4, ks = f % 4;
        const int ch = 32 * R + row;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = embed7(wA + (size_t)ch * PP, P, jj, 2 * ks + h);
    } else {
        const int g = f - FA;
        const int Rp = g / (2 * MT), ksB = g % (2 * MT);
        const int R = ksB >> 1, s = ksB & 1;
        const int tap = 32 * Rp + row;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int ch = 32 * R + 16 * s + 8 * (jj >> 2) + 4 * h + (jj & 3);
            v[jj] = embed7(wB + (size_t)ch * PP, P, tap >> 3, tap & 7);
        }
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        __bf16 hh = (__bf16)v[jj];
        hi[jj] = hh;
        lo[jj] = (__bf16)(v[jj] - (float)hh);
    }
    uint4 *hi_dst, *lo_dst;
    if (f < FA) { hi_dst = out + (size_t)f * 64; lo_dst = out + (size_t)(FA + f) * 64; }
    else { hi_dst = out + (size_t)(2 * FA + (f - FA)) * 64; lo_dst = out + (size_t)(2 * FA + FB + (f - FA)) * 64; }
    hi_dst[lane] = __builtin_bit_cast(uint4, hi);
    lo_dst[lane] = __builtin_bit_cast(uint4, lo);
}

// ------------------------------------------------------------------------------------------
// MODE_FWD / MODE_FIRST: zout = ST(zin + sgn * A r, tau)              (net.py:85,87)
// MODE_BWD            : zout = [gate != 0] * (zin + A-like r),  dtau partials   (reverse sweep)
template <int MT, int PREC, int MODE>
__global__ __launch_bounds__(256) void k_stage(FusedParams p)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_RT + LDS_RSUM + 5 * LDS_TAU];
    __bf16 *rt = reinterpret_cast<__bf16 *>(smem);                    // [hl][q][col][PITCH]
    float *rsum_all = reinterpret_cast<float *>(smem + LDS_RT);        // [wave][RTH][RTW]
    float *tau_s = reinterpret_cast<float *>(smem + LDS_RT + LDS_RSUM);
    float *tacc_s = tau_s + 64;                                        // backward: [wave][64] dtau sums

    const int M = 32 * MT;
    int bid = blockIdx.x;
    const int txi = bid % p.tilesX; bid /= p.tilesX;
    const int tyi = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int tx0 = txi * TW, ty0 = tyi * TH;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wxi = wid % WX, wyi = wid / WX;
    const int c = lane & 31, h = lane >> 5;
    const size_t HW = (size_t)p.H * p.W;

    // ---- prologue: zero LDS, stage tau and the residual tile (4 row-shifted bf16 hi/lo copies)
    {
        uint4 *z4 = reinterpret_cast<uint4 *>(smem);
        for (int i = tid; i < (LDS_RT + LDS_RSUM) / 16; i += 256) z4[i] = make_uint4(0, 0, 0, 0);
        if (MODE == MODE_BWD) tacc_s[tid] = 0.0f;                      // 4 x 64 floats
        else if (tid < M) tau_s[tid] = p.tau[(size_t)n * M + tid];
    }
    __syncthreads();
    {
        const float *rimg = p.r + (size_t)n * HW;
        for (int i = tid; i < RTH * RTW; i += 256) {
            const int yy = i / RTW, xx = i % RTW;
            const int gy = ty0 - HALO + yy, gx = tx0 - HALO + xx;
            float v = 0.0f;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) v = rimg[(size_t)gy * p.W + gx];
            const __bf16 hh = (__bf16)v;
            const __bf16 ll = (__bf16)(v - (float)hh);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (yy - q >= 0) {
                    rt[(0 * 4 + q) * COPY + xx * PITCH + (yy - q)] = hh;
                    if (PREC == 0) rt[(1 * 4 + q) * COPY + xx * PITCH + (yy - q)] = ll;
                }
        }
    }

    // ---- weights: MFMA A operands, resident in registers for the whole tile
    constexpr int FA = MT * 4, FB = 4 * MT;
    bf16x8 wAh[MT][4], wAl[MT][4], wBh[2][2 * MT], wBl[2][2 * MT];
#pragma unroll
    for (int R = 0; R < MT; ++R)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wAh[R][ks] = __builtin_bit_cast(bf16x8, p.frags[(size_t)(R * 4 + ks) * 64 + lane]);
            if (PREC == 0) wAl[R][ks] = __builtin_bit_cast(bf16x8, p.frags[(size_t)(FA + R * 4 + ks) * 64 + lane]);
        }
#pragma unroll
    for (int Rp = 0; Rp < 2; ++Rp)
#pragma unroll
        for (int kb = 0; kb < 2 * MT; ++kb) {
            wBh[Rp][kb] = __builtin_bit_cast(bf16x8, p.frags[(size_t)(2 * FA + Rp * 2 * MT + kb) * 64 + lane]);
            if (PREC == 0)
                wBl[Rp][kb] = __builtin_bit_cast(bf16x8, p.frags[(size_t)(2 * FA + FB + Rp * 2 * MT + kb) * 64 + lane]);
        }
    __syncthreads();

    const int xl = wxi * 32 + c;             // tile-local pixel column of this lane
    const int x = tx0 + xl;
    float *rsum = rsum_all + wid * (RTH * RTW);       // this wave's slab: in-order adds of one wave only
    float ring[7][4];                        // row-direction col2im sums for halo rows yl .. yl+6, by (j & 3)
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) ring[i][j] = 0.0f;
    float tsum[MT][16];                      // backward: per-lane partial threshold gradients
#pragma unroll
    for (int R = 0; R < MT; ++R)
#pragma unroll
        for (int v = 0; v < 16; ++v) tsum[R][v] = 0.0f;

    // per-lane channel offsets of the MFMA C/D layout: register v of tile R is channel
    // 32R + 8(v>>2) + 4h + (v&3) of pixel column c
    const size_t lane_base = ((size_t)n * M + 4 * h) * HW + x;
    const bool xok = x < p.W;
    const bool has_base = (MODE == MODE_FWD) || (MODE == MODE_BWD && p.zin != nullptr);

    float zc[MT][16], gc[MT][16];            // current block's fat inputs (base, gate)
    auto load_block = [&](int b, float (&zz)[MT][16], float (&gg)[MT][16]) {
        const int y = ty0 + wyi * RB + b;
        const bool ok = xok && (y < p.H) && (b < RB);
        const size_t pix = lane_base + (size_t)y * p.W;
#pragma unroll
        for (int R = 0; R < MT; ++R)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const size_t idx = pix + (size_t)(32 * R + 8 * (v >> 2) + (v & 3)) * HW;
                zz[R][v] = (has_base && ok) ? p.zin[idx] : 0.0f;
                if (MODE == MODE_BWD) gg[R][v] = ok ? p.gate[idx] : 0.0f;
            }
    };
    if (MODE != MODE_FIRST) load_block(0, zc, gc);

#pragma unroll 1
    for (int b = 0; b < RB; ++b) {
        const int yl = wyi * RB + b;         // tile-local image row of this block
        const int y = ty0 + yl;
        const bool valid = xok && (y < p.H);

        // prefetch the next block's fat inputs (issued before this block's stores)
        float zn[MT][16], gn[MT][16];
        if (MODE != MODE_FIRST) load_block(b + 1, zn, gn);

        // -- im2col fragments of the thin input: 8 consecutive rows yl..yl+7 of column xl + j
        const int q = b & 3, e = yl - q;     // copy q is shifted up by q rows: 8-byte aligned window
        bf16x8 rh[4], rl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int col = xl + 2 * ks + h;
            const __bf16 *ph = rt + (0 * 4 + q) * COPY + col * PITCH + e;
            const bf16x4 a0 = *reinterpret_cast<const bf16x4 *>(ph);
            const bf16x4 a1 = *reinterpret_cast<const bf16x4 *>(ph + 4);
            rh[ks] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            if (PREC == 0) {
                const __bf16 *pl = rt + (1 * 4 + q) * COPY + col * PITCH + e;
                const bf16x4 b0 = *reinterpret_cast<const bf16x4 *>(pl);
                const bf16x4 b1 = *reinterpret_cast<const bf16x4 *>(pl + 4);
                rl[ks] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }

        // -- analysis-like GEMM
        f32x16 acc[MT];
#pragma unroll
        for (int R = 0; R < MT; ++R) {
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[R][v] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (PREC == 0) {
                    acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wAl[R][ks], rh[ks], acc[R], 0, 0, 0);
                    acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wAh[R][ks], rl[ks], acc[R], 0, 0, 0);
                }
                acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wAh[R][ks], rh[ks], acc[R], 0, 0, 0);
            }
        }

        // -- epilogue
        const size_t pix = lane_base + (size_t)y * p.W;
#pragma unroll
        for (int R = 0; R < MT; ++R)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int chl = 32 * R + 8 * (v >> 2) + (v & 3);        // + 4h folded into lane_base
                const size_t idx = pix + (size_t)chl * HW;
                float zz;
                if (MODE == MODE_BWD) {
                    const float gt = gc[R][v];
                    zz = gt != 0.0f ? zc[R][v] + acc[R][v] : 0.0f;
                    tsum[R][v] += gt > 0.0f ? -zz : (gt < 0.0f ? zz : 0.0f);
                    if (valid) p.zout[idx] = zz;
                } else {
                    const float base = (MODE == MODE_FWD) ? zc[R][v] : 0.0f;
                    const float u = fmaf(p.sgn, acc[R][v], base);
                    zz = cdl_shrink(u, tau_s[chl + 4 * h]);
                    if (valid) p.zout[idx] = zz; else zz = 0.0f;
                }
                acc[R][v] = zz;
            }
        if (MODE != MODE_FIRST) {
#pragma unroll
            for (int R = 0; R < MT; ++R)
#pragma unroll
                for (int v = 0; v < 16; ++v) { zc[R][v] = zn[R][v]; if (MODE == MODE_BWD) gc[R][v] = gn[R][v]; }
        }
        if (MODE == MODE_BWD && !p.do_synth) continue;

        // -- synthesis-like GEMM: the accumulator tiles are the B operand (k = channel) as they stand
        f32x16 D[2];
#pragma unroll
        for (int Rp = 0; Rp < 2; ++Rp)
#pragma unroll
            for (int v = 0; v < 16; ++v) D[Rp][v] = 0.0f;
#pragma unroll
        for (int R = 0; R < MT; ++R)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 zh, zl;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const float val = acc[R][8 * s + jj];
                    const __bf16 hh = (__bf16)val;
                    zh[jj] = hh;
                    if (PREC == 0) zl[jj] = (__bf16)(val - (float)hh);
                }
#pragma unroll
                for (int Rp = 0; Rp < 2; ++Rp) {
                    if (PREC == 0) {
                        D[Rp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wBl[Rp][2 * R + s], zh, D[Rp], 0, 0, 0);
                        D[Rp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wBh[Rp][2 * R + s], zl, D[Rp], 0, 0, 0);
                    }
                    D[Rp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wBh[Rp][2 * R + s], zh, D[Rp], 0, 0, 0);
                }
            }

        // -- col2im, row direction: tap row i = 4Rp + (v>>2) of image row y lands on halo row yl + i;
        //    ring[i] collects halo row yl + i, ring[0] is complete after this block: flush and rotate
#pragma unroll
        for (int Rp = 0; Rp < 2; ++Rp)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = 4 * Rp + (v >> 2);
                if (i <= 6) ring[i][v & 3] += D[Rp][v];
            }
        // (the two lane halves overlap in their target columns: one half per instruction keeps every
        //  add of an instruction on a distinct word, so the summation order is fixed)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int jl = 0; jl < 4; ++jl)
                if (h == hh && 4 * hh + jl <= 6) atomicAdd(&rsum[yl * RTW + xl + 4 * hh + jl], ring[0][jl]);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int jl = 0; jl < 4; ++jl) ring[i][jl] = ring[i + 1][jl];
#pragma unroll
        for (int jl = 0; jl < 4; ++jl) ring[6][jl] = 0.0f;
    }

    if (MODE == MODE_BWD) {
        // threshold-gradient partials: sum over the 32 pixel lanes of each half, then over waves
#pragma unroll
        for (int R = 0; R < MT; ++R)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float sv = tsum[R][v];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) sv += __shfl_xor(sv, off, 64);
                if (c == 0) tacc_s[wid * 64 + 32 * R + 8 * (v >> 2) + 4 * h + (v & 3)] = sv;
            }
        __syncthreads();
        if (tid < M)
            p.dtau[(size_t)blockIdx.x * M + tid] =
                (tacc_s[tid] + tacc_s[64 + tid]) + (tacc_s[128 + tid] + tacc_s[192 + tid]);
        if (!p.do_synth) return;
    }

    // ---- the 6 halo rows below the wave's last image row are still in the ring
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int jl = 0; jl < 4; ++jl)
                if (h == hh && 4 * hh + jl <= 6)
                    atomicAdd(&rsum[(wyi * RB + RB + i) * RTW + xl + 4 * hh + jl], ring[i][jl]);
    __syncthreads();
    float *patch = p.patches + ((size_t)(n * p.tilesY + tyi) * p.tilesX + txi) * (RTH * RTW);
    for (int i = tid; i < RTH * RTW; i += 256)
        patch[i] = (rsum_all[i] + rsum_all[RTH * RTW + i]) + (rsum_all[2 * RTH * RTW + i] + rsum_all[3 * RTH * RTW + i]);
}

// ------------------------------------------------------------------------------------------
// out[n,Y,X] = (mask ? mask : 1) * alpha * sum_{patches covering (Y,X)} patch - (sub ? sub : 0)
__global__ __launch_bounds__(256) void k_assemble(const float *__restrict__ patches,
                                                  const float *__restrict__ mask,
                                                  const float *__restrict__ sub, float alpha,
                                                  float *__restrict__ out, int N, int H, int W,
                                                  int tilesX, int tilesY)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)N * H * W;
    if (i >= total) return;
    const int X = i % W;
    const size_t r = i / W;
    const int Y = r % H, n = r / H;
    const int tyc = Y / TH, txc = X / TW;
    float sum = 0.0f;
    for (int ty = tyc - 1; ty <= tyc + 1; ++ty) {
        if (ty < 0 || ty >= tilesY) continue;
        const int yy = Y - (ty * TH - HALO);
        if (yy < 0 || yy >= RTH) continue;
        for (int tx = txc - 1; tx <= txc + 1; ++tx) {
            if (tx < 0 || tx >= tilesX) continue;
            const int xx = X - (tx * TW - HALO);
            if (xx < 0 || xx >= RTW) continue;
            sum += patches[(((size_t)n * tilesY + ty) * tilesX + tx) * (RTH * RTW) + yy * RTW + xx];
        }
    }
    sum *= alpha;
    if (mask) sum *= mask[i];
    if (sub) sum -= sub[i];
    out[i] = sum;
}


// ------------------------------------------------------------------------------------------
// Filter gradients on the matrix cores.
//   dw[ch][tap] = sum_px X[ch][px] * im2col(T)[tap][px]      (D = A * B with the PIXEL index as MFMA k)
// A operand = X[ch][k = 16 pixels of a row]: the fat tensor arrives with lanes = pixels (coalesced),
// so it is transposed through LDS: each lane stores its channels as packed bf16 into a [pixel][32 ch]
// image with 64-B rows (8-byte slots XOR-swizzled by (pixel>>1)&7: conflict-free stores) and the
// fragments come back through ds_read_b64_tr_b16 (conflict-free).
// B operand = im2col(T)[k = pixel][tap (i,j)] = T[y-3+i][x-3+j .. +7]: 8 consecutive columns of the
// one-channel halo tile; 4 column-shifted bf16 copies keep every such window 8-byte aligned.
// Workgroups stride over the 64 x 16 tiles, each wave accumulating all [op][ch][tap] tiles over its
// own 32 x 8 pixels; waves are then summed through LDS and one partial per workgroup is written.
constexpr int TROWS = RTH + 1;               // halo rows + the (never used) i = 7 pad row
constexpr int TPITCH = 72;                   // bf16 elements per row of a shifted copy (144 B)
constexpr int TCOPY = TROWS * TPITCH;
constexpr int WG_THIN_BYTES = 2 * 2 * 4 * TCOPY * 2;      // [op][hl][shift] copies
constexpr int IMG_ELEMS = 32 * 32;           // one [32 px][32 ch] bf16 image (2 KB)

struct WgradParams {
    const float *X[2];       // fat (N,M,H,W), nullptr = operator absent
    const float *T[2];       // thin (N,H,W)
    float *partial;          // (gridDim.x, 2, M, 64)
    int N, H, W, tilesX, tilesY, numTiles;
};

template <int MT, int PREC>
__global__ __launch_bounds__(256) void k_wgrad2d(WgradParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    __bf16 *thin = reinterpret_cast<__bf16 *>(dsm);                            // [op][hl][s][TCOPY]
    __bf16 *imgs = reinterpret_cast<__bf16 *>(dsm + WG_THIN_BYTES);            // [wave][op][R][hl][IMG_ELEMS]
    constexpr int M = 32 * MT;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wxi = wid % WX, wyi = wid / WX;
    const int c = lane & 31, h = lane >> 5;
    const size_t HW = (size_t)p.H * p.W;
    __bf16 *wimg = imgs + (size_t)wid * (2 * MT * 2) * IMG_ELEMS;

    f32x16 acc[2][MT][2];
#pragma unroll
    for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int R = 0; R < MT; ++R)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[op][R][tt][v] = 0.0f;

    // zero the thin copies once: pad elements are read (into ignored tap columns) and must stay finite
    for (int i = tid; i < WG_THIN_BYTES / 16; i += 256) reinterpret_cast<uint4 *>(dsm)[i] = make_uint4(0, 0, 0, 0);

    for (int t = blockIdx.x; t < p.numTiles; t += gridDim.x) {
        int bid = t;
        const int txi = bid % p.tilesX; bid /= p.tilesX;
        const int tyi = bid % p.tilesY;
        const int n = bid / p.tilesY;
        const int tx0 = txi * TW, ty0 = tyi * TH;
        __syncthreads();                                  // previous tile's readers are done
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            if (!p.X[op]) continue;
            const float *timg = p.T[op] + (size_t)n * HW;
            for (int i = tid; i < RTH * RTW; i += 256) {
                const int yy = i / RTW, xx = i % RTW;
                const int gy = ty0 - HALO + yy, gx = tx0 - HALO + xx;
                float v = 0.0f;
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) v = timg[(size_t)gy * p.W + gx];
                const __bf16 hh = (__bf16)v;
                const __bf16 ll = (__bf16)(v - (float)hh);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (xx - s >= 0) {
                        thin[((op * 2 + 0) * 4 + s) * TCOPY + yy * TPITCH + xx - s] = hh;
                        if (PREC == 0) thin[((op * 2 + 1) * 4 + s) * TCOPY + yy * TPITCH + xx - s] = ll;
                    }
            }
        }
        __syncthreads();

        const int x = tx0 + wxi * 32 + c;
        const bool xok = x < p.W;
#pragma unroll 1
        for (int b = 0; b < RB; ++b) {
            const int yl = wyi * RB + b, y = ty0 + yl;
            const bool valid = xok && y < p.H;
            // ---- fat operands: registers (lanes = pixels) -> bf16 hi/lo -> transposition images
#pragma unroll
            for (int op = 0; op < 2; ++op) {
                if (!p.X[op]) continue;
                const float *xp = p.X[op] + ((size_t)n * M + 4 * h) * HW + (size_t)y * p.W + x;
#pragma unroll
                for (int R = 0; R < MT; ++R)
#pragma unroll
                    for (int qv = 0; qv < 4; ++qv) {
                        bf16x4 hi4, lo4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float val = valid ? xp[(size_t)(32 * R + 8 * qv + e) * HW] : 0.0f;
                            const __bf16 hh = (__bf16)val;
                            hi4[e] = hh;
                            lo4[e] = (__bf16)(val - (float)hh);
                        }
                        const int slot = (2 * qv + h) ^ ((c >> 1) & 7);      // 8-byte slot of channels 8qv+4h..+3
                        __bf16 *dst = wimg + (size_t)((op * MT + R) * 2) * IMG_ELEMS + c * 32 + slot * 4;
                        *reinterpret_cast<bf16x4 *>(dst) = hi4;
                        if (PREC == 0) *reinterpret_cast<bf16x4 *>(dst + IMG_ELEMS) = lo4;
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int op = 0; op < 2; ++op) {
                    if (!p.X[op]) continue;
                    // B fragments: tap column of this lane, 8 consecutive pixels 16kk + 8h + (0..7)
                    bf16x8 Bh[2], Bl[2];
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int tap = 32 * tt + c, ti = tap >> 3, tj = tap & 7;
                        const int e0 = wxi * 32 + 16 * kk + 8 * h + (tj & 4);
                        const __bf16 *ph = thin + ((op * 2 + 0) * 4 + (tj & 3)) * TCOPY + (yl + ti) * TPITCH + e0;
                        const bf16x4 a0 = *reinterpret_cast<const bf16x4 *>(ph);
                        const bf16x4 a1 = *reinterpret_cast<const bf16x4 *>(ph + 4);
                        Bh[tt] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                        if (PREC == 0) {
                            const __bf16 *pl = ph + 4 * TCOPY;
                            const bf16x4 b0 = *reinterpret_cast<const bf16x4 *>(pl);
                            const bf16x4 b1 = *reinterpret_cast<const bf16x4 *>(pl + 4);
                            Bl[tt] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                    }
#pragma unroll
                    for (int R = 0; R < MT; ++R) {
                        // A fragments by transposed reads: lane (i = lane&15 of group g) gets channel
                        // 16(g&1) + i, pixels 16kk + 8h + 4*half + (0..3)
                        const int gg = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
                        bf16x8 Ah, Al;
                        {
                            bf16x4 part[2][2];
#pragma unroll
                            for (int half = 0; half < 2; ++half) {
                                const int row = 16 * kk + 8 * h + 4 * half + qq;
                                const int slot = (4 * gg + pp) ^ ((row >> 1) & 7);
                                const __bf16 *src = wimg + (size_t)((op * MT + R) * 2) * IMG_ELEMS + row * 32 + slot * 4;
                                part[0][half] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                    (__attribute__((address_space(3))) bf16x4 *)(src));
                                if (PREC == 0)
                                    part[1][half] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                        (__attribute__((address_space(3))) bf16x4 *)(src + IMG_ELEMS));
                            }
                            Ah = __builtin_shufflevector(part[0][0], part[0][1], 0, 1, 2, 3, 4, 5, 6, 7);
                            if (PREC == 0) Al = __builtin_shufflevector(part[1][0], part[1][1], 0, 1, 2, 3, 4, 5, 6, 7);
                        }
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            if (PREC == 0) {
                                acc[op][R][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh[tt], acc[op][R][tt], 0, 0, 0);
                                acc[op][R][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl[tt], acc[op][R][tt], 0, 0, 0);
                            }
                            acc[op][R][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh[tt], acc[op][R][tt], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- sum the four waves through LDS and write this workgroup's partial
    // (wave 0 stores, waves 1..3 add in turn: fixed order, every lane owns its own words)
    float *red = reinterpret_cast<float *>(dsm);                               // [2][M][64]
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wid == w) {
#pragma unroll
            for (int op = 0; op < 2; ++op)
#pragma unroll
                for (int R = 0; R < MT; ++R)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int v = 0; v < 16; ++v) {
                            const int ch = 32 * R + (v & 3) + 8 * (v >> 2) + 4 * h;
                            float *dstw = &red[(op * M + ch) * 64 + 32 * tt + c];
                            *dstw = (w == 0 ? 0.0f : *dstw) + acc[op][R][tt][v];
                        }
        }
    }
    __syncthreads();
    float *dst = p.partial + (size_t)blockIdx.x * 2 * M * 64;
    for (int i = tid; i < 2 * M * 64; i += 256) dst[i] = red[i];
}

// dw[ch][i'][j'] = alpha * sum_g partial[g][op][ch][8(i'+off) + (j'+off)].  32 outputs per workgroup,
// 8 strided partial sums each, combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ partial, int G,
                                                      float *__restrict__ dw0, float alpha0,
                                                      float *__restrict__ dw1, float alpha1, int M, int P)
{
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int t = blockIdx.x * 32 + o;
    const int per = M * P * P;
    const bool live = t < 2 * per;
    float sum = 0.0f;
    int op = 0, r = 0;
    if (live) {
        op = t / per; r = t % per;
        const int ch = r / (P * P), ij = r % (P * P), off = (7 - P) / 2;
        const int tap = 8 * (ij / P + off) + (ij % P + off);
        for (int g = part; g < G; g += 8) sum += partial[((size_t)(g * 2 + op) * M + ch) * 64 + tap];
    }
    red[part][o] = sum;
    __syncthreads();
    if (part == 0 && live) {
        float *dw = op ? dw1 : dw0;
        if (dw) {
            const float tot = ((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) +
                              ((red[4][o] + red[5][o]) + (red[6][o] + red[7][o]));
            dw[r] = (op ? alpha1 : alpha0) * tot;
        }
    }
}

// dt0[m] = sum_rows partial[row][m]; dt1[m] = sum_rows c[row / per_img] * partial[row][m]; one
// workgroup, 16 strided partial sums per channel, fixed combination order (deterministic).
__global__ __launch_bounds__(1024) void k_dtau_reduce(const float *__restrict__ partial,
                                                      const float *__restrict__ c, float *__restrict__ dt0,
                                                      float *__restrict__ dt1, int N, int per_img, int M)
{
    __shared__ float r0[16][64], r1[16][64];
    const int m = threadIdx.x & 63, part = threadIdx.x >> 6;
    float a0 = 0.0f, a1 = 0.0f;
    const int rows = N * per_img;
    if (m < M)
        for (int row = part; row < rows; row += 16) {
            const float v = partial[(size_t)row * M + m];
            a0 += v;
            if (c) a1 = fmaf(c[row / per_img], v, a1);
        }
    r0[part][m] = a0;
    r1[part][m] = a1;
    __syncthreads();
    if (part == 0 && m < M) {
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) { s0 += r0[q][m]; s1 += r1[q][m]; }
        dt0[m] = s0;
        dt1[m] = s1;
    }
}

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline bool fused_shape_ok(const cdl_geom *g)
{
    if (!cdl_geom_ok(g)) return false;
    if (g->C != 1 || g->D != 1 || g->Pd != 1 || g->sd != 1 || g->sh != 1 || g->sw != 1) return false;
    if (g->Ph != g->Pw || g->Ph > 7 || (g->Ph & 1) == 0 || g->ph != g->Ph / 2 || g->pw != g->Pw / 2) return false;
    if (g->M != 32 && g->M != 64) return false;
    return true;
}

template <int MT, int PREC>
int launch_stage(const FusedParams &p, int mode, dim3 grid, hipStream_t st)
{
    if (mode == MODE_FWD) k_stage<MT, PREC, MODE_FWD><<<grid, 256, 0, st>>>(p);
    else if (mode == MODE_FIRST) k_stage<MT, PREC, MODE_FIRST><<<grid, 256, 0, st>>>(p);
    else k_stage<MT, PREC, MODE_BWD><<<grid, 256, 0, st>>>(p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

int dispatch_stage(const cdl_geom *g, const FusedParams &p, int mode, int precision, hipStream_t st)
{
    dim3 grid((unsigned)((size_t)p.N * p.tilesX * p.tilesY));
    if (g->M == 64) return precision == 0 ? launch_stage<2, 0>(p, mode, grid, st) : launch_stage<2, 1>(p, mode, grid, st);
    return precision == 0 ? launch_stage<1, 0>(p, mode, grid, st) : launch_stage<1, 1>(p, mode, grid, st);
}

}  // namespace

extern "C" {

int cdl_fused2d_supported(const cdl_geom *g) { return fused_shape_ok(g) ? 1 : 0; }

size_t cdl_fused2d_frag_bytes(int M) { return (size_t)2 * (M / 32) * 8 * 64 * 16; }

size_t cdl_fused2d_patch_floats(const cdl_geom *g)
{
    if (!fused_shape_ok(g)) return 0;
    const size_t tilesX = (g->W + TW - 1) / TW, tilesY = (g->H + TH - 1) / TH;
    return (size_t)g->N * tilesX * tilesY * RTH * RTW;
}

int cdl_fused2d_prep(const float *wA, const float *wB, void *frags, int M, int P, void *stream)
{
    if (!wA || !wB || !frags || (M != 32 && M != 64) || P < 1 || P > 7 || !(P & 1)) return CDL_EINVAL;
    const int MT = M / 32, threads = 8 * MT * 64;
    k_prep<<<(threads + 255) / 256, 256, 0, S(stream)>>>(wA, wB, reinterpret_cast<uint4 *>(frags), MT, P);
    CDL_LAUNCH_CHECK();
    return 0;
}

int cdl_fused2d_iter_fwd(const cdl_geom *g, const float *r, const float *zin, const float *tau,
                         const void *frags, float sgn, float *zout, float *patches, int precision,
                         void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (!r || !tau || !frags || !zout || !patches || zout == zin) return CDL_EINVAL;
    if (precision != 0 && precision != 1) return CDL_EINVAL;
    FusedParams p = {};
    p.r = r; p.zin = zin; p.zout = zout; p.tau = tau;
    p.frags = reinterpret_cast<const uint4 *>(frags);
    p.patches = patches; p.sgn = sgn; p.do_synth = 1;
    p.N = g->N; p.H = g->H; p.W = g->W;
    p.tilesX = (g->W + TW - 1) / TW; p.tilesY = (g->H + TH - 1) / TH;
    return dispatch_stage(g, p, zin ? MODE_FWD : MODE_FIRST, precision, S(stream));
}

size_t cdl_fused2d_tiles(const cdl_geom *g)
{
    if (!fused_shape_ok(g)) return 0;
    return (size_t)g->N * ((g->W + TW - 1) / TW) * ((g->H + TH - 1) / TH);
}

int cdl_fused2d_stage_bwd(const cdl_geom *g, const float *thin, const float *base, const float *gate,
                          const void *frags, float *du_out, float *patches, float *dtau_partial,
                          int do_synth, int precision, void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (!thin || !gate || !frags || !du_out || !dtau_partial || du_out == base || du_out == gate) return CDL_EINVAL;
    if (do_synth && !patches) return CDL_EINVAL;
    if (precision != 0 && precision != 1) return CDL_EINVAL;
    FusedParams p = {};
    p.r = thin; p.zin = base; p.gate = gate; p.zout = du_out; p.dtau = dtau_partial;
    p.frags = reinterpret_cast<const uint4 *>(frags);
    p.patches = patches; p.sgn = 1.0f; p.do_synth = do_synth ? 1 : 0;
    p.N = g->N; p.H = g->H; p.W = g->W;
    p.tilesX = (g->W + TW - 1) / TW; p.tilesY = (g->H + TH - 1) / TH;
    return dispatch_stage(g, p, MODE_BWD, precision, S(stream));
}

int cdl_fused2d_dtau_reduce(const cdl_geom *g, const float *dtau_partial, const float *c, float *dt0,
                            float *dt1, void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (!dtau_partial || !dt0 || !dt1) return CDL_EINVAL;
    const int per_img = ((g->W + TW - 1) / TW) * ((g->H + TH - 1) / TH);
    k_dtau_reduce<<<1, 1024, 0, S(stream)>>>(dtau_partial, c, dt0, dt1, g->N, per_img, g->M);
    CDL_LAUNCH_CHECK();
    return 0;
}


static int wgrad_grid(const cdl_geom *g)
{
    const size_t tiles = (size_t)g->N * ((g->W + TW - 1) / TW) * ((g->H + TH - 1) / TH);
    return (int)(tiles < 512 ? tiles : 512);
}

size_t cdl_fused2d_wgrad_workspace_floats(const cdl_geom *g)
{
    if (!fused_shape_ok(g)) return 0;
    return (size_t)wgrad_grid(g) * 2 * g->M * 64;
}

int cdl_fused2d_wgrad(const cdl_geom *g, const float *X0, const float *T0, float alpha0, float *dw0,
                      const float *X1, const float *T1, float alpha1, float *dw1, float *workspace,
                      int precision, void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (!workspace || (!X0 && !X1)) return CDL_EINVAL;
    if ((X0 && (!T0 || !dw0)) || (X1 && (!T1 || !dw1))) return CDL_EINVAL;
    if (precision != 0 && precision != 1) return CDL_EINVAL;
    WgradParams p = {};
    p.X[0] = X0; p.T[0] = T0; p.X[1] = X1; p.T[1] = T1;
    p.partial = workspace;
    p.N = g->N; p.H = g->H; p.W = g->W;
    p.tilesX = (g->W + TW - 1) / TW; p.tilesY = (g->H + TH - 1) / TH;
    p.numTiles = p.N * p.tilesX * p.tilesY;
    const int G = wgrad_grid(g);
    const int MT = g->M / 32;
    const size_t lds = (size_t)WG_THIN_BYTES + (size_t)4 * 2 * MT * 2 * IMG_ELEMS * 2;
    const void *fn;
    if (MT == 2) fn = precision == 0 ? (const void *)k_wgrad2d<2, 0> : (const void *)k_wgrad2d<2, 1>;
    else fn = precision == 0 ? (const void *)k_wgrad2d<1, 0> : (const void *)k_wgrad2d<1, 1>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
    if (MT == 2) {
        if (precision == 0) k_wgrad2d<2, 0><<<G, 256, lds, S(stream)>>>(p);
        else k_wgrad2d<2, 1><<<G, 256, lds, S(stream)>>>(p);
    } else {
        if (precision == 0) k_wgrad2d<1, 0><<<G, 256, lds, S(stream)>>>(p);
        else k_wgrad2d<1, 1><<<G, 256, lds, S(stream)>>>(p);
    }
    CDL_LAUNCH_CHECK();
    const int total = 2 * g->M * g->Ph * g->Pw;
    k_wgrad_reduce<<<(total + 31) / 32, 256, 0, S(stream)>>>(workspace, G, X0 ? dw0 : nullptr, alpha0,
                                                               X1 ? dw1 : nullptr, alpha1, g->M, g->Ph);
    CDL_LAUNCH_CHECK();
    return 0;
}

int cdl_fused2d_assemble(const cdl_geom *g, const float *patches, const float *mask, const float *sub,
                         float alpha, float *out, void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (!patches || !out) return CDL_EINVAL;
    const size_t total = (size_t)g->N * g->H * g->W;
    k_assemble<<<(unsigned)((total + 255) / 256), 256, 0, S(stream)>>>(
        patches, mask, sub, alpha, out, g->N, g->H, g->W, (g->W + TW - 1) / TW, (g->H + TH - 1) / TH);
    CDL_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
