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
constexpr int LDS_RSUM = RTH * RTW * 4;
constexpr int LDS_TAU = 64 * 4;

struct FusedParams {
    const float *r;          // (N,H,W) residual feeding the analysis (yp for k = 0)
    const float *zin;        // (N,M,H,W) or nullptr
    float *zout;             // (N,M,H,W)
    const float *tau;        // (N,M)
    const uint4 *frags;      // prepared weights, see k_prep
    float *patches;          // (N,tilesY,tilesX,RTH,RTW)
    float sgn;               // u = zin + sgn * acc
    int N, H, W, tilesX, tilesY;
};

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
template <int MT, int PREC, bool HAS_Z>
__global__ __launch_bounds__(256) void k_iter_fwd(FusedParams p)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_RT + LDS_RSUM + LDS_TAU];
    __bf16 *rt = reinterpret_cast<__bf16 *>(smem);                    // [hl][q][col][PITCH]
    float *rsum = reinterpret_cast<float *>(smem + LDS_RT);            // [RTH][RTW]
    float *tau_s = reinterpret_cast<float *>(smem + LDS_RT + LDS_RSUM);

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
        if (tid < M) tau_s[tid] = p.tau[(size_t)n * M + tid];
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
    float ring[RB + 6][4];                   // row-direction col2im sums: [halo row - wave row0][j & 3]
#pragma unroll
    for (int i = 0; i < RB + 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) ring[i][j] = 0.0f;

#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const int yl = wyi * RB + b;         // tile-local image row of this block
        const int y = ty0 + yl;
        const bool valid = (y < p.H) && (x < p.W);

        // -- im2col fragments of the residual: 8 consecutive rows yl..yl+7 of column xl + j
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

        // -- analysis GEMM
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

        // -- epilogue: z' = ST(z + sgn*acc, tau); accumulator register v of tile R is channel
        //    32R + 8(v>>2) + 4h + (v&3) of pixel (y, x)
        const size_t pix = (size_t)y * p.W + x;
#pragma unroll
        for (int R = 0; R < MT; ++R)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int ch = 32 * R + 8 * (v >> 2) + 4 * h + (v & 3);
                const size_t idx = ((size_t)n * M + ch) * HW + pix;
                float base = 0.0f;
                if (HAS_Z && valid) base = p.zin[idx];
                const float u = fmaf(p.sgn, acc[R][v], base);
                float zz = cdl_shrink(u, tau_s[ch]);
                if (valid) p.zout[idx] = zz; else zz = 0.0f;
                acc[R][v] = zz;
            }

        // -- synthesis GEMM: the accumulator tiles are the B operand (k = channel) as they stand
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

        // -- col2im, row direction: tap row i = 4Rp + (v>>2) of image row y lands on halo row yl + i
#pragma unroll
        for (int Rp = 0; Rp < 2; ++Rp)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = 4 * Rp + (v >> 2);
                if (i <= 6) ring[b + i][v & 3] += D[Rp][v];
            }
    }

    // ---- col2im, column direction: tap column j = 4h + (v&3) lands on halo column xl + j
#pragma unroll
    for (int i = 0; i < RB + 6; ++i)
#pragma unroll
        for (int jl = 0; jl < 4; ++jl) {
            const int j = 4 * h + jl;
            if (j <= 6) atomicAdd(&rsum[(wyi * RB + i) * RTW + xl + j], ring[i][jl]);
        }
    __syncthreads();
    float *patch = p.patches + ((size_t)(n * p.tilesY + tyi) * p.tilesX + txi) * (RTH * RTW);
    for (int i = tid; i < RTH * RTW; i += 256) patch[i] = rsum[i];
}

// ------------------------------------------------------------------------------------------
// out[n,Y,X] = (mask ? mask : 1) * sum_{patches covering (Y,X)} patch - (sub ? sub : 0)
__global__ __launch_bounds__(256) void k_assemble(const float *__restrict__ patches,
                                                  const float *__restrict__ mask,
                                                  const float *__restrict__ sub,
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
    if (mask) sum *= mask[i];
    if (sub) sum -= sub[i];
    out[i] = sum;
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
int launch_iter(const FusedParams &p, bool has_z, dim3 grid, hipStream_t st)
{
    if (has_z) k_iter_fwd<MT, PREC, true><<<grid, 256, 0, st>>>(p);
    else k_iter_fwd<MT, PREC, false><<<grid, 256, 0, st>>>(p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
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
    FusedParams p;
    p.r = r; p.zin = zin; p.zout = zout; p.tau = tau;
    p.frags = reinterpret_cast<const uint4 *>(frags);
    p.patches = patches; p.sgn = sgn;
    p.N = g->N; p.H = g->H; p.W = g->W;
    p.tilesX = (g->W + TW - 1) / TW; p.tilesY = (g->H + TH - 1) / TH;
    dim3 grid((unsigned)((size_t)p.N * p.tilesX * p.tilesY));
    const bool hz = zin != nullptr;
    if (g->M == 64) return precision == 0 ? launch_iter<2, 0>(p, hz, grid, S(stream)) : launch_iter<2, 1>(p, hz, grid, S(stream));
    return precision == 0 ? launch_iter<1, 0>(p, hz, grid, S(stream)) : launch_iter<1, 1>(p, hz, grid, S(stream));
}

int cdl_fused2d_assemble(const cdl_geom *g, const float *patches, const float *mask, const float *sub,
                         float *out, void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (!patches || !out) return CDL_EINVAL;
    const size_t total = (size_t)g->N * g->H * g->W;
    k_assemble<<<(unsigned)((total + 255) / 256), 256, 0, S(stream)>>>(
        patches, mask, sub, out, g->N, g->H, g->W, (g->W + TW - 1) / TW, (g->H + TH - 1) / TH);
    CDL_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
