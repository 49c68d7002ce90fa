// Dense (many-channel to many-channel) unit-stride convolution on the matrix cores: the tier of cdl_analysis /
// cdl_synthesis for geometries with C >= 16 "image" channels -- the Conv3d(M, M, 3x3x3) of the reference's
// ResidualBlock (model/net.py:105-120) and its data gradient.  With many input channels the natural GEMM k index
// is the input channel at a fixed filter tap (the sparse-dictionary kernels of cdl_analysis_mfma.hip put the taps
// on k and gather an im2col operand; here no gather is needed):
//     out[o, p] = sum_{tap} sum_{i} W_tap[o, i] * x[i, p + tap]
// is, per tap, a (32R+l32 = o) x (k = i) by (k = i) x (l32 = pixel) MFMA product whose B operand is 8 consecutive
// channels of one pixel: one 16-byte LDS read from a [pixel][channel] bf16 plane.
//
// Workgroup (512 threads, 8 waves): a 32 x 16 tile of output pixels at one depth, MT <= 2 tiles of 32 output
// channels (more: blockIdx.y); wave w owns rows 2w, 2w+1.  The k loop runs over stages = (16-channel chunk, kd):
// each stage parks the (16+Ph-1) x (32+Pw-1) x 16-channel input window (fp32 -> bf16 hi + lo) and that stage's
// Ph*Pw weight fragments in LDS (75 KB for 3x3), then issues Ph*Pw taps x 2 rows x MT x 3 MFMAs (split-bf16:
// hi*hi + hi*lo + lo*hi, fp32 accumulate); the next stage's global loads are in flight in registers meanwhile
// (two co-resident workgroups with unpipelined staging measured 0.22 ms per 64->64 3x3x3 convolution of a
// 16 x 128 x 128 volume, 31 % of the MFMA rate).
// Epilogue straight from the accumulators (lanes l32 = 32 consecutive x: 128-byte segments):
//     v = alpha * acc;  v *= mask;  v += add [where add_gate != 0];  v -= sub;  v = ST(v, tau);  v = max(v, 0);
//     v = 0 where out_gate == 0
// which covers the analysis role (add = zin, tau), the synthesis role (transposed + flipped weights, input gate,
// mask, sub) and the ResidualBlock's fused relu.
#include "cdl_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

namespace {

constexpr int DTX = 32, DTY = 16, DNT = 512, DKC = 16;

struct DenseArgs {
    const float *x, *in_gate;
    const uint4 *frags;
    float alpha;
    const float *add, *add_gate, *mask, *sub, *tau, *out_gate;
    int relu;
    float *out;
    int N, I, O, D, H, W, Pd, Ph, Pw, tilesX, tilesY, NCC, MTT;
    CDL_DBG_FIELD(int dbg;)                                // probe build only (CDL_DENSE_DEBUG): 1 no global loads, 2 no MFMAs, 4 no epilogue
};

// frags[(((cc*Pd + kd)*taps + tap)*MTT + R)*2 + {hi,lo}][64]: lane (l32, h) holds output channel o = 32R + l32,
// input channels i = 16cc + 8h + 0..7 of W_tap.  transpose: the filters are (I, O, P) and flipped (conv-transpose).
__global__ void k_dense_prep(const float *__restrict__ w, uint4 *__restrict__ frags, int O, int I, int Pd, int Ph,
                             int Pw, int MTT, int NCC, int transpose)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int taps = Ph * Pw;
    if (t >= NCC * Pd * taps * MTT * 64) return;
    const int lane = t & 63;
    int r = t >> 6;
    const int R = r % MTT; r /= MTT;
    const int tap = r % taps; r /= taps;
    const int kd = r % Pd, cc = r / Pd;
    const int o = 32 * R + (lane & 31), h = lane >> 5, ki = tap / Pw, kj = tap % Pw;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int i = DKC * cc + 8 * h + e;
        float v = 0.0f;
        if (o < O && i < I)
            v = transpose ? w[((((size_t)i * O + o) * Pd + (Pd - 1 - kd)) * Ph + (Ph - 1 - ki)) * Pw + (Pw - 1 - kj)]
                          : w[((((size_t)o * I + i) * Pd + kd) * Ph + ki) * Pw + kj];
        const __bf16 hh = (__bf16)v;
        hi[e] = hh;
        lo[e] = (__bf16)(v - (float)hh);
    }
    const size_t base = (size_t)(t >> 6) * 2 * 64;
    frags[base + lane] = __builtin_bit_cast(uint4, hi);
    frags[base + 64 + lane] = __builtin_bit_cast(uint4, lo);
}

template <int MT>
__global__ __launch_bounds__(DNT, 4) void k_dense(DenseArgs a)   // HIP: 2nd argument = waves per SIMD -> <= 128 VGPRs, two workgroups per CU
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int XH = DTY + a.Ph - 1, XW = DTX + a.Pw - 1, taps = a.Ph * a.Pw;
    const int npix = XH * XW;
    uint4 *xh = reinterpret_cast<uint4 *>(smem);           // [2 halves][npix] 8 bf16 each (hi parts)
    uint4 *xl = xh + (size_t)npix * 2;                     // lo parts
    uint4 *wl = xl + (size_t)npix * 2;                     // [taps][MT][2][64]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l32 = lane & 31, h = lane >> 5;
    // workgroup -> tile: workgroups are dealt to the 8 XCDs round robin, so workgroup b gets the (b / 8)-th tile of
    // XCD (b % 8)'s contiguous eighth of the tile list: tiles that share halo rows and kd planes share an L2
    // (round-robin tiles read 424 MB from HBM per launch against 67 MB of input, profiles/r01_dense_pmc.json)
    const int nb = gridDim.x, xcd = blockIdx.x & 7, fl = nb >> 3, rem = nb & 7;
    int b = xcd * fl + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
    const int tx = b % a.tilesX; b /= a.tilesX;
    const int ty = b % a.tilesY; b /= a.tilesY;
    const int zd = b % a.D, n = b / a.D;
    const int r0 = blockIdx.y * MT;
    const int pd = a.Pd / 2, ph = a.Ph / 2, pw = a.Pw / 2;
    const size_t plane = (size_t)a.H * a.W, slab = (size_t)a.D * plane;

    f32x16 acc[2][MT];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int R = 0; R < MT; ++R)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[rr][R][v] = 0.0f;

    // Two workgroups share a CU (<= 128 VGPRs, 75 KB LDS each), so one's staging and epilogue run under the other's
    // MFMAs.  CDL_DENSE_DEBUG probes on a 64 -> 64 3x3x3 convolution of a 16 x 128 x 128 volume (0.22 ms): MFMAs
    // ~0.10 ms (0.07 at the peak rate), staging arithmetic + LDS writes 0.05, exposed global loads 0.05, epilogue
    // 0.05 when nothing overlaps.  Variants measured at the same 0.22-0.24 ms: one workgroup per CU with the next
    // stage's loads held in registers under the MFMAs, and that plus a software-pipelined tap loop.
    const int stages = a.NCC * a.Pd;
#pragma unroll 1
    for (int st = 0; st < stages; ++st) {
        const int cc = st / a.Pd, kd = st % a.Pd;
        const int d = zd - pd + kd;
        if (d < 0 || d >= a.D) continue;                   // uniform: a plane of zero padding
        __syncthreads();                                   // the previous stage's readers are done
        // input window: (pixel, half) items, consecutive threads on consecutive x
        for (int it = threadIdx.x; it < npix * 2; it += DNT) {
            const int half = it / npix, pix = it - half * npix;
            const int row = pix / XW, col = pix - row * XW;
            const int yy = ty * DTY - ph + row, xx = tx * DTX - pw + col;
            const int c0 = DKC * cc + 8 * half;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.0f;
            if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
                const size_t base = ((size_t)n * a.I + c0) * slab + (size_t)d * plane + (size_t)yy * a.W + xx;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (c0 + e < a.I && !CDL_DBG(a.dbg, 1)) v[e] = a.x[base + e * slab];
                if (a.in_gate) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (c0 + e < a.I && a.in_gate[base + e * slab] == 0.0f) v[e] = 0.0f;
                }
            }
            bf16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 hh = (__bf16)v[e];
                hi[e] = hh;
                lo[e] = (__bf16)(v[e] - (float)hh);
            }
            xh[half * npix + pix] = __builtin_bit_cast(uint4, hi);    // [half][pixel]: a 16-lane group of a
            xl[half * npix + pix] = __builtin_bit_cast(uint4, lo);    // ds_read_b128 covers 256 contiguous bytes
        }
        // this stage's weight fragments for channel tiles r0 .. r0+MT-1
        {   // one batch of loads, then the LDS stores (a rolled copy loop was a chain of dependent round trips)
            constexpr int NWR = 5;
            const int nW = taps * MT * 128;
            const u32x4 *fs = reinterpret_cast<const u32x4 *>(a.frags) + (size_t)st * taps * a.MTT * 128;
            u32x4 wreg[NWR];
#pragma unroll
            for (int j = 0; j < NWR; ++j) {
                const int i = (threadIdx.x + j * DNT < nW && !CDL_DBG(a.dbg, 1)) ? threadIdx.x + j * DNT : 0;
                const int tap = i / (MT * 128), rem = i - tap * (MT * 128);
                wreg[j] = fs[(unsigned)((tap * a.MTT + r0) * 128 + rem)];
            }
#pragma unroll
            for (int j = 0; j < NWR; ++j)
                if (threadIdx.x + j * DNT < nW) reinterpret_cast<u32x4 *>(wl)[threadIdx.x + j * DNT] = wreg[j];
            for (int i = threadIdx.x + NWR * DNT; i < nW; i += DNT) {      // 5 x 5 taps: the tail goes straight through
                const int tap = i / (MT * 128), rem = i - tap * (MT * 128);
                reinterpret_cast<u32x4 *>(wl)[i] = fs[(unsigned)((tap * a.MTT + r0) * 128 + rem)];
            }
        }
        __syncthreads();
        int ki = 0, kj = 0;
#pragma unroll 1
        for (int tap = 0; tap < (CDL_DBG(a.dbg, 2) ? 0 : taps); ++tap) {
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int pix = (2 * wv + rr + ki) * XW + l32 + kj;
                bh[rr] = __builtin_bit_cast(bf16x8, xh[h * npix + pix]);
                bl[rr] = __builtin_bit_cast(bf16x8, xl[h * npix + pix]);
            }
#pragma unroll
            for (int R = 0; R < MT; ++R) {
                const bf16x8 ah = __builtin_bit_cast(bf16x8, wl[(tap * MT + R) * 128 + lane]);
                const bf16x8 al = __builtin_bit_cast(bf16x8, wl[(tap * MT + R) * 128 + 64 + lane]);
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    acc[rr][R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[rr], acc[rr][R], 0, 0, 0);
                    acc[rr][R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[rr], acc[rr][R], 0, 0, 0);
                    acc[rr][R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[rr], acc[rr][R], 0, 0, 0);
                }
            }
            if (++kj == a.Pw) { kj = 0; ++ki; }
        }
    }

    // epilogue: register v of tile R is output channel 32(r0+R) + 8(v>>2) + 4h + (v&3) of pixel column l32
    const int x = tx * DTX + l32;
    const size_t nbase = (size_t)n * a.O * slab + (size_t)zd * plane;   // offsets inside the sample are 32-bit
    const float *add_n = a.add ? a.add + nbase : nullptr;
    const float *ag_n = (a.add && a.add_gate) ? a.add_gate + nbase : nullptr;
    const float *mask_n = a.mask ? a.mask + nbase : nullptr;
    const float *sub_n = a.sub ? a.sub + nbase : nullptr;
    const float *og_n = a.out_gate ? a.out_gate + nbase : nullptr;
    float *out_n = a.out + nbase;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int y = ty * DTY + 2 * wv + rr;
        if (y >= a.H || x >= a.W || CDL_DBG(a.dbg, 4)) continue;
        const int pix = y * a.W + x;
#pragma unroll
        for (int R = 0; R < MT; ++R) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = 32 * (r0 + R) + 8 * (v >> 2) + 4 * h + (v & 3);
                if (m >= a.O) continue;
                const int idx = m * (int)slab + pix;
                float val = a.alpha * acc[rr][R][v];
                if (mask_n) val *= mask_n[idx];
                if (add_n) {
                    const float av = add_n[idx];
                    if (!ag_n || ag_n[idx] != 0.0f) val += av;
                }
                if (sub_n) val -= sub_n[idx];
                if (a.tau) val = cdl_shrink(val, a.tau[n * a.O + m]);
                if (a.relu) val = fmaxf(val, 0.0f);
                if (og_n && og_n[idx] == 0.0f) val = 0.0f;
                out_n[idx] = val;
            }
        }
    }
}

struct DensePlan {
    int tilesX, tilesY, NCC, MTT, MT, ngy;
    size_t frag_uint4, lds, tiles;
};

// in: channels on the input side, out: on the output side (analysis role: C -> M; synthesis role: M -> C)
bool dense_plan(const cdl_geom *g, int in, int out, DensePlan *p)
{
    if (g->sd != 1 || g->sh != 1 || g->sw != 1) return false;
    if (in < 16 || out < 16) return false;                 // few channels: the sparse-dictionary kernels
    if (g->Ph > 5 || g->Pw > 5 || !(g->Pd & 1) || !(g->Ph & 1) || !(g->Pw & 1)) return false;
    if (g->pd != g->Pd / 2 || g->ph != g->Ph / 2 || g->pw != g->Pw / 2) return false;
    const size_t slab = (size_t)g->D * g->H * g->W;
    if ((size_t)out * slab >= ((size_t)1 << 31) || (size_t)in * slab >= ((size_t)1 << 31)) return false;   // 32-bit offsets inside one sample
    p->NCC = (in + DKC - 1) / DKC;
    p->MTT = (out + 31) / 32;
    p->MT = p->MTT >= 2 ? 2 : 1;
    p->ngy = (p->MTT + p->MT - 1) / p->MT;
    p->MTT = p->ngy * p->MT;                                // padded to whole groups
    p->tilesX = (g->W + DTX - 1) / DTX;
    p->tilesY = (g->H + DTY - 1) / DTY;
    p->tiles = (size_t)g->N * g->D * p->tilesX * p->tilesY;
    if (p->tiles >= ((size_t)1 << 31) || p->ngy > 65535) return false;
    const int taps = g->Ph * g->Pw;
    p->frag_uint4 = (size_t)p->NCC * g->Pd * taps * p->MTT * 128;
    const size_t npix = (size_t)(DTY + g->Ph - 1) * (DTX + g->Pw - 1);
    p->lds = npix * 2 * 2 * 16 + (size_t)taps * p->MT * 128 * 16;
    return p->lds <= 160 * 1024;
}

template <int MT>
int launch_dense(const DensePlan &p, const DenseArgs &a, hipStream_t st)
{
    if (p.lds > 64 * 1024) {
        const int rc = cdl_ensure_dynamic_lds((const void *)k_dense<MT>, 160 * 1024);      // per device
        if (rc) return rc;
    }
    k_dense<MT><<<dim3((unsigned)p.tiles, (unsigned)p.ngy), DNT, p.lds, st>>>(a);
    CDL_LAUNCH_CHECK();
    return 0;
}

}  // namespace

size_t cdl_dense_ws_floats(const cdl_geom *g, int transpose)
{
    DensePlan p;
    const int in = transpose ? g->M : g->C, out = transpose ? g->C : g->M;
    return dense_plan(g, in, out, &p) ? p.frag_uint4 * 4 : 0;
}

// transpose = 0: x (N,C,..) -> out (N,M,..) with filters (M,C,P) (the analysis role)
// transpose = 1: x (N,M,..) -> out (N,C,..), the conv-transpose with the same filters (the synthesis role)
// CDL_EUNSUPPORTED: the caller falls back to the other tiers.
int cdl_dense_conv(const cdl_geom *g, int transpose, const float *x, const float *in_gate, const float *w,
                   float alpha, const float *add, const float *add_gate, const float *mask, const float *sub,
                   const float *tau, int relu, const float *out_gate, float *out, float *ws, size_t ws_floats,
                   void *stream)
{
    DensePlan p;
    const int in = transpose ? g->M : g->C, outc = transpose ? g->C : g->M;
    if (!dense_plan(g, in, outc, &p)) return CDL_EUNSUPPORTED;
    if (!ws || ws_floats < p.frag_uint4 * 4 || (reinterpret_cast<size_t>(ws) & 15)) return CDL_EUNSUPPORTED;
    uint4 *frags = reinterpret_cast<uint4 *>(ws);
    const int nprep = p.NCC * g->Pd * g->Ph * g->Pw * p.MTT * 64;
    k_dense_prep<<<(nprep + 255) / 256, 256, 0, S(stream)>>>(w, frags, outc, in, g->Pd, g->Ph, g->Pw, p.MTT, p.NCC,
                                                              transpose);
    CDL_LAUNCH_CHECK();
    const DenseArgs a{x, in_gate, frags, alpha, add, add_gate, mask, sub, tau, out_gate, relu, out,
                      g->N, in, outc, g->D, g->H, g->W, g->Pd, g->Ph, g->Pw, p.tilesX, p.tilesY, p.NCC, p.MTT
                      CDL_DBG_COMMA(cdl_opts().dense_debug)};
    return p.MT == 2 ? launch_dense<2>(p, a, S(stream)) : launch_dense<1>(p, a, S(stream));
}

// ------------------------------------------------------------------------------------------------------------
// Filter gradient of the dense convolution (3x3 in-plane taps, any Pd):
//     dW[o, i, kd, ki, kj] = alpha * sum_{n,d,y,x} G[n,o,d,y,x] [gate != 0] * X[n,i,d+kd-pd,y+ki-1,x+kj-1]
// Per tap a (o) x (k = pixel) by (k = pixel) x (i) product: A = 8 consecutive pixels of one G channel, B = the same
// 8 pixels of one X channel shifted by the tap.  Workgroup (8 waves): a 4 x 32 tile of pixels at one (n, d), 64
// output x 64 input channels, the 9 in-plane taps of ONE kd (blockIdx.y); wave = (32x32 channel pair, taps 0-4 or
// 5-8), 5 accumulator tiles.  G and X tiles sit in LDS as bf16 hi/lo rows ([channel][pixel], padded strides);
// the kj = 0 / 2 operands are the aligned 16-byte row chunk funnel-shifted by one element with the neighbouring
// dword (v_alignbyte), so no shifted copies are stored.  A workgroup walks tiles blockIdx.x, +gridDim.x, ...
// accumulating in registers and writes one partial bank; k_dense_wfold adds the partials in a fixed order.
namespace {

constexpr int WTX = 32, WTY = 4, WNT = 512;
constexpr int GST = WTX * WTY + 8;           // elements per G channel row (16-byte pad: banks)
constexpr int XCOLS = WTX + 16;              // 8 halo slots either side keep every chunk 16-byte aligned
constexpr int XST = (WTY + 2) * XCOLS + 8;   // elements per X channel
constexpr int WG_TOTAL = 255;                // workgroups over all (kd, channel-group) pairs: one round of the 256 CUs

__device__ __forceinline__ void split8(const float (&v)[8], uint4 &hi, uint4 &lo)
{
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hh = (__bf16)v[e];
        a[e] = hh;
        b[e] = (__bf16)(v[e] - (float)hh);
    }
    hi = __builtin_bit_cast(uint4, a);
    lo = __builtin_bit_cast(uint4, b);
}

// the 8 elements starting one element before (kj == 0) / after (kj == 2) the aligned chunk c at element e0
__device__ __forceinline__ uint4 shifted(const __bf16 *plane, int e0, uint4 c, int kj)
{
    if (kj == 1) return c;
    if (kj == 0) {
        const unsigned p = *reinterpret_cast<const unsigned *>(plane + e0 - 2);
        return uint4{__builtin_amdgcn_alignbyte(c.x, p, 2), __builtin_amdgcn_alignbyte(c.y, c.x, 2),
                     __builtin_amdgcn_alignbyte(c.z, c.y, 2), __builtin_amdgcn_alignbyte(c.w, c.z, 2)};
    }
    const unsigned nx = *reinterpret_cast<const unsigned *>(plane + e0 + 8);
    return uint4{__builtin_amdgcn_alignbyte(c.y, c.x, 2), __builtin_amdgcn_alignbyte(c.z, c.y, 2),
                 __builtin_amdgcn_alignbyte(c.w, c.z, 2), __builtin_amdgcn_alignbyte(nx, c.w, 2)};
}

// the 8 k-steps of one tile for the taps TAP0 .. TAP0+NTAP-1 (compile-time: the operand reads of a k-step -- 2 + 2*NTAP
// 16-byte chunks and the halo dwords -- are issued together, no branch between them; with run-time taps every
// shifted operand was a branch and an exposed LDS round trip, 4x the matrix time)
template <int KJ>
__device__ __forceinline__ uint4 shifted_c(const __bf16 *plane, int e0, uint4 c)
{
    if constexpr (KJ == 1) return c;
    if constexpr (KJ == 0) {
        const unsigned p = *reinterpret_cast<const unsigned *>(plane + e0 - 2);
        return uint4{__builtin_amdgcn_alignbyte(c.x, p, 2), __builtin_amdgcn_alignbyte(c.y, c.x, 2),
                     __builtin_amdgcn_alignbyte(c.z, c.y, 2), __builtin_amdgcn_alignbyte(c.w, c.z, 2)};
    }
    const unsigned nx = *reinterpret_cast<const unsigned *>(plane + e0 + 8);
    return uint4{__builtin_amdgcn_alignbyte(c.y, c.x, 2), __builtin_amdgcn_alignbyte(c.z, c.y, 2),
                 __builtin_amdgcn_alignbyte(c.w, c.z, 2), __builtin_amdgcn_alignbyte(nx, c.w, 2)};
}

template <int TAP, int T, int TAP0, int NTAP>
__device__ __forceinline__ void read_taps(const __bf16 *xh, const __bf16 *xl, int xbase, bf16x8 (&bh)[5], bf16x8 (&bl)[5])
{
    if constexpr (T < NTAP) {
        constexpr int ki = TAP / 3, kj = TAP % 3;
        const int e0 = xbase + ki * XCOLS;
        const uint4 ch = *reinterpret_cast<const uint4 *>(xh + e0);
        const uint4 cl = *reinterpret_cast<const uint4 *>(xl + e0);
        bh[T] = __builtin_bit_cast(bf16x8, shifted_c<kj>(xh, e0, ch));
        bl[T] = __builtin_bit_cast(bf16x8, shifted_c<kj>(xl, e0, cl));
        read_taps<TAP + 1, T + 1, TAP0, NTAP>(xh, xl, xbase, bh, bl);
    }
}

template <int TAP0, int NTAP>
__device__ __forceinline__ void wgrad_tile(const __bf16 *gh, const __bf16 *gl, const __bf16 *xh, const __bf16 *xl,
                                           int Ro, int Ri, int l32, int h, f32x16 (&acc)[5])
{
#pragma unroll 1
    for (int ks = 0; ks < WTY * WTX / 16; ++ks) {
        const int r = ks >> 1, c0 = 16 * (ks & 1) + 8 * h;
        const int ga = (32 * Ro + l32) * GST + r * WTX + c0;
        const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(gh + ga));
        const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(gl + ga));
        bf16x8 bh[5], bl[5];
        read_taps<TAP0, 0, TAP0, NTAP>(xh, xl, (32 * Ri + l32) * XST + r * XCOLS + 8 + c0, bh, bl);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[t], 0, 0, 0);
    }
}

template <bool VEC>
__global__ __launch_bounds__(WNT) void k_dense_wgrad(const float *__restrict__ G, const float *__restrict__ gate,
                                                     const float *__restrict__ X, float *__restrict__ partial,
                                                     int N, int O, int I, int D, int H, int W, int Pd, int tilesX,
                                                     int tilesY, int ntiles CDL_DBG_COMMA(int dbg))
{
    extern __shared__ __align__(16) unsigned char smem[];
    __bf16 *gh = reinterpret_cast<__bf16 *>(smem);         // [64][GST]
    __bf16 *gl = gh + 64 * GST;
    __bf16 *xh = gl + 64 * GST;                            // [64][XST]
    __bf16 *xl = xh + 64 * XST;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l32 = lane & 31, h = lane >> 5;
    const int kd = blockIdx.y, o0 = 64 * blockIdx.z, pd = Pd / 2;
    // <= 32 channels on both sides: one 32 x 32 channel pair, the 9 taps dealt over the 8 waves (wave 7 takes two) --
    // the 2 x 2 pair layout would spend three quarters of its MFMAs on zero padding
    const bool narrow = I <= 32 && O - o0 <= 32;
    const int Ro = narrow ? 0 : (wv >> 1) & 1, Ri = narrow ? 0 : wv & 1;
    const int tap0 = narrow ? (wv < 7 ? wv : 7) : ((wv >> 2) ? 5 : 0);
    const int ntap = narrow ? (wv < 7 ? 1 : 2) : ((wv >> 2) ? 4 : 5);
    const size_t plane = (size_t)H * W;

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.0f;

    // The next tile's global loads are issued into registers before the current tile's MFMAs and parked in LDS
    // after them (one batch of loads in flight per tile instead of a chain of dependent round trips).
    // items per thread: 2 G chunks (8 pixels of one channel), 3 X chunks, 2 X halo elements.  Every load is
    // unconditional at a clamped address and zeroed by a select afterwards: divergent branches around the loads ended
    // in vmcnt(0) joins, which serialised them and kept them from running under the previous tile's MFMAs.
    constexpr int NG = 64 * WTY * 4 / WNT, NX = 64 * (WTY + 2) * 4 / WNT, NH = (64 * (WTY + 2) * 2 + WNT - 1) / WNT;
    float pg[NG][8], pgg[NG][8], px[NX][8], phal[NH];
    auto tile_valid = [&](int t) { const int d = (t / (tilesX * tilesY)) % D; return d + kd - pd >= 0 && d + kd - pd < D; };
    // a workgroup walks tiles blockIdx.x, +gridDim.x, ...; gridDim.x is a multiple of 8, so the Pd workgroups that
    // need the same G tile at about the same time (same blockIdx.x, different kd) sit on one XCD and share its L2
    // (contiguous tile ranges per workgroup read MORE from HBM: 928 MB against 792 MB per launch)
    auto next_tile = [&](int t) { while (t < ntiles && !tile_valid(t)) t += gridDim.x; return t; };
    auto load8 = [&](const float *base, int off, int x0, bool rowok, float (&v)[8]) {
        if constexpr (VEC) {                               // W % 4 == 0, aligned bases, x0 % 8 == 0
            const bool ok2 = rowok && x0 + 4 <= W, ok = rowok && x0 + 8 <= W;
            const float4 a = *reinterpret_cast<const float4 *>(base + (ok2 ? off : 0));
            const float4 c = *reinterpret_cast<const float4 *>(base + (ok ? off + 4 : 0));
            v[0] = ok2 ? a.x : 0.0f; v[1] = ok2 ? a.y : 0.0f; v[2] = ok2 ? a.z : 0.0f; v[3] = ok2 ? a.w : 0.0f;
            v[4] = ok ? c.x : 0.0f; v[5] = ok ? c.y : 0.0f; v[6] = ok ? c.z : 0.0f; v[7] = ok ? c.w : 0.0f;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = rowok && x0 + e < W;
                const float t = base[ok ? off + e : 0];
                v[e] = ok ? t : 0.0f;
            }
        }
    };
    auto load_tile = [&](int tile) {
        int b = tile;
        const int tx = b % tilesX; b /= tilesX;
        const int ty = b % tilesY; b /= tilesY;
        const int d = b % D, n = b / D;
        const int dz = d + kd - pd;
        // uniform bases + 32-bit offsets inside one sample (the plan bounds channels * D * H * W below 2^31)
        const float *Gt = G + ((size_t)n * O * D + d) * plane;
        const float *Tt = gate ? gate + ((size_t)n * O * D + d) * plane : nullptr;
        const float *Xt = X + ((size_t)n * I * D + dz) * plane;
        const int cstride = D * (int)plane;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int it = threadIdx.x + j * WNT;
            const int cg = it & 3, row = (it >> 2) & 3, o = it >> 4;
            const int y = ty * WTY + row, x0 = tx * WTX + cg * 8, oo = o0 + o;
            const bool rowok = oo < O && y < H;
            const int off = rowok ? oo * cstride + y * W + x0 : 0;
            load8(Gt, off, x0, rowok, pg[j]);
            if (Tt) load8(Tt, off, x0, rowok, pgg[j]);     // uniform
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) pgg[j][e] = 1.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int it = threadIdx.x + j * WNT;
            const int cg = it & 3, row = (it >> 2) % (WTY + 2), i = it / (4 * (WTY + 2));
            const int y = ty * WTY - 1 + row, x0 = tx * WTX + cg * 8;
            const bool rowok = i < I && y >= 0 && y < H;
            load8(Xt, rowok ? i * cstride + y * W + x0 : 0, x0, rowok, px[j]);
        }
#pragma unroll
        for (int j = 0; j < NH; ++j) {                     // the one element either side of every tile row
            const int it = threadIdx.x + j * WNT;
            const int side = it & 1, row = (it >> 1) % (WTY + 2), i = it / (2 * (WTY + 2));
            const int y = ty * WTY - 1 + row, xe = side ? tx * WTX + WTX : tx * WTX - 1;
            const bool ok = i < I && y >= 0 && y < H && xe >= 0 && xe < W;      // i >= 64 (items past the end): i < I fails
            const float t = Xt[ok ? i * cstride + y * W + xe : 0];
            phal[j] = ok ? t : 0.0f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int it = threadIdx.x + j * WNT;
            const int cg = it & 3, row = (it >> 2) & 3, o = it >> 4;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = pgg[j][e] == 0.0f ? 0.0f : pg[j][e];
            uint4 hi, lo;
            split8(v, hi, lo);
            *reinterpret_cast<uint4 *>(gh + o * GST + row * WTX + cg * 8) = hi;
            *reinterpret_cast<uint4 *>(gl + o * GST + row * WTX + cg * 8) = lo;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int it = threadIdx.x + j * WNT;
            const int cg = it & 3, row = (it >> 2) % (WTY + 2), i = it / (4 * (WTY + 2));
            uint4 hi, lo;
            split8(px[j], hi, lo);
            *reinterpret_cast<uint4 *>(xh + i * XST + row * XCOLS + 8 + cg * 8) = hi;
            *reinterpret_cast<uint4 *>(xl + i * XST + row * XCOLS + 8 + cg * 8) = lo;
        }
#pragma unroll
        for (int j = 0; j < NH; ++j) {
            const int it = threadIdx.x + j * WNT;
            if (it < 64 * (WTY + 2) * 2) {
                const int side = it & 1, row = (it >> 1) % (WTY + 2), i = it / (2 * (WTY + 2));
                const __bf16 hh = (__bf16)phal[j];
                const int slot = i * XST + row * XCOLS + (side ? 8 + WTX : 7);   // the other 7 halo slots are never used:
                xh[slot] = hh;                                                    // the funnel shift keeps one element
                xl[slot] = (__bf16)(phal[j] - (float)hh);
            }
        }
    };

    int tile = next_tile(blockIdx.x);
    if (tile < ntiles && !CDL_DBG(dbg, 1)) load_tile(tile);
    if CDL_DBG(dbg, 1) {
#pragma unroll
        for (int j = 0; j < NG; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) { pg[j][e] = 1.0f; pgg[j][e] = 1.0f; }
#pragma unroll
        for (int j = 0; j < NX; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) px[j][e] = 1.0f;
#pragma unroll
        for (int j = 0; j < NH; ++j) phal[j] = 1.0f;
    }
#pragma unroll 1
    while (tile < ntiles) {
        __syncthreads();                                   // the previous tile's readers are done
        if (!CDL_DBG(dbg, 4)) store_tile();
        __syncthreads();
        tile = next_tile(tile + gridDim.x);
        if (tile < ntiles && !CDL_DBG(dbg, 1)) load_tile(tile);
        if (!CDL_DBG(dbg, 2)) {
            if (narrow) {
                switch (wv) {                              // wave-uniform
                case 0: wgrad_tile<0, 1>(gh, gl, xh, xl, 0, 0, l32, h, acc); break;
                case 1: wgrad_tile<1, 1>(gh, gl, xh, xl, 0, 0, l32, h, acc); break;
                case 2: wgrad_tile<2, 1>(gh, gl, xh, xl, 0, 0, l32, h, acc); break;
                case 3: wgrad_tile<3, 1>(gh, gl, xh, xl, 0, 0, l32, h, acc); break;
                case 4: wgrad_tile<4, 1>(gh, gl, xh, xl, 0, 0, l32, h, acc); break;
                case 5: wgrad_tile<5, 1>(gh, gl, xh, xl, 0, 0, l32, h, acc); break;
                case 6: wgrad_tile<6, 1>(gh, gl, xh, xl, 0, 0, l32, h, acc); break;
                default: wgrad_tile<7, 2>(gh, gl, xh, xl, 0, 0, l32, h, acc); break;
                }
            } else if (wv >> 2) wgrad_tile<5, 4>(gh, gl, xh, xl, Ro, Ri, l32, h, acc);
            else wgrad_tile<0, 5>(gh, gl, xh, xl, Ro, Ri, l32, h, acc);
        }
    }
    // partial[((og * Pd + kd) * nwg + wg) * 9 + tap][o_local 64][i 64]
    float *dst = partial + ((size_t)(blockIdx.z * Pd + kd) * gridDim.x + blockIdx.x) * 9 * 4096;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        if (t < ntap) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int ol = 32 * Ro + 8 * (v >> 2) + 4 * h + (v & 3);
                dst[(size_t)(tap0 + t) * 4096 + ol * 64 + 32 * Ri + l32] = acc[t][v];
            }
        }
    }
}

// one 64-thread row per (o, kd, tap) and i, four rows per workgroup summing interleaved slices of the partial
// banks (a serial loop over all of them was latency bound), combined through LDS in a fixed order
__global__ __launch_bounds__(256) void k_dense_wfold(const float *__restrict__ partial, float *__restrict__ dw,
                                                     float alpha, int O, int I, int Pd, int nwg)
{
    __shared__ float part[4][64];
    const int i = threadIdx.x & 63, slice = threadIdx.x >> 6;
    int r = blockIdx.x;
    const int tap = r % 9; r /= 9;
    const int kd = r % Pd, o = r / Pd;
    const int og = o >> 6, ol = o & 63;
    const float *src = partial + ((size_t)(og * Pd + kd) * nwg * 9 + tap) * 4096 + ol * 64 + i;
    float s = 0.0f;
    for (int wg = slice; wg < nwg; wg += 4) s += src[(size_t)wg * 9 * 4096];
    part[slice][i] = s;
    __syncthreads();
    if (slice == 0 && i < I)
        dw[(((size_t)o * I + i) * Pd + kd) * 9 + tap] = alpha * (((part[0][i] + part[1][i]) + part[2][i]) + part[3][i]);
}

struct WgradPlan {
    int tilesX, tilesY, ntiles, nwg, ogroups;
    size_t ws;
};

bool wgrad_plan(const cdl_geom *g, WgradPlan *p)
{
    if (g->sd != 1 || g->sh != 1 || g->sw != 1) return false;
    if (g->Ph != 3 || g->Pw != 3 || !(g->Pd & 1)) return false;
    if (g->pd != g->Pd / 2 || g->ph != 1 || g->pw != 1) return false;
    if (g->C < 16 || g->C > 64 || g->M < 16) return false;
    const size_t vol = (size_t)g->D * g->H * g->W;
    if ((size_t)g->M * vol >= ((size_t)1 << 31) || (size_t)g->C * vol >= ((size_t)1 << 31)) return false;
    p->tilesX = (g->W + WTX - 1) / WTX;
    p->tilesY = (g->H + WTY - 1) / WTY;
    const size_t nt = (size_t)g->N * g->D * p->tilesX * p->tilesY;
    if (nt >= ((size_t)1 << 31) || nt < 8) return false;   // tiny launches: the per-filter-row kernel
    p->ntiles = (int)nt;
    p->ogroups = (g->M + 63) / 64;
    if (p->ogroups > 65535 || g->Pd > 65535) return false;
    p->nwg = WG_TOTAL / (g->Pd * p->ogroups);
    if (p->nwg >= 8) p->nwg &= ~7;                        // multiple of 8: XCD = blockIdx.x % 8 for every (kd, group)
    if (p->nwg < 1) p->nwg = 1;
    if (p->nwg > p->ntiles) p->nwg = p->ntiles;
    p->ws = (size_t)p->ogroups * g->Pd * p->nwg * 9 * 4096;
    return true;
}

}  // namespace

size_t cdl_dense_wgrad_ws_floats(const cdl_geom *g)
{
    WgradPlan p;
    return wgrad_plan(g, &p) ? p.ws : 0;
}

// dw (M,C,Pd,3,3) = alpha * correlation of F (N,M,..) [gate != 0] with x (N,C,..); CDL_EUNSUPPORTED: fall back
int cdl_dense_wgrad(const cdl_geom *g, const float *F, const float *gate, const float *x, float alpha, float *dw,
                    float *ws, size_t ws_floats, void *stream)
{
    WgradPlan p;
    if (!wgrad_plan(g, &p)) return CDL_EUNSUPPORTED;
    if (!ws || ws_floats < p.ws) return CDL_EUNSUPPORTED;
    const int vec = (g->W % 4 == 0) && !((reinterpret_cast<size_t>(F) | reinterpret_cast<size_t>(x) |
                                           reinterpret_cast<size_t>(gate)) & 15);
    const size_t lds = (size_t)(64 * GST + 64 * XST) * 2 * sizeof(__bf16);
    if (int rc = cdl_ensure_dynamic_lds((const void *)k_dense_wgrad<true>, (int)lds)) return rc;
    if (int rc = cdl_ensure_dynamic_lds((const void *)k_dense_wgrad<false>, (int)lds)) return rc;
    CDL_DBG_FIELD(const int dbg = cdl_opts().dense_debug;)
    const dim3 grid((unsigned)p.nwg, (unsigned)g->Pd, (unsigned)p.ogroups);
    if (vec)
        k_dense_wgrad<true><<<grid, WNT, lds, S(stream)>>>(F, gate, x, ws, g->N, g->M, g->C, g->D, g->H, g->W, g->Pd,
                                                           p.tilesX, p.tilesY, p.ntiles CDL_DBG_COMMA(dbg));
    else
        k_dense_wgrad<false><<<grid, WNT, lds, S(stream)>>>(F, gate, x, ws, g->N, g->M, g->C, g->D, g->H, g->W, g->Pd,
                                                            p.tilesX, p.tilesY, p.ntiles CDL_DBG_COMMA(dbg));
    CDL_LAUNCH_CHECK();
    k_dense_wfold<<<(unsigned)(g->M * g->Pd * 9), 256, 0, S(stream)>>>(ws, dw, alpha, g->M, g->C, g->Pd, p.nwg);
    CDL_LAUNCH_CHECK();
    return 0;
}
