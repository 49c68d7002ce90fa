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
//                                                                          (7-row ring), column direction
//                                                                          through ds_add_f32
//
// fp32-grade accuracy on bf16 matrix cores: every operand is split v = hi + lo (two bf16) and each
// product is hi*hi + hi*lo + lo*hi with fp32 accumulation (relative error ~2^-16 per product, random
// sign; measured end-to-end parity is recorded in DESIGN.md).  PREC = 1 drops the lo terms (plain bf16).
//
// Work decomposition: workgroup = 256 threads = 4 waves, ONE per SIMD, = 64 x 16 pixel tile of one image (2 x 2 waves of
// 32 x 8); persistent grid, one workgroup per CU.  (Rounds 1-2: 8 waves on 64 x 32 tiles, two per SIMD so that a wave's memory
// waits were covered by its partner -- -DCDL_WY=4 still builds that form.  One wave per SIMD costs the forward nothing and the
// reverse stage 11 %, and frees the registers -- up to 512 per wave -- and the LDS that let the reverse stage accumulate
// dA_k itself: one fat pass per iteration less)  Weight fragments: staged through LDS (32 KB) once per
// workgroup, then resident in every wave's registers.  Each
// workgroup writes its (16+6) x (64+6) partial synthesis patch; k_assemble sums the <= 4 overlapping
// patches per pixel in a fixed order and applies mask / -yp.  Every reduction (col2im slabs, dtau,
// filter gradients) is combined in a fixed order: results are bit-reproducible run to run.
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <mutex>
#include <utility>
#include <vector>

#include "cdl_common.h"

// This file is compiled TWICE (csrc/Makefile).  The default object carries the C ABI; the second one (-DCDL_F2D_WY4) is the
// same code on 64 x 32 tiles with two waves per SIMD -- rounds 1-2's form -- with every entry point suffixed _wy4.  The
// whole-sweep entry points of the default object hand sweeps with bf16 code STORAGE (CDL_LAY_BLK16) to it: those kernels
// are compute-bound, and lose at one wave per SIMD what the fp32-storage sweep gains from it (28.5 against 26.5 ms/step).
#ifdef CDL_F2D_WY4
#define CDL_WY 4
#define cdl_fused2d_assemble cdl_fused2d_assemble_wy4
#define cdl_fused2d_backward cdl_fused2d_backward_wy4
#define cdl_fused2d_code_bytes cdl_fused2d_code_bytes_wy4
#define cdl_fused2d_dtau_reduce cdl_fused2d_dtau_reduce_wy4
#define cdl_fused2d_forward cdl_fused2d_forward_wy4
#define cdl_fused2d_frag_bytes cdl_fused2d_frag_bytes_wy4
#define cdl_fused2d_iter_fwd cdl_fused2d_iter_fwd_wy4
#define cdl_fused2d_map_words cdl_fused2d_map_words_wy4
#define cdl_fused2d_patch_floats cdl_fused2d_patch_floats_wy4
#define cdl_fused2d_prep cdl_fused2d_prep_wy4
#define cdl_fused2d_stage_bwd cdl_fused2d_stage_bwd_wy4
#define cdl_fused2d_stage_bwd_da cdl_fused2d_stage_bwd_da_wy4
#define cdl_fused2d_support_map cdl_fused2d_support_map_wy4
#define cdl_fused2d_supported cdl_fused2d_supported_wy4
#define cdl_fused2d_tiles cdl_fused2d_tiles_wy4
#define cdl_fused2d_timing cdl_fused2d_timing_wy4
#define cdl_fused2d_timing_read cdl_fused2d_timing_read_wy4
#define cdl_fused2d_wgrad cdl_fused2d_wgrad_wy4
#define cdl_fused2d_wgrad_workspace_floats cdl_fused2d_wgrad_workspace_floats_wy4
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifndef CDL_WY
#define CDL_WY 2                            // wave rows per workgroup: 2 (64 x 16 tiles, one wave per SIMD) or 4 (64 x 32, two)
#endif
constexpr int WX = 2, WY = CDL_WY;          // waves per workgroup along x / y
constexpr int NW = WX * WY;                 // 4 waves
constexpr int NT = 64 * NW;                 // 256 threads
constexpr int RB = 8;                       // row blocks (image rows) per wave
constexpr int TW = 32 * WX, TH = RB * WY;   // 64 x 16 tile
constexpr int HALO = 3;                     // filters are embedded in a 7 x 7 (padded 8 x 8) tap grid
constexpr int RTW = TW + 2 * HALO, RTH = TH + 2 * HALO;   // 70 x 22 residual tile / patch
constexpr int RTC = RTW + 2;                // columns kept in LDS (col 70 is the zero-weight pad tap)
constexpr int PITCH = WY == 2 ? 28 : 44;    // bf16 elements per LDS column (>= RTH + 2; 56 / 88 B: 14 / 22 dwords between the columns of
                                            // neighbouring lanes -- conflict-free b64 reads)
constexpr int COPY = RTC * PITCH;           // elements per shifted copy
constexpr int LDS_RT = 2 * 4 * COPY * 2;    // bytes: {hi,lo} x 4 row-shifted copies             (50688)
constexpr int SLAB = RTH * RTW;             // floats per col2im slab
constexpr int LDS_RSUM = 4 * SLAB * 4;      // 4 slabs by wave parity (wxi&1, wyi&1): waves sharing a slab
                                            // never touch the same word, slabs are summed in a fixed order
constexpr int LDS_W = 32 * 64 * 16;         // 32 weight fragments of 1 KB (split3, M = 64)
constexpr int LDS_TAU = 64 * 4;
constexpr int LDS_TACC = NW * 64 * 4;
constexpr int LDS_STAGE = LDS_RT + LDS_RSUM + LDS_W + LDS_TAU + LDS_TACC;

constexpr int GW_TH = 16;                    // tile rows of the filter-gradient kernel
constexpr int GW_RTH = GW_TH + 2 * HALO;     // 22
constexpr int TROWS = GW_RTH + 1;            // halo rows + the (never used) i = 7 pad row
constexpr int TPITCH = 72;                   // bf16 elements per row of a shifted copy (144 B)
constexpr int TCOPY = TROWS * TPITCH;
constexpr int WG_THIN_BYTES = 2 * 2 * 4 * TCOPY * 2;      // [op][hl][shift] copies               (52992)
constexpr int IMG_ELEMS = 32 * 32;           // one [32 px][32 ch] bf16 image (2 KB)

constexpr int LDS_DA_THIN = 2 * 4 * TCOPY * 2;           // reverse stage with dA_k: r_k as [hl][shift] copies   (26496)
constexpr int LDS_DA = LDS_DA_THIN + NW * 2 * 2 * IMG_ELEMS * 2;   // + [wave][channel tile][hl] transposition images

struct FusedParams {
    const float *r;          // (N,H,W) thin input of the analysis-like half (r_k, yp, q_{k+1} or g_xp)
    const float *zin;        // (N,M,H,W) or nullptr: z_k (forward) / du_{k+1} (backward)
    unsigned *map;           // support/sign bit planes of z_{k+1}, (N,4,H,W) words: written by the forward
                             // (nullable), read by the backward instead of the fat z_{k+1} (see k_support_map)
    float *zout;             // (N,M,H,W): z_{k+1} (forward) / du_k (backward)
    const float *tau;        // forward: (N,M) thresholds
    float *dtau;             // backward: (numWG, M) per-workgroup partial sums of -sign(z_{k+1}) * du_k
    const uint4 *frags;      // prepared weights, see k_prep
    float *patches;          // (N,tilesY,tilesX,RTH,RTW)
    float sgn;               // u = zin + sgn * acc
    int do_synth;            // 0: skip the synthesis-like half (last backward stage)
    CDL_DBG_FIELD(int dbg;)  // probe build only (CDL_FUSED_DEBUG): 1 no stores, 2 no synthesis, 32 no LDS adds, 64 no ST,
                             // 4 no analysis MFMAs, 8 no thin staging, 16 no fat loads; results are wrong
    int N, H, W, tilesX, tilesY;
    int rev;                 // walk the tiles from the last to the first (see "snake order" at the sweeps)
    const float *r2;         // reverse stage with DA: (N,H,W) thin operand of dA_k = alpha * du_k (x) im2col(r2) (r_k or yp)
    float *da_partial;       // (gridDim.x, 2, M, 64): per-workgroup partial of dA_k in operator slot 0 (k_wgrad_reduce's layout)
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

__device__ __forceinline__ void prep_one(const float *__restrict__ wA, const float *__restrict__ wB,
                                         uint4 *__restrict__ out, int MT, int P, int t)
{
    const int FA = MT * 4, FB = 4 * MT;
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

__global__ void k_prep(const float *__restrict__ wA, const float *__restrict__ wB,
                       uint4 *__restrict__ out, int MT, int P)
{
    prep_one(wA, wB, out, MT, P, blockIdx.x * blockDim.x + threadIdx.x);
}

// every (A, B) pair of a sweep in one launch: pair = blockIdx.y, its fragments at out + pair * frag_uint4
constexpr int PREP_BATCH = 32;
struct PrepBatch {
    const float *wA[PREP_BATCH];
    const float *wB[PREP_BATCH];
};
__global__ void k_prep_batch(PrepBatch b, uint4 *__restrict__ out, int frag_uint4, int MT, int P)
{
    prep_one(b.wA[blockIdx.y], b.wB[blockIdx.y], out + (size_t)blockIdx.y * frag_uint4, MT, P,
             blockIdx.x * blockDim.x + threadIdx.x);
}

// Fat tensors are addressed through buffer descriptors: the descriptor covers one image's (M,H,W)
// block, the per-lane part of the address is ONE 32-bit VGPR byte offset shared by every channel of
// the block and the channel stride goes in an SGPR -- no 64-bit per-access address registers, and an
// out-of-image lane simply carries an out-of-range offset (loads return 0, stores are dropped).
// cache-policy bits of the fat buffer accesses (0 = default; 2 = nt, streaming): build-time experiment knobs
#ifndef CDL_FAT_LD_AUX
#define CDL_FAT_LD_AUX 0
#endif
#ifndef CDL_FAT_ST_AUX
#define CDL_FAT_ST_AUX 0
#endif
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fat_rsrc(const float *base, size_t img_floats)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)(img_floats * 4), 0x00020000);
}
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, CDL_FAT_LD_AUX));
}
__device__ __forceinline__ void buf_st(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, CDL_FAT_ST_AUX);
}
constexpr int OOB = 0x7fff0000;              // byte offset beyond any descriptor range

// ---- layouts of the fat tensors ------------------------------------------------------------------------
// LAY_NCHW : the reference's (N, M, H, W) fp32.  A wave instruction of the MFMA C/D layout (lane = pixel,
//            register = channel) touches two 128-B segments in two channel planes; 32 dword accesses per
//            32-pixel row block.  What callers see (z_K, user gradients) is always this.
// LAY_BLK  : pixel-blocked fp32 [n][y][x/32][M/4][32 px][4 ch] for tensors that stay INSIDE a sweep (z_1 ..
//            z_{K-1}, du_k): the 4 consecutive channels a lane owns per register quad are 16 contiguous
//            bytes, a wave instruction covers 1 KiB contiguous, a row block 8 KiB contiguous -- 8 dwordx4
//            accesses, no shuffles.  tools/probes/probe_stream.hip (same box, pure load -> store stream of
//            the cfg2 code tensor): plane-strided dwords 5.08 TB/s, this 5.65 TB/s.
// LAY_BLK16: the same blocking with bf16 elements (8 bytes per lane): opt-in reduced-precision STORAGE
//            (half the fat bytes; the 1e-5 parity gate does not hold in this mode).
enum { LAY_NCHW = 0, LAY_BLK = 1, LAY_BLK16 = 2 };
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__host__ __device__ inline size_t blk_image_elems(int M, int H, int W) { return (size_t)H * ((W + 31) / 32) * M * 32; }
template <int LAY> __host__ __device__ inline size_t lay_image_bytes(int M, int H, int W)
{
    return LAY == LAY_NCHW ? (size_t)M * H * W * 4 : blk_image_elems(M, H, W) * (LAY == LAY_BLK ? 4 : 2);
}

// Buffer descriptor from a WAVE-UNIFORM pointer.  hipcc cannot prove uniformity of anything derived from
// threadIdx (a wave index, an operator picked per wave group): it then wraps EVERY buffer access in a
// readfirstlane "waterfall" loop, which serialises the accesses (seen in k_wgrad2d: each of the 8-32 loads of
// a row block had its own loop; with bf16 storage each was followed by s_waitcnt vmcnt(0)).  Reading the
// pointer words through readfirstlane makes the descriptor provably scalar.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void *base, size_t bytes)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void *b = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(b, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// round-to-nearest-even fp32 -> bf16 -> fp32 (what a bf16-stored code tensor holds)
__device__ __forceinline__ float bf16_round(float v) { return (float)(__bf16)v; }

// 16 consecutive-quad accesses of one accumulator tile set: registers 4q..4q+3 of tile R are channels
// 32R + 8q + 4h + (0..3) = quad 8R + 2q + h of the blocked layout; `voff` carries (y, xb, h, c), the
// scalar offset the quad pair.
// LAY_BLK16 loads kept PACKED (two bf16 per register): decoded where the values are consumed, so that the wait for
// the loads lands there and not right behind their issue (decoding at the load serialised load latency and the
// GEMM that should cover it: 0.62 against 0.44 ms for the fp32 layout)
template <int MT>
__device__ __forceinline__ void blk16_load_raw(u32x2 (&z)[MT][4], __amdgpu_buffer_rsrc_t rs, int voff)
{
#pragma unroll
    for (int R = 0; R < MT; ++R)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            z[R][q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (8 * R + 2 * q) * 32 * 8, CDL_FAT_LD_AUX));
}
__device__ __forceinline__ float blk16_decode(const u32x2 &u, int e)
{
    const unsigned w = (e & 2) ? u.y : u.x;
    return __builtin_bit_cast(float, (e & 1) ? (w & 0xffff0000u) : (w << 16));
}

template <int LAY, int MT>
__device__ __forceinline__ void blk_load(float (&z)[MT][16], __amdgpu_buffer_rsrc_t rs, int voff)
{
    constexpr int EB = LAY == LAY_BLK ? 16 : 8;
#pragma unroll
    for (int R = 0; R < MT; ++R)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int soff = (8 * R + 2 * q) * 32 * EB;
            if constexpr (LAY == LAY_BLK) {
                typedef __attribute__((ext_vector_type(4))) float f32x4;
                const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, CDL_FAT_LD_AUX));
                z[R][4 * q + 0] = f.x; z[R][4 * q + 1] = f.y; z[R][4 * q + 2] = f.z; z[R][4 * q + 3] = f.w;
            } else {
                const u32x2 u = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, CDL_FAT_LD_AUX));
                z[R][4 * q + 0] = __builtin_bit_cast(float, u.x << 16);
                z[R][4 * q + 1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
                z[R][4 * q + 2] = __builtin_bit_cast(float, u.y << 16);
                z[R][4 * q + 3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
            }
        }
}
// (values of a LAY_BLK16 store must already be bf16-representable: the caller rounds, so that the value it goes
//  on computing with is the stored one)
template <int LAY>
__device__ __forceinline__ void blk_store4(float a0, float a1, float a2, float a3, __amdgpu_buffer_rsrc_t rs, int voff, int quad_pair)
{
    constexpr int EB = LAY == LAY_BLK ? 16 : 8;
    const int soff = quad_pair * 32 * EB;
    if constexpr (LAY == LAY_BLK) {
        typedef __attribute__((ext_vector_type(4))) float f32x4;
        const f32x4 f = {a0, a1, a2, a3};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f), rs, voff, soff, CDL_FAT_ST_AUX);
    } else {
        u32x2 u;
        u.x = (__builtin_bit_cast(unsigned, a0) >> 16) | (__builtin_bit_cast(unsigned, a1) & 0xffff0000u);
        u.y = (__builtin_bit_cast(unsigned, a2) >> 16) | (__builtin_bit_cast(unsigned, a3) & 0xffff0000u);
        __builtin_amdgcn_raw_buffer_store_b64(u, rs, voff, soff, CDL_FAT_ST_AUX);
    }
}

// Sum over the 32 pixel lanes of each half-wave of N per-lane values, leaving total #i on lane
// (c with c mod N == i): a halving butterfly, N-1 exchanges instead of 5N.  Template recursion keeps
// every register index static.
template <int N>
struct LaneTransposeSum {
    static __device__ __forceinline__ float run(const float (&v)[N], int c)
    {
        constexpr int n = N / 2;
        const bool up = (c & n) != 0;
        float w[n];
#pragma unroll
        for (int k = 0; k < n; ++k) {
            const float keep = up ? v[k + n] : v[k];
            const float send = up ? v[k] : v[k + n];
            w[k] = keep + __shfl_xor(send, n, 64);
        }
        return LaneTransposeSum<n>::run(w, c);
    }
};
template <>
struct LaneTransposeSum<1> {
    static __device__ __forceinline__ float run(const float (&v)[1], int) { return v[0]; }
};
template <int N>
__device__ __forceinline__ float lane_transpose_sum(const float (&v)[N], int c)
{
    float r = LaneTransposeSum<N>::run(v, c);
    if (N == 16) r += __shfl_xor(r, 16, 64);           // lanes c and c^16 hold halves of the same index
    return r;
}

// lane i <- lane i-1 over the whole wave (lane 0 <- 0): one DPP modifier on the consuming add
__device__ __forceinline__ float wave_shr1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
// lanes 0..31 receive the value held by lanes 32..63 (lanes 32..63 receive 0).
// v_permlane32_swap vdst, src exchanges vdst[32..63] with src[0..31]; with src = 0 the new src is
// {v.hi, 0}.  Inline asm because hipcc (ROCm 7.2) returns the SAME register for both results of
// __builtin_amdgcn_permlane32_swap (tools/probes/probe_dpp.hip); s_nop 1 = the two wait states the
// instruction needs after a VALU write of either operand.
__device__ __forceinline__ float upper_half_to_lower(float v)
{
    float lo = 0.0f;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(v), "+v"(lo));
    return lo;
}
// col2im, column direction, for one halo row: lane (c, h) holds tap columns j = 4h + (0..3) of pixel
// column c; returns on lane L (0..37) the sum over j of the value of column L - j.  The upper half's
// taps are first moved onto the lower lanes, then a 6-step Horner chain of DPP wave shifts lets the
// terms that leave column 31 collect on lanes 32..37 (the overlap with the next wave's columns).
__device__ __forceinline__ float col2im_row(const float (&rv)[4], int h)
{
    const float r4 = upper_half_to_lower(rv[0]);
    const float r5 = upper_half_to_lower(rv[1]);
    const float r6 = upper_half_to_lower(rv[2]);
    const bool lo = h == 0;
    float s = lo ? r6 : 0.0f;
    s = wave_shr1(s) + (lo ? r5 : 0.0f);
    s = wave_shr1(s) + (lo ? r4 : 0.0f);
    s = wave_shr1(s) + (lo ? rv[3] : 0.0f);
    s = wave_shr1(s) + (lo ? rv[2] : 0.0f);
    s = wave_shr1(s) + (lo ? rv[1] : 0.0f);
    s = wave_shr1(s) + (lo ? rv[0] : 0.0f);
    return s;
}

// ------------------------------------------------------------------------------------------
// MODE_FWD / MODE_FIRST: zout = ST(zin + sgn * A r, tau)              (net.py:85,87)
// MODE_BWD            : zout = [z_{k+1} != 0] * (zin + A-like r),  dtau partials   (reverse sweep);
//                       support and sign of z_{k+1} come from the 2-bit map, not from the fat tensor
//   LIN / LOUT: layouts of zin and zout (LAY_*)
//   DA (reverse stage only): the filter gradient dA_k = du_k (x) im2col(r2) rides in the launch -- du_k is transposed per row
//   block through wave-private LDS images exactly as k_wgrad2d does it, and accumulated in 2 x MT more accumulator tiles
template <int MT, int PREC, int MODE, int LIN, int LOUT, bool DA = false>
__global__ __launch_bounds__(NT) void k_stage(FusedParams p)
{
    static_assert(!DA || (MODE == MODE_BWD && TH == GW_TH), "dA_k rides in the reverse stage, on k_wgrad2d's 64 x 16 tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *rt = reinterpret_cast<__bf16 *>(smem);                               // [hl][q][col][PITCH]
    float *rsum_all = reinterpret_cast<float *>(smem + LDS_RT);                   // [4][RTH][RTW]
    const uint4 *wl = reinterpret_cast<const uint4 *>(smem + LDS_RT + LDS_RSUM);  // weight fragments
    float *tau_s = reinterpret_cast<float *>(smem + LDS_RT + LDS_RSUM + LDS_W);
    float *tacc_s = tau_s + 64;                                                   // backward: [wave][64]
    __bf16 *thin2 = reinterpret_cast<__bf16 *>(smem + LDS_STAGE);                 // DA: [hl][s][TCOPY] copies of r2's tile
    __bf16 *imgs2 = reinterpret_cast<__bf16 *>(smem + LDS_STAGE + LDS_DA_THIN);   // DA: [wave][R][hl][IMG_ELEMS]

    constexpr int M = 32 * MT;
    constexpr int FA = MT * 4, FB = 4 * MT;
    constexpr int NFRAG = (PREC != 1 ? 2 : 1) * (FA + FB);   // bf16 mode stages only the hi fragments
    constexpr int NSTG = (RTH * RTW + NT - 1) / NT;          // thin-tile elements staged per thread
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wxi = wid % WX, wyi = wid / WX;
    const int c = lane & 31, h = lane >> 5;
    const size_t HW = (size_t)p.H * p.W;
    const int numTiles = p.N * p.tilesY * p.tilesX;

    // ---- once per workgroup: zero LDS, weight fragments -> LDS
    {
        uint4 *z4 = reinterpret_cast<uint4 *>(smem);
        for (int i = tid; i < (LDS_RT + LDS_RSUM) / 16; i += NT) z4[i] = make_uint4(0, 0, 0, 0);
        uint4 *wdst = reinterpret_cast<uint4 *>(smem + LDS_RT + LDS_RSUM);
        if (PREC != 1) {
            for (int i = tid; i < NFRAG * 64; i += NT) wdst[i] = p.frags[i];
        } else {                                            // [A hi | B hi] compacted
            for (int i = tid; i < FA * 64; i += NT) wdst[i] = p.frags[i];
            for (int i = tid; i < FB * 64; i += NT) wdst[FA * 64 + i] = p.frags[2 * FA * 64 + i];
        }
    }
    // thin-tile staging: registers first (so the next tile's loads fly during this tile's GEMMs), then
    // 4 row-shifted bf16 hi/lo copies in LDS.  Every in-tile word is rewritten per tile (zeros outside
    // the image); the pad words stay zero from the pass above.
    float stg[NSTG];
    auto stage_load = [&](int t) {
        int bid = p.rev ? numTiles - 1 - t : t;
        const int txi = bid % p.tilesX; bid /= p.tilesX;
        const int tyi = bid % p.tilesY;
        const int n = bid / p.tilesY;
        const float *rimg = p.r + (size_t)n * HW;
#pragma unroll
        for (int k = 0; k < NSTG; ++k) {
            const int i = tid + k * NT;
            const int yy = i / RTW, xx = i % RTW;
            const int gy = tyi * TH - HALO + yy, gx = txi * TW - HALO + xx;
            float v = 0.0f;
            if (i < RTH * RTW && t < numTiles && !CDL_DBG(p.dbg, 8) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                v = rimg[(size_t)gy * p.W + gx];
            stg[k] = v;
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int k = 0; k < NSTG; ++k) {
            const int i = tid + k * NT;
            if (i >= RTH * RTW) continue;
            const int yy = i / RTW, xx = i % RTW;
            const __bf16 hh = (__bf16)stg[k];
            const __bf16 ll = (__bf16)(stg[k] - (float)hh);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (yy - q >= 0) {
                    rt[(0 * 4 + q) * COPY + xx * PITCH + (yy - q)] = hh;
                    if (PREC != 1) rt[(1 * 4 + q) * COPY + xx * PITCH + (yy - q)] = ll;
                }
        }
    };
    // DA: the tile of r2 under this tile (22 x 70 with the halo), as 4 column-shifted bf16 hi/lo copies (k_wgrad2d's
    // format: every 8-pixel window of an im2col row 8-byte aligned); loaded a tile ahead like the thin input
    constexpr int NSTG2 = DA ? (GW_RTH * RTW + NT - 1) / NT : 1;
    float stg2[NSTG2];
    auto stage2_load = [&](int t) {
        int bid = p.rev ? numTiles - 1 - t : t;
        const int txi = bid % p.tilesX; bid /= p.tilesX;
        const int tyi = bid % p.tilesY;
        const int n = bid / p.tilesY;
        const float *timg = p.r2 + (t < numTiles ? (size_t)n * HW : 0);      // (past the last tile: nothing is read)
#pragma unroll
        for (int k = 0; k < NSTG2; ++k) {
            const int i = tid + k * NT;
            const int yy = i / RTW, xx = i % RTW;
            const int gy = tyi * TH - HALO + yy, gx = txi * TW - HALO + xx;
            const bool ok = i < GW_RTH * RTW && t < numTiles && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const float v = timg[ok ? (size_t)gy * p.W + gx : 0];
            stg2[k] = ok ? v : 0.0f;
        }
    };
    auto stage2_store = [&]() {
#pragma unroll
        for (int k = 0; k < NSTG2; ++k) {
            const int i = tid + k * NT;
            if (i >= GW_RTH * RTW) continue;
            const int yy = i / RTW, xx = i % RTW;
            const __bf16 hh = (__bf16)stg2[k];
            const __bf16 ll = (__bf16)(stg2[k] - (float)hh);
#pragma unroll
            for (int sft = 0; sft < 4; ++sft)
                if (xx - sft >= 0) {
                    thin2[(0 * 4 + sft) * TCOPY + yy * TPITCH + xx - sft] = hh;
                    if (PREC != 1) thin2[(1 * 4 + sft) * TCOPY + yy * TPITCH + xx - sft] = ll;
                }
        }
    };
    f32x16 dacc[DA ? MT : 1][2];              // DA: dA_k[channel tile][tap tile], over every tile of this workgroup
    if constexpr (DA) {
#pragma unroll
        for (int R = 0; R < MT; ++R)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int v = 0; v < 16; ++v) dacc[R][tt][v] = 0.0f;
        // pad elements of the copies are read (into ignored tap columns) and must stay finite
        for (int i = tid; i < LDS_DA_THIN / 16; i += NT) reinterpret_cast<uint4 *>(thin2)[i] = make_uint4(0, 0, 0, 0);
        stage2_load(blockIdx.x);
    }
    stage_load(blockIdx.x);
    __syncthreads();                                        // zero pass done before the first fill
    stage_store();
    if constexpr (DA) stage2_store();

    // fragment offsets inside the LDS weight area
    constexpr int OFF_AH = 0;
    constexpr int OFF_AL = FA;                               // split3 only
    constexpr int OFF_BH = (PREC != 1 ? 2 * FA : FA);
    constexpr int OFF_BL = 2 * FA + FB;                      // split3 only
    // One wave per SIMD: every weight fragment of the launch lives in this wave's registers (32 x 4 at M = 64, split3; the
    // 512-register budget has the room) -- with no partner wave on the SIMD, the LDS latency of a fragment read in front of
    // (almost) every MFMA was most of the row block's time.  Two waves per SIMD: read from LDS where used.
    constexpr bool WREG = WY == 2;
    bf16x8 wreg[WREG ? NFRAG : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) wreg[f] = __builtin_bit_cast(bf16x8, wl[f * 64 + lane]);
    }
    auto wfrag = [&](int f) __attribute__((always_inline)) {
        if constexpr (WREG) return wreg[f];
        else return __builtin_bit_cast(bf16x8, wl[f * 64 + lane]);
    };

    const int xl = wxi * 32 + c;             // tile-local pixel column of this lane
    float *rsum = rsum_all + ((wxi & 1) + 2 * (wyi & 1)) * SLAB;
    const bool has_base = (MODE == MODE_FWD) || (MODE == MODE_BWD && p.zin != nullptr);
    const int hw4 = (int)HW * 4;

#pragma unroll 1
    for (int t = blockIdx.x; t < numTiles; t += gridDim.x) {
    const int tile = p.rev ? numTiles - 1 - t : t;
    int bid = tile;
    const int txi = bid % p.tilesX; bid /= p.tilesX;
    const int tyi = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int tx0 = txi * TW, ty0 = tyi * TH;
    const int x = tx0 + xl;
    if (MODE != MODE_BWD && tid < M) tau_s[tid] = p.tau[(size_t)n * M + tid];
    __syncthreads();                         // thin copies + tau of this tile are in place
    if (MODE != MODE_BWD) stage_load(t + gridDim.x);   // next tile's thin loads fly during the GEMMs below

    float ring[7][4];                        // row-direction col2im sums for halo rows yl .. yl+6, by (j & 3)
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) ring[i][j] = 0.0f;
    float tsum[MT * 16];                     // backward: per-lane partial threshold gradients
#pragma unroll
    for (int i = 0; i < MT * 16; ++i) tsum[i] = 0.0f;
    // thresholds are >= 0 whenever project() runs (net.py:70); a negative one (3-D trainer, never projected) sends
    // the whole tile through the general shrinkage -- wave-uniform, so the common case pays 3 instructions per
    // element (u - clamp(u, -t, t)) instead of 10
    const bool tau_neg = MODE != MODE_BWD && __ballot(lane < M && !(tau_s[lane] >= 0.0f)) != 0ull;   // negative OR NaN: the general, NaN-preserving form
    float taur[MT * 16];                     // forward: this lane's 16 MT thresholds
    if (MODE != MODE_BWD) {
#pragma unroll
        for (int R = 0; R < MT; ++R)
#pragma unroll
            for (int qv = 0; qv < 4; ++qv) {
                const float4 t4 = *reinterpret_cast<const float4 *>(&tau_s[32 * R + 8 * qv + 4 * h]);
                taur[16 * R + 4 * qv + 0] = t4.x; taur[16 * R + 4 * qv + 1] = t4.y;
                taur[16 * R + 4 * qv + 2] = t4.z; taur[16 * R + 4 * qv + 3] = t4.w;
            }
    }

    // register v of accumulator tile R is channel 32R + 8(v>>2) + 4h + (v&3) of pixel column c
    const bool xok = x < p.W;
    const size_t img_in = lay_image_bytes<LIN>(M, p.H, p.W), img_out = lay_image_bytes<LOUT>(M, p.H, p.W);
    const __amdgpu_buffer_rsrc_t rs_in = uniform_rsrc(
        has_base ? (const char *)p.zin + (size_t)n * img_in : (const char *)p.zout, has_base ? img_in : 0);
    // bit 16R + v of the lane's words = register v of tile R: plane 2h holds [z != 0], plane 2h + 1 the sign bit
    unsigned *const map_n = p.map ? p.map + ((size_t)n * 4 + 2 * h) * HW : nullptr;
    const __amdgpu_buffer_rsrc_t rs_out = uniform_rsrc((const char *)p.zout + (size_t)n * img_out, img_out);
    const int lane_off = (int)((4 * h) * HW + x) * 4;
    // blocked layouts: block (y, xb) holds M/4 quads of [32 px][4 ch]; the lane's part is (h, c)
    const int XB = (p.W + 31) / 32, xb = (tx0 >> 5) + wxi;
    constexpr int EB_IN = LIN == LAY_BLK16 ? 8 : 16, EB_OUT = LOUT == LAY_BLK16 ? 8 : 16;
    const int lane_blk_in = (h * 32 + c) * EB_IN, lane_blk_out = (h * 32 + c) * EB_OUT;

#pragma unroll 1
    for (int b = 0; b < RB; ++b) {
        const int yl = wyi * RB + b;         // tile-local image row of this block
        const int y = ty0 + yl;
        const bool valid = xok && (y < p.H);
        const int voff = valid ? lane_off + y * p.W * 4 : OOB;
        const int blk = (y * XB + xb) * (M / 4) * 32;            // first [px][4 ch] slot of the block
        const int voff_in = LIN == LAY_NCHW ? voff : (valid ? blk * EB_IN + lane_blk_in : OOB);
        const int voff_st = CDL_DBG(p.dbg, 1) ? OOB : (LOUT == LAY_NCHW ? voff : (valid ? blk * EB_OUT + lane_blk_out : OOB));

        // -- fat inputs of this block, issued first: the analysis MFMAs below (and the partner
        //    wave on this SIMD) run while they are in flight
        float zc[MT][16];
        u32x2 zr[MT][4];                     // LAY_BLK16: packed, decoded in the epilogue
        if (MODE != MODE_FIRST && LIN == LAY_BLK16) {
            blk16_load_raw<MT>(zr, rs_in, voff_in);
        } else if (MODE != MODE_FIRST && CDL_DBG(p.dbg, 16)) {
#pragma unroll
            for (int R = 0; R < MT; ++R)
#pragma unroll
                for (int v = 0; v < 16; ++v) zc[R][v] = 0.25f;
        } else if (MODE != MODE_FIRST && LIN != LAY_NCHW) {
            blk_load<LIN, MT>(zc, rs_in, voff_in);
        } else if (MODE != MODE_FIRST) {
#pragma unroll
            for (int R = 0; R < MT; ++R)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int soff = (32 * R + 8 * (v >> 2) + (v & 3)) * hw4;
                    zc[R][v] = buf_ld(rs_in, voff, soff);
                }
        }
        unsigned sup = 0, sgb = 0;           // backward: support / sign words of this lane's pixel (0 outside)
        if (MODE == MODE_BWD && valid) {
            sup = map_n[(size_t)y * p.W + x];
            sgb = map_n[HW + (size_t)y * p.W + x];
        }

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
            if (PREC != 1) {
                const __bf16 *pl = rt + (1 * 4 + q) * COPY + col * PITCH + e;
                const bf16x4 b0 = *reinterpret_cast<const bf16x4 *>(pl);
                const bf16x4 b1 = *reinterpret_cast<const bf16x4 *>(pl + 4);
                rl[ks] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }

        __builtin_amdgcn_sched_barrier(0);   // phase fences: keep the scheduler from hoisting every LDS
                                             // weight read of the block to its top (register blow-up)
        // -- analysis-like GEMM (weights from LDS)
        // k-step outer, channel tile inner: the fragments of one k-step (all tiles) are read together, the products
        // alternate between the accumulator tiles, and the next k-step's fragments are read while they issue.  (Tile
        // outer / k-step inner was 12 dependent products per accumulator, each fragment read followed by a full LDS
        // wait: cdl_fusedg.hip's timeline study)
        f32x16 acc[MT];
#pragma unroll
        for (int R = 0; R < MT; ++R)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[R][v] = 0.0f;
        if CDL_DBG(p.dbg, 4) {
#pragma unroll
            for (int R = 0; R < MT; ++R) acc[R][0] = (float)rh[0][0] + (float)rl[1][1];
        } else {
            bf16x8 wh[2][MT], wlo[2][MT];
#pragma unroll
            for (int R = 0; R < MT; ++R) {
                wh[0][R] = wfrag(OFF_AH + R * 4);
                if (PREC != 1) wlo[0][R] = wfrag(OFF_AL + R * 4);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cu = ks & 1, nx = cu ^ 1;
                if (ks + 1 < 4) {
#pragma unroll
                    for (int R = 0; R < MT; ++R) {
                        wh[nx][R] = wfrag(OFF_AH + R * 4 + ks + 1);
                        if (PREC != 1) wlo[nx][R] = wfrag(OFF_AL + R * 4 + ks + 1);
                    }
                }
                if (PREC != 1) {
#pragma unroll
                    for (int R = 0; R < MT; ++R) acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[cu][R], rh[ks], acc[R], 0, 0, 0);
#pragma unroll
                    for (int R = 0; R < MT; ++R) acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cu][R], rl[ks], acc[R], 0, 0, 0);
                    if (PREC == 2) {                                     // split4: the lo * lo term as well
#pragma unroll
                        for (int R = 0; R < MT; ++R) acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[cu][R], rl[ks], acc[R], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int R = 0; R < MT; ++R) acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cu][R], rh[ks], acc[R], 0, 0, 0);
            }
        }

        __builtin_amdgcn_sched_barrier(0);
        // -- epilogue
        auto epilogue = [&](auto general_shrink) {
#pragma unroll
        for (int R = 0; R < MT; ++R)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int chl = 32 * R + 8 * (v >> 2) + (v & 3);        // + 4h folded into lane_off
                float zz;
                float zbase = 0.0f;
                if (MODE != MODE_FIRST) zbase = LIN == LAY_BLK16 ? blk16_decode(zr[R][v >> 2], v & 3) : zc[R][v];
                if (MODE == MODE_BWD) {
                    const bool on = (sup >> (16 * R + v)) & 1u;         // never for out-of-image lanes
                    zz = on ? zbase + acc[R][v] : 0.0f;
                    tsum[16 * R + v] += ((sgb >> (16 * R + v)) & 1u) ? zz : -zz;      // -sign(z') * du
                } else {
                    const float base = (MODE == MODE_FWD) ? zbase : 0.0f;
                    const float u = fmaf(p.sgn, acc[R][v], base);
                    // t >= 0: sign(u) relu(|u| - t) == u - clamp(u, -t, t) with the same rounding; NaN in u stays NaN
                    const float tt = taur[16 * R + v];
                    const float st = decltype(general_shrink)::value ? cdl_shrink(u, tt)
                                                                     : u - __builtin_amdgcn_fmed3f(u, -tt, tt);
                    zz = CDL_DBG(p.dbg, 64) ? u : (valid ? st : 0.0f);
                }
                if (LOUT == LAY_BLK16) zz = bf16_round(zz);             // the code IS its stored value from here on
                if (LOUT == LAY_NCHW) buf_st(zz, rs_out, voff_st, chl * hw4);
                acc[R][v] = zz;
            }
        };
        if (tau_neg) epilogue(std::true_type{});   // wave-uniform
        else epilogue(std::false_type{});
        if (LOUT != LAY_NCHW) {
#pragma unroll
            for (int R = 0; R < MT; ++R)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    blk_store4<LOUT>(acc[R][4 * q], acc[R][4 * q + 1], acc[R][4 * q + 2], acc[R][4 * q + 3], rs_out,
                                     voff_st, 8 * R + 2 * q);
        }
        if (MODE != MODE_BWD && map_n) {      // training forward: 2 bits per code element for the reverse sweep
            unsigned ws = 0, wg = 0;
#pragma unroll
            for (int R = 0; R < MT; ++R)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float zz = acc[R][v];
                    ws |= (zz != 0.0f ? 1u : 0u) << (16 * R + v);
                    wg |= (__builtin_bit_cast(unsigned, zz) >> 31) << (16 * R + v);
                }
            if (valid && !CDL_DBG(p.dbg, 1)) {
                map_n[(size_t)y * p.W + x] = ws;
                map_n[HW + (size_t)y * p.W + x] = wg;
            }
        }
        if constexpr (DA) {
            // dA_k += du_k (x) im2col(r2) for this row block: k_wgrad2d's product with the fat operand taken from the
            // accumulator registers (register v of tile R = channel 32R + 8(v>>2) + 4h + (v&3) of pixel column c: the
            // layout its loads have) instead of from memory
            __bf16 *wimg = imgs2 + (size_t)wid * (MT * 2) * IMG_ELEMS;
#pragma unroll
            for (int R = 0; R < MT; ++R)
#pragma unroll
                for (int qv = 0; qv < 4; ++qv) {
                    bf16x4 hi4, lo4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float val = acc[R][4 * qv + e];
                        const __bf16 hh = (__bf16)val;
                        hi4[e] = hh;
                        lo4[e] = (__bf16)(val - (float)hh);
                    }
                    const int slot = (2 * qv + h) ^ ((c >> 1) & 7);      // 8-byte slot of channels 8qv+4h..+3
                    __bf16 *dst = wimg + (size_t)(R * 2) * IMG_ELEMS + c * 32 + slot * 4;
                    *reinterpret_cast<bf16x4 *>(dst) = hi4;
                    if (PREC != 1 && LOUT != LAY_BLK16) *reinterpret_cast<bf16x4 *>(dst + IMG_ELEMS) = lo4;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 Bh[2], Bl[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int tap = 32 * tt + c, ti = tap >> 3, tj = tap & 7;
                    const int e0 = wxi * 32 + 16 * kk + 8 * h + (tj & 4);
                    const __bf16 *ph = thin2 + (0 * 4 + (tj & 3)) * TCOPY + (yl + ti) * TPITCH + e0;
                    const bf16x4 a0 = *reinterpret_cast<const bf16x4 *>(ph);
                    const bf16x4 a1 = *reinterpret_cast<const bf16x4 *>(ph + 4);
                    Bh[tt] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    if (PREC != 1) {
                        const __bf16 *pl = ph + 4 * TCOPY;
                        const bf16x4 b0 = *reinterpret_cast<const bf16x4 *>(pl);
                        const bf16x4 b1 = *reinterpret_cast<const bf16x4 *>(pl + 4);
                        Bl[tt] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
#pragma unroll
                for (int R = 0; R < MT; ++R) {
                    const int gg = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
                    bf16x4 part[2][2];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int row = 16 * kk + 8 * h + 4 * half + qq;
                        const int slot = (4 * gg + pp) ^ ((row >> 1) & 7);
                        const __bf16 *src = wimg + (size_t)(R * 2) * IMG_ELEMS + row * 32 + slot * 4;
                        part[0][half] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (__attribute__((address_space(3))) bf16x4 *)(src));
                        if (PREC != 1 && LOUT != LAY_BLK16)
                            part[1][half] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                (__attribute__((address_space(3))) bf16x4 *)(src + IMG_ELEMS));
                    }
                    const bf16x8 Ah = __builtin_shufflevector(part[0][0], part[0][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    bf16x8 Al;
                    if (PREC != 1 && LOUT != LAY_BLK16) Al = __builtin_shufflevector(part[1][0], part[1][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        if (PREC != 1) {
                            if (LOUT != LAY_BLK16) dacc[R][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh[tt], dacc[R][tt], 0, 0, 0);
                            dacc[R][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl[tt], dacc[R][tt], 0, 0, 0);
                            if (PREC == 2 && LOUT != LAY_BLK16) dacc[R][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bl[tt], dacc[R][tt], 0, 0, 0);
                        }
                        dacc[R][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh[tt], dacc[R][tt], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (MODE == MODE_BWD && !p.do_synth) continue;
        if CDL_DBG(p.dbg, 2) { ring[0][0] += acc[0][0] + acc[MT - 1][15]; continue; }

        __builtin_amdgcn_sched_barrier(0);
        // -- synthesis-like GEMM: the accumulator tiles are the B operand (k = channel) as they stand
        f32x16 D[2];
#pragma unroll
        for (int Rp = 0; Rp < 2; ++Rp)
#pragma unroll
            for (int v = 0; v < 16; ++v) D[Rp][v] = 0.0f;
        {
            // 16-channel k-steps q = 2R + s; the weight fragments of step q+1 (both tap tiles) are read while the products
            // of step q issue, and the products alternate between the two tap tiles
            bf16x8 bwh[2][2], bwl[2][2];
#pragma unroll
            for (int Rp = 0; Rp < 2; ++Rp) {
                bwh[0][Rp] = wfrag(OFF_BH + Rp * 2 * MT);
                if (PREC != 1) bwl[0][Rp] = wfrag(OFF_BL + Rp * 2 * MT);
            }
#pragma unroll
            for (int q = 0; q < 2 * MT; ++q) {
                const int R = q >> 1, s = q & 1, cu = q & 1, nx = cu ^ 1;
                if (q + 1 < 2 * MT) {
#pragma unroll
                    for (int Rp = 0; Rp < 2; ++Rp) {
                        bwh[nx][Rp] = wfrag(OFF_BH + Rp * 2 * MT + q + 1);
                        if (PREC != 1) bwl[nx][Rp] = wfrag(OFF_BL + Rp * 2 * MT + q + 1);
                    }
                }
                bf16x8 zh, zl;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const float val = acc[R][8 * s + jj];
                    const __bf16 hh = (__bf16)val;
                    zh[jj] = hh;
                    if (PREC != 1 && LOUT != LAY_BLK16) zl[jj] = (__bf16)(val - (float)hh);   // a bf16-stored code has no lo part
                }
                if (PREC != 1) {
#pragma unroll
                    for (int Rp = 0; Rp < 2; ++Rp) D[Rp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bwl[cu][Rp], zh, D[Rp], 0, 0, 0);
                    if (LOUT != LAY_BLK16) {
#pragma unroll
                        for (int Rp = 0; Rp < 2; ++Rp) D[Rp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bwh[cu][Rp], zl, D[Rp], 0, 0, 0);
                        if (PREC == 2) {
#pragma unroll
                            for (int Rp = 0; Rp < 2; ++Rp) D[Rp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bwl[cu][Rp], zl, D[Rp], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int Rp = 0; Rp < 2; ++Rp) D[Rp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bwh[cu][Rp], zh, D[Rp], 0, 0, 0);
            }
        }

        __builtin_amdgcn_sched_barrier(0);
        // -- col2im, row direction: tap row i = 4Rp + (v>>2) of image row y lands on halo row yl + i;
        //    ring[i] collects halo row yl + i, ring[0] is complete after this block: flush and rotate
#pragma unroll
        for (int Rp = 0; Rp < 2; ++Rp)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int i = 4 * Rp + (v >> 2);
                if (i <= 6) ring[i][v & 3] += D[Rp][v];
            }
        // Column direction in registers (col2im_row), then ONE plain LDS store per lane: within a slab
        // every word is written by exactly one wave, once per tile (slabs are zeroed between tiles),
        // so no atomics and no read-modify-write are needed and the result is order-independent.
        {
            const float cs = col2im_row(ring[0], h);
            if (lane < 38 && !CDL_DBG(p.dbg, 32)) rsum[yl * RTW + 32 * wxi + lane] = cs;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int jl = 0; jl < 4; ++jl) ring[i][jl] = ring[i + 1][jl];
#pragma unroll
        for (int jl = 0; jl < 4; ++jl) ring[6][jl] = 0.0f;
    }

    // ---- the 6 halo rows below the wave's last image row are still in the ring
    if (MODE != MODE_BWD || p.do_synth) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float cs = col2im_row(ring[i], h);
            if (lane < 38 && !CDL_DBG(p.dbg, 32)) rsum[(wyi * RB + RB + i) * RTW + 32 * wxi + lane] = cs;
        }
    }
    if (MODE == MODE_BWD) {
        // lane c of each half holds the wave's sum for accumulator index i = c mod (16 MT):
        // register v = i & 15 of tile R = i >> 4  ->  channel 32R + 8(v>>2) + 4h + (v&3)
        const float tacc = lane_transpose_sum(tsum, c);
        if (c < MT * 16) {
            const int R = c >> 4, v = c & 15;
            tacc_s[wid * 64 + 32 * R + 8 * (v >> 2) + 4 * h + (v & 3)] = tacc;
        }
    }
    __syncthreads();                         // every wave is done with the thin copies and the slabs
    if (MODE == MODE_BWD) stage_load(t + gridDim.x);   // (the backward stage has no registers to spare earlier)
    if constexpr (DA) stage2_load(t + gridDim.x);
    if (MODE == MODE_BWD && tid < M) {
        float sacc = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sacc += tacc_s[w * 64 + tid];
        p.dtau[(size_t)tile * M + tid] = sacc;
    }
    if (MODE != MODE_BWD || p.do_synth) {    // patch out (fixed slab order), slabs re-zeroed for the next tile
        float *patch = p.patches + (size_t)tile * SLAB;
        for (int i = tid; i < SLAB; i += NT) {
            patch[i] = (rsum_all[i] + rsum_all[SLAB + i]) + (rsum_all[2 * SLAB + i] + rsum_all[3 * SLAB + i]);
            rsum_all[i] = 0.0f; rsum_all[SLAB + i] = 0.0f; rsum_all[2 * SLAB + i] = 0.0f; rsum_all[3 * SLAB + i] = 0.0f;
        }
    }
    stage_store();                           // next tile's thin copies (registers loaded above)
    if constexpr (DA) stage2_store();
    }   // tile loop
    if constexpr (DA) {
        // sum the waves through LDS (wave 0 stores, the others add in turn: fixed order, every lane owns its words) and
        // write this workgroup's partial in k_wgrad_reduce's layout, operator slot 0
        float *red = reinterpret_cast<float *>(smem);                           // [M][64] over the thin copies
        for (int w = 0; w < NW; ++w) {
            __syncthreads();
            if (wid == w) {
#pragma unroll
                for (int R = 0; R < MT; ++R)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int v = 0; v < 16; ++v) {
                            const int ch = 32 * R + (v & 3) + 8 * (v >> 2) + 4 * h;
                            float *dstw = &red[ch * 64 + 32 * tt + c];
                            *dstw = (w == 0 ? 0.0f : *dstw) + dacc[R][tt][v];
                        }
            }
        }
        __syncthreads();
        float *dst = p.da_partial + (size_t)blockIdx.x * 2 * M * 64;
        for (int i = tid; i < M * 64; i += NT) dst[i] = red[i];
    }
}

// ------------------------------------------------------------------------------------------
// out[n,Y,X] = (mask ? mask : 1) * alpha * sum_{patches covering (Y,X)} patch - (sub ? sub : 0)
// A pixel is covered by its own tile's patch and, within HALO of a tile border, by the neighbour's across
// that border (and the diagonal one in a corner): at most 4 patches, visited in (tile row, tile column) order.
__device__ __forceinline__ float patch_sum(const float *__restrict__ patches, int n, int Y, int X,
                                           int tilesX, int tilesY)
{
    const int tyc = Y / TH, txc = X / TW;
    const int ly = Y - tyc * TH, lx = X - txc * TW;
    const int hx = (lx < HALO && txc > 0) ? -1 : ((lx >= TW - HALO && txc + 1 < tilesX) ? 1 : 0);
    const int vy = (ly < HALO && tyc > 0) ? -1 : ((ly >= TH - HALO && tyc + 1 < tilesY) ? 1 : 0);
    const float *own = patches + (((size_t)n * tilesY + tyc) * tilesX + txc) * SLAB + (ly + HALO) * RTW + lx + HALO;
    const ptrdiff_t dx = (ptrdiff_t)hx * (SLAB - TW), dy = (ptrdiff_t)vy * ((ptrdiff_t)tilesX * SLAB - TH * RTW);
    // the same pixel in the neighbour's patch: one tile over (SLAB words) and TW columns / TH rows back
    float sum = 0.0f;
    if (vy < 0) {
        if (hx < 0) sum += own[dy + dx];
        sum += own[dy];
        if (hx > 0) sum += own[dy + dx];
    }
    if (hx < 0) sum += own[dx];
    sum += own[0];
    if (hx > 0) sum += own[dx];
    if (vy > 0) {
        if (hx < 0) sum += own[dy + dx];
        sum += own[dy];
        if (hx > 0) sum += own[dy + dx];
    }
    return sum;
}

// ASM_ROWS image rows per thread: 4 for batches (fewer, fatter workgroups, 4 independent load chains), 1 for
// single small images where the launch is latency-bound and wants every workgroup it can get
template <int ASM_ROWS>
__global__ __launch_bounds__(256) void k_assemble(const float *__restrict__ patches,
                                                  const float *__restrict__ mask,
                                                  const float *__restrict__ sub, float alpha,
                                                  float *__restrict__ out, int N, int H, int W,
                                                  int tilesX, int tilesY)
{
    // grid (ceil(W/256), ceil(H/ASM_ROWS), N): no runtime divisions (TW, TH are powers of two), rows coalesced
    const int X = blockIdx.x * 256 + threadIdx.x, n = blockIdx.z;
    if (X >= W) return;
#pragma unroll
    for (int j = 0; j < ASM_ROWS; ++j) {
        const int Y = blockIdx.y * ASM_ROWS + j;
        if (Y >= H) break;
        const size_t i = ((size_t)n * H + Y) * W + X;
        float sum = alpha * patch_sum(patches, n, Y, X, tilesX, tilesY);
        if (mask) sum *= mask[i];
        if (sub) sum -= sub[i];
        out[i] = sum;
    }
}

// Batch form (W % 4 == 0): a thread owns 4 consecutive pixels of 2 rows -- the thin operands move as 16-byte vectors,
// a quarter of the threads issue half the memory instructions per pixel (same sums in the same order: bit-identical)
__global__ __launch_bounds__(256) void k_assemble_v4(const float *__restrict__ patches,
                                                     const float *__restrict__ mask, const float *__restrict__ sub,
                                                     float alpha, float *__restrict__ out, int N, int H, int W,
                                                     int tilesX, int tilesY)
{
    // 4 consecutive pixels x 2 rows per thread.  The terms of a pixel are added in patch_sum's order (row above, own
    // row, row below; left neighbour, own tile, right neighbour), but the row candidates are wave-uniform branches and
    // the 8 loads of a patch row (4 own + up to 4 from the horizontal neighbour) are issued side by side instead of as
    // per-pixel dependent chains.  X is a multiple of 4 and TW of 64: the 4 pixels share their tile.
    const int X = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, n = blockIdx.z;
    if (X >= W) return;
    const int txc = X / TW, lx = X - txc * TW;
    int hx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        hx[e] = (lx + e < HALO && txc > 0) ? -1 : ((lx + e >= TW - HALO && txc + 1 < tilesX) ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int Y = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 2 + j;
        if (Y >= H) break;
        const int tyc = Y / TH, ly = Y - tyc * TH;
        const int vy = (ly < HALO && tyc > 0) ? -1 : ((ly >= TH - HALO && tyc + 1 < tilesY) ? 1 : 0);        // uniform
        const float *own = patches + (((size_t)n * tilesY + tyc) * tilesX + txc) * SLAB + (ly + HALO) * RTW + lx + HALO;
        const ptrdiff_t dy = (ptrdiff_t)vy * ((ptrdiff_t)tilesX * SLAB - TH * RTW);
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        auto row = [&](ptrdiff_t off) {
            float a[4], b[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = own[off + e];
                b[e] = hx[e] != 0 ? own[off + e + (ptrdiff_t)hx[e] * (SLAB - TW)] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (hx[e] < 0) acc[e] += b[e];
                acc[e] += a[e];
                if (hx[e] > 0) acc[e] += b[e];
            }
        };
        if (vy < 0) row(dy);
        row(0);
        if (vy > 0) row(dy);
        const size_t i = ((size_t)n * H + Y) * W + X;
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
}

// ------------------------------------------------------------------------------------------
// Support / sign bit planes of a code tensor in the layout k_stage reads and writes them:
// map[n][2h + s][y][x], s = 0: bit (16R + v) = [z[n][ch][y][x] != 0], s = 1: its sign bit, with
// ch = 32R + 8(v>>2) + 4h + (v&3).  The training forward emits this map next to z_{k+1} (16 B per pixel
// against 256 B for M = 64), so the reverse stage reads 2 bits per element instead of the fat tensor.
// This kernel builds the same map from a tensor for callers that only hold z (tests, stepwise drivers).
__global__ __launch_bounds__(256) void k_support_map(const float *__restrict__ z, unsigned *__restrict__ map,
                                                     int N, int M, int H, int W)
{
    const size_t HW = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * 2 * HW) return;
    const size_t pix = i % HW;
    const int h = (int)((i / HW) % 2), n = (int)(i / (2 * HW));
    unsigned ws = 0, wg = 0;
    for (int R = 0; R < M / 32; ++R)
        for (int v = 0; v < 16; ++v) {
            const int ch = 32 * R + 8 * (v >> 2) + 4 * h + (v & 3);
            const float zz = z[((size_t)n * M + ch) * HW + pix];
            ws |= (zz != 0.0f ? 1u : 0u) << (16 * R + v);
            wg |= (__builtin_bit_cast(unsigned, zz) >> 31) << (16 * R + v);
        }
    map[((size_t)n * 4 + 2 * h) * HW + pix] = ws;
    map[((size_t)n * 4 + 2 * h + 1) * HW + pix] = wg;
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
// Workgroup = 8 waves: waves 0-3 reduce operator pair 0, waves 4-7 pair 1, each group covering a
// 64 x 16 pixel tile as 2 x 2 waves of 32 x 8; workgroups stride over the tiles keeping their
// [ch][tap] accumulators in registers; waves are then summed through LDS in a fixed order and one
// partial per workgroup is written.
struct WgradParams {
    const float *X[2];       // fat (N,M,H,W), nullptr = operator absent
    const float *T[2];       // thin (N,H,W)
    float *partial;          // (gridDim.x, 2, M, 64)
    int N, H, W, tilesX, tilesY, numTiles;
    int rev;                 // tiles from last to first
    int single;              // one operator (X[1] = X[0], T[1] = T[0]): both wave groups work on it, on alternate tiles; their
                             // two partial slots are added by k_wgrad_reduce
};

//   LAY: layout of the fat operands X[0], X[1] (LAY_*)
template <int MT, int PREC, int LAY>
__global__ __launch_bounds__(512) void k_wgrad2d(WgradParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    __bf16 *thin = reinterpret_cast<__bf16 *>(dsm);                            // [op][hl][s][TCOPY]
    __bf16 *imgs = reinterpret_cast<__bf16 *>(dsm + WG_THIN_BYTES);            // [wave][R][hl][IMG_ELEMS]
    constexpr int M = 32 * MT;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int op = wid >> 2, wq = wid & 3;                                     // operator pair, wave in group
    const int wxi = wq & 1, wyi = wq >> 1;
    const int c = lane & 31, h = lane >> 5;
    const size_t HW = (size_t)p.H * p.W;
    __bf16 *wimg = imgs + (size_t)wid * (MT * 2) * IMG_ELEMS;
    f32x16 acc[MT][2];
#pragma unroll
    for (int R = 0; R < MT; ++R)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[R][tt][v] = 0.0f;

    // zero the thin copies once: pad elements are read (into ignored tap columns) and must stay finite
    for (int i = tid; i < WG_THIN_BYTES / 16; i += 512) reinterpret_cast<uint4 *>(dsm)[i] = make_uint4(0, 0, 0, 0);

    const bool present = p.X[op] != nullptr;
    const int nsteps = p.single ? (p.numTiles + 1) / 2 : p.numTiles;
    for (int t0 = blockIdx.x; t0 < nsteps; t0 += gridDim.x) {
        const int t = p.single ? 2 * t0 + op : t0;         // single operator: the wave groups take alternate tiles
        const bool active = present && t < p.numTiles;    // (per wave group)
        int bid = p.rev ? p.numTiles - 1 - t : t;
        const int txi = bid % p.tilesX; bid /= p.tilesX;
        const int tyi = bid % p.tilesY;
        const int n = active ? bid / p.tilesY : 0;
        const int tx0 = txi * TW, ty0 = tyi * GW_TH;
        __syncthreads();                                  // previous tile's readers are done
        if (active) {                                     // each group of 256 threads stages its own thin tile
            const float *timg = p.T[op] + (size_t)n * HW;
            // all of a thread's elements are loaded (clamped addresses, no branches) before any is converted: one at a time
            // was seven dependent global-load latencies per 64 x 16 tile, a quarter of the single-operator launch
            constexpr int NSW = (GW_RTH * RTW + 255) / 256;
            float tv[NSW];
#pragma unroll
            for (int k = 0; k < NSW; ++k) {
                const int i = (tid & 255) + k * 256;
                const int yy = i / RTW, xx = i % RTW;
                const int gy = ty0 - HALO + yy, gx = tx0 - HALO + xx;
                const bool ok = i < GW_RTH * RTW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                const float v = timg[ok ? (size_t)gy * p.W + gx : 0];
                tv[k] = ok ? v : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < NSW; ++k) {
                const int i = (tid & 255) + k * 256;
                if (i >= GW_RTH * RTW) continue;
                const int yy = i / RTW, xx = i % RTW;
                const __bf16 hh = (__bf16)tv[k];
                const __bf16 ll = (__bf16)(tv[k] - (float)hh);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (xx - s >= 0) {
                        thin[((op * 2 + 0) * 4 + s) * TCOPY + yy * TPITCH + xx - s] = hh;
                        if (PREC != 1) thin[((op * 2 + 1) * 4 + s) * TCOPY + yy * TPITCH + xx - s] = ll;
                    }
            }
        }
        __syncthreads();
        if (!active) continue;

        const int x = tx0 + wxi * 32 + c;
        const bool xok = x < p.W;
#pragma unroll 1
        for (int b = 0; b < RB; ++b) {
            const int yl = wyi * RB + b, y = ty0 + yl;
            const bool valid = xok && y < p.H;
            // ---- fat operand: registers (lanes = pixels) -> bf16 hi/lo -> transposition images
            const size_t img_b = lay_image_bytes<LAY>(M, p.H, p.W);
            const __amdgpu_buffer_rsrc_t rs = uniform_rsrc((const char *)p.X[op] + (size_t)n * img_b, img_b);   // op: per wave group
            const int voff = valid ? (int)((4 * h) * HW + (size_t)y * p.W + x) * 4 : OOB;
            const int hw4 = (int)HW * 4;
            float xv[MT][16];
            u32x2 xr[MT][4];                 // LAY_BLK16: the stored bf16 pairs ARE the hi fragments; there is no lo part
            if (LAY == LAY_BLK16) {
                const int XB = (p.W + 31) / 32, xb = (tx0 >> 5) + wxi;
                blk16_load_raw<MT>(xr, rs, valid ? ((y * XB + xb) * (M / 4) * 32 + h * 32 + c) * 8 : OOB);
            } else if (LAY == LAY_BLK) {
                const int XB = (p.W + 31) / 32, xb = (tx0 >> 5) + wxi;
                blk_load<LAY_BLK, MT>(xv, rs, valid ? ((y * XB + xb) * (M / 4) * 32 + h * 32 + c) * 16 : OOB);
            } else {
#pragma unroll
                for (int R = 0; R < MT; ++R)
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        xv[R][v] = buf_ld(rs, voff, (32 * R + 8 * (v >> 2) + (v & 3)) * hw4);
            }
#pragma unroll
            for (int R = 0; R < MT; ++R)
#pragma unroll
                for (int qv = 0; qv < 4; ++qv) {
                    bf16x4 hi4, lo4;
                    if (LAY == LAY_BLK16) {
                        hi4 = __builtin_bit_cast(bf16x4, xr[R][qv]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float val = xv[R][4 * qv + e];
                            const __bf16 hh = (__bf16)val;
                            hi4[e] = hh;
                            lo4[e] = (__bf16)(val - (float)hh);
                        }
                    }
                    const int slot = (2 * qv + h) ^ ((c >> 1) & 7);      // 8-byte slot of channels 8qv+4h..+3
                    __bf16 *dst = wimg + (size_t)(R * 2) * IMG_ELEMS + c * 32 + slot * 4;
                    *reinterpret_cast<bf16x4 *>(dst) = hi4;
                    if (PREC != 1 && LAY != LAY_BLK16) *reinterpret_cast<bf16x4 *>(dst + IMG_ELEMS) = lo4;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
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
                    if (PREC != 1) {
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
                    bf16x4 part[2][2];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int row = 16 * kk + 8 * h + 4 * half + qq;
                        const int slot = (4 * gg + pp) ^ ((row >> 1) & 7);
                        const __bf16 *src = wimg + (size_t)(R * 2) * IMG_ELEMS + row * 32 + slot * 4;
                        part[0][half] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (__attribute__((address_space(3))) bf16x4 *)(src));
                        if (PREC != 1 && LAY != LAY_BLK16)
                            part[1][half] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                (__attribute__((address_space(3))) bf16x4 *)(src + IMG_ELEMS));
                    }
                    const bf16x8 Ah = __builtin_shufflevector(part[0][0], part[0][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    bf16x8 Al;
                    if (PREC != 1 && LAY != LAY_BLK16) Al = __builtin_shufflevector(part[1][0], part[1][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        if (PREC != 1) {
                            if (LAY != LAY_BLK16) acc[R][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh[tt], acc[R][tt], 0, 0, 0);
                            acc[R][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl[tt], acc[R][tt], 0, 0, 0);
                            if (PREC == 2 && LAY != LAY_BLK16) acc[R][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bl[tt], acc[R][tt], 0, 0, 0);
                        }
                        acc[R][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh[tt], acc[R][tt], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- sum the four waves of each group through LDS (wave 0 stores, 1..3 add in turn: fixed
    //      order, every lane owns its own words) and write this workgroup's partial
    float *red = reinterpret_cast<float *>(dsm);                               // [2][M][64]
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wq == w) {
#pragma unroll
            for (int R = 0; R < MT; ++R)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int ch = 32 * R + (v & 3) + 8 * (v >> 2) + 4 * h;
                        float *dstw = &red[(op * M + ch) * 64 + 32 * tt + c];
                        *dstw = (w == 0 ? 0.0f : *dstw) + acc[R][tt][v];
                    }
        }
    }
    __syncthreads();
    float *dst = p.partial + (size_t)blockIdx.x * 2 * M * 64;
    for (int i = tid; i < 2 * M * 64; i += 512) dst[i] = red[i];
}

// dw[ch][i'][j'] = alpha * sum_g partial[g][op][ch][8(i'+off) + (j'+off)].  32 outputs per workgroup,
// 32 strided partial sums each (1024 threads: the kernel is latency-bound, 16 loads per thread), combined
// in a fixed order (deterministic).
//   merge: both operator slots hold partials of ONE gradient (single-operator launches of k_wgrad2d): dw0 gets their sum
__global__ __launch_bounds__(1024) void k_wgrad_reduce(const float *__restrict__ partial, int G,
                                                       float *__restrict__ dw0, float alpha0,
                                                       float *__restrict__ dw1, float alpha1, int M, int P, int merge = 0)
{
    __shared__ float red[32][33];
    const int o = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int t = blockIdx.x * 32 + o;
    const int per = M * P * P;
    const bool live = t < (merge ? per : 2 * per);
    float sum = 0.0f;
    int op = 0, r = 0;
    if (live) {
        op = t / per; r = t % per;
        const int ch = r / (P * P), ij = r % (P * P), off = (7 - P) / 2;
        const int tap = 8 * (ij / P + off) + (ij % P + off);
        for (int g = part; g < G; g += 32) {
            sum += partial[((size_t)(g * 2 + op) * M + ch) * 64 + tap];
            if (merge) sum += partial[((size_t)(g * 2 + 1) * M + ch) * 64 + tap];
        }
    }
    red[part][o] = sum;
    __syncthreads();
    if (part == 0 && live) {
        float *dw = op ? dw1 : dw0;
        if (dw) {
            float tot = 0.0f;
#pragma unroll
            for (int k = 0; k < 32; ++k) tot += red[k][o];
            dw[r] = (op ? alpha1 : alpha0) * tot;
        }
    }
}

// dt0[m] = sum over workgroups; dt1[m] = sum_n c[n] * (sum over the workgroups of image n).  4 channels x 256
// row-strided partial sums per workgroup, tree-combined in a fixed order.
__global__ __launch_bounds__(1024) void k_dtau_reduce(const float *__restrict__ partial,
                                                      const float *__restrict__ c, float *__restrict__ dt0,
                                                      float *__restrict__ dt1, int N, int per_img, int M)
{
    __shared__ float r0[256][4], r1[256][4];
    const int mi = threadIdx.x & 3, part = threadIdx.x >> 2;
    const int m = blockIdx.x * 4 + mi;
    float a0 = 0.0f, a1 = 0.0f;
    const int rows = N * per_img;
    if (m < M)
        for (int row = part; row < rows; row += 256) {
            const float v = partial[(size_t)row * M + m];
            a0 += v;
            if (c) a1 = fmaf(c[row / per_img], v, a1);
        }
    r0[part][mi] = a0;
    r1[part][mi] = a1;
    __syncthreads();
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if (part < stride) { r0[part][mi] += r0[part + stride][mi]; r1[part][mi] += r1[part + stride][mi]; }
        __syncthreads();
    }
    if (part == 0 && m < M) { dt0[m] = r0[0][mi]; dt1[m] = r1[0][mi]; }
}

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline bool fused_shape_ok(const cdl_geom *g)
{
    if (!cdl_geom_ok(g)) return false;
    if (g->C != 1 || g->D != 1 || g->Pd != 1 || g->sd != 1 || g->sh != 1 || g->sw != 1) return false;
    if (g->Ph != g->Pw || g->Ph > 7 || (g->Ph & 1) == 0 || g->ph != g->Ph / 2 || g->pw != g->Pw / 2) return false;
    if (g->M != 32 && g->M != 64) return false;
    if (blk_image_elems(g->M, g->H, g->W) * 4 >= ((size_t)1 << 31)) return false;   // per-image buffer descriptor range
    return true;
}

inline int debug_flags() { return cdl_opts().fused_debug; }

inline int tiles_x(const cdl_geom *g) { return (g->W + TW - 1) / TW; }
inline int tiles_y(const cdl_geom *g) { return (g->H + TH - 1) / TH; }

// `precision` argument of the entry points: bits 0-3 arithmetic (0 split-bf16 x3, 1 plain bf16, 2 split-bf16 x4: the
// lo * lo products as well -- exact fp32 products, for objectives that difference two forward passes), bit 4
// CDL_TILES_REVERSED, bits 5-6 layout of the fat INPUT(s), bits 7-8 layout of the fat OUTPUT (LAY_*)
struct Flags {
    int prec, rev, lin, lout;
    bool ok;
};
inline Flags parse_flags(int precision)
{
    Flags f;
    f.prec = precision & 15;
    f.rev = (precision >> 4) & 1;
    f.lin = (precision >> 5) & 3;
    f.lout = (precision >> 7) & 3;
    f.ok = (precision >> 9) == 0 && (f.prec == 0 || f.prec == 1 || f.prec == 2) && f.lin <= LAY_BLK16 && f.lout <= LAY_BLK16;
    return f;
}

template <int MT, int PREC, int MODE, int LIN, int LOUT>
int launch_stage_one(const FusedParams &p, dim3 grid, hipStream_t st)
{
    if constexpr (MODE == MODE_BWD && TH == GW_TH) {
        if (p.r2) {                                          // the reverse stage that also accumulates dA_k
            if (int rc = cdl_ensure_dynamic_lds((const void *)k_stage<MT, PREC, MODE, LIN, LOUT, true>, LDS_STAGE + LDS_DA)) return rc;
            k_stage<MT, PREC, MODE, LIN, LOUT, true><<<grid, NT, LDS_STAGE + LDS_DA, st>>>(p);
            hipError_t e = hipGetLastError();
            return e == hipSuccess ? 0 : -(int)e;
        }
    }
    if (p.r2) return CDL_EUNSUPPORTED;
    if (int rc = cdl_ensure_dynamic_lds((const void *)k_stage<MT, PREC, MODE, LIN, LOUT>, LDS_STAGE)) return rc;
    k_stage<MT, PREC, MODE, LIN, LOUT><<<grid, NT, LDS_STAGE, st>>>(p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// The layout pairs the sweeps and the step-wise entry points use: same layout on both sides, or NCHW on one
// side (z_K leaves a sweep as NCHW, a user gradient enters one as NCHW).  MODE_FIRST has no fat input.
template <int MT, int PREC, int MODE>
int launch_stage_lay(const FusedParams &p, int lin, int lout, dim3 grid, hipStream_t st)
{
    if (MODE == MODE_FIRST) lin = lout;
#define CDL_LAY_CASE(a, b) if (lin == a && lout == b) return launch_stage_one<MT, PREC, MODE, a, b>(p, grid, st)
    CDL_LAY_CASE(LAY_NCHW, LAY_NCHW);
    CDL_LAY_CASE(LAY_BLK, LAY_BLK);
    CDL_LAY_CASE(LAY_BLK16, LAY_BLK16);
    if constexpr (MODE == MODE_FWD) {
        CDL_LAY_CASE(LAY_BLK, LAY_NCHW);
        CDL_LAY_CASE(LAY_BLK16, LAY_NCHW);
    }
    if constexpr (MODE == MODE_BWD) {
        CDL_LAY_CASE(LAY_NCHW, LAY_BLK);
        CDL_LAY_CASE(LAY_NCHW, LAY_BLK16);
    }
#undef CDL_LAY_CASE
    return CDL_EUNSUPPORTED;
}

template <int MT, int PREC>
int launch_stage(const FusedParams &p, int mode, int lin, int lout, dim3 grid, hipStream_t st)
{
    if (mode == MODE_FWD) return launch_stage_lay<MT, PREC, MODE_FWD>(p, lin, lout, grid, st);
    if (mode == MODE_FIRST) return launch_stage_lay<MT, PREC, MODE_FIRST>(p, lin, lout, grid, st);
    return launch_stage_lay<MT, PREC, MODE_BWD>(p, lin, lout, grid, st);
}

// persistent workgroups: one per CU (most of its LDS each), striding over the tiles
int stage_grid(const cdl_geom *, const FusedParams &p)
{
    const size_t tiles = (size_t)p.N * p.tilesX * p.tilesY;
    size_t cus = (size_t)cdl_cu_count();
    const int cap = cdl_opts().fused_grid;                   // experiments only: fewer persistent workgroups
    if (cap > 0 && (size_t)cap < cus) cus = (size_t)cap;
    return (int)(tiles < cus ? tiles : cus);
}

int dispatch_stage(const cdl_geom *g, const FusedParams &p, int mode, const Flags &f, hipStream_t st)
{
    dim3 grid((unsigned)stage_grid(g, p));
    if (g->M == 64)
        return f.prec == 0   ? launch_stage<2, 0>(p, mode, f.lin, f.lout, grid, st)
               : f.prec == 1 ? launch_stage<2, 1>(p, mode, f.lin, f.lout, grid, st)
                             : launch_stage<2, 2>(p, mode, f.lin, f.lout, grid, st);
    return f.prec == 0   ? launch_stage<1, 0>(p, mode, f.lin, f.lout, grid, st)
           : f.prec == 1 ? launch_stage<1, 1>(p, mode, f.lin, f.lout, grid, st)
                         : launch_stage<1, 2>(p, mode, f.lin, f.lout, grid, st);
}

int wgrad_grid(const cdl_geom *g)
{
    const size_t tiles = (size_t)g->N * tiles_x(g) * ((g->H + GW_TH - 1) / GW_TH);
    return (int)(tiles < 512 ? tiles : 512);
}

template <int MT, int PREC, int LAY>
int launch_wgrad_one(const WgradParams &p, int G, hipStream_t st)
{
    const size_t lds = (size_t)WG_THIN_BYTES + (size_t)8 * MT * 2 * IMG_ELEMS * 2;
    if (int rc = cdl_ensure_dynamic_lds((const void *)k_wgrad2d<MT, PREC, LAY>, (int)lds)) return rc;
    k_wgrad2d<MT, PREC, LAY><<<G, 512, lds, st>>>(p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <int MT, int PREC>
int launch_wgrad(const WgradParams &p, int lay, int G, hipStream_t st)
{
    if (lay == LAY_BLK) return launch_wgrad_one<MT, PREC, LAY_BLK>(p, G, st);
    if (lay == LAY_BLK16) return launch_wgrad_one<MT, PREC, LAY_BLK16>(p, G, st);
    return launch_wgrad_one<MT, PREC, LAY_NCHW>(p, G, st);
}

}  // namespace

extern "C" {

int cdl_fused2d_supported(const cdl_geom *g) { return fused_shape_ok(g) ? 1 : 0; }

size_t cdl_fused2d_frag_bytes(int M) { return (size_t)2 * (M / 32) * 8 * 64 * 16; }

// (buffers a caller sizes with these serve sweeps of either object: the larger of the two tile geometries)
#ifndef CDL_F2D_WY4
size_t cdl_fused2d_patch_floats_wy4(const cdl_geom *g);
size_t cdl_fused2d_tiles_wy4(const cdl_geom *g);
size_t cdl_fused2d_wgrad_workspace_floats_wy4(const cdl_geom *g);
static size_t larger(size_t a, size_t b) { return a > b ? a : b; }
#else
static size_t larger(size_t a, size_t) { return a; }
#endif

size_t cdl_fused2d_patch_floats(const cdl_geom *g)
{
    if (!fused_shape_ok(g)) return 0;
    const size_t own = (size_t)g->N * tiles_x(g) * tiles_y(g) * SLAB;
#ifndef CDL_F2D_WY4
    return larger(own, cdl_fused2d_patch_floats_wy4(g));
#else
    return own;
#endif
}

size_t cdl_fused2d_tiles(const cdl_geom *g)
{
    if (!fused_shape_ok(g)) return 0;
    const size_t own = (size_t)g->N * tiles_x(g) * tiles_y(g);
#ifndef CDL_F2D_WY4
    return larger(own, cdl_fused2d_tiles_wy4(g));
#else
    return own;
#endif
}

int cdl_fused2d_prep(const float *wA, const float *wB, void *frags, int M, int P, void *stream)
{
    if (!wA || !wB || !frags || (M != 32 && M != 64) || P < 1 || P > 7 || !(P & 1)) return CDL_EINVAL;
    const int MT = M / 32, threads = 8 * MT * 64;
    k_prep<<<(threads + 255) / 256, 256, 0, S(stream)>>>(wA, wB, reinterpret_cast<uint4 *>(frags), MT, P);
    CDL_LAUNCH_CHECK();
    return 0;
}

size_t cdl_fused2d_code_bytes(const cdl_geom *g, int layout)
{
    if (!fused_shape_ok(g) || layout < 0 || layout > LAY_BLK16) return 0;
    const size_t per = layout == LAY_NCHW ? lay_image_bytes<LAY_NCHW>(g->M, g->H, g->W)
                     : layout == LAY_BLK ? lay_image_bytes<LAY_BLK>(g->M, g->H, g->W)
                                         : lay_image_bytes<LAY_BLK16>(g->M, g->H, g->W);
    return (size_t)g->N * per;
}

size_t cdl_fused2d_map_words(const cdl_geom *g)
{
    if (!fused_shape_ok(g)) return 0;
    return (size_t)g->N * 4 * g->H * g->W;
}

int cdl_fused2d_support_map(const cdl_geom *g, const float *z, unsigned *map, void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (!z || !map) return CDL_EINVAL;
    const size_t total = (size_t)g->N * 2 * g->H * g->W;
    k_support_map<<<(unsigned)((total + 255) / 256), 256, 0, S(stream)>>>(z, map, g->N, g->M, g->H, g->W);
    CDL_LAUNCH_CHECK();
    return 0;
}

int cdl_fused2d_iter_fwd(const cdl_geom *g, const float *r, const float *zin, const float *tau,
                         const void *frags, float sgn, float *zout, float *patches, unsigned *map_out,
                         int precision, void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (!r || !tau || !frags || !zout || !patches || zout == zin) return CDL_EINVAL;
    const Flags f = parse_flags(precision);
    if (!f.ok) return CDL_EINVAL;
    FusedParams p = {};
    p.rev = f.rev;
    p.r = r; p.zin = zin; p.zout = zout; p.tau = tau; p.map = map_out;
    p.frags = reinterpret_cast<const uint4 *>(frags);
    p.patches = patches; p.sgn = sgn; p.do_synth = 1; CDL_DBG_FIELD(p.dbg = debug_flags();)
    p.N = g->N; p.H = g->H; p.W = g->W;
    p.tilesX = tiles_x(g); p.tilesY = tiles_y(g);
    return dispatch_stage(g, p, zin ? MODE_FWD : MODE_FIRST, f, S(stream));
}

// r2 != nullptr: the launch also produces dA = alpha * du_out (x) im2col(r2) (cdl_fused2d_wgrad's first operator pair, same
// arithmetic), through `workspace` (cdl_fused2d_wgrad_workspace_floats) -- one fat read of du_out less than the two-launch form
static int stage_bwd(const cdl_geom *g, const float *thin, const float *base, const unsigned *map, const void *frags,
                     float *du_out, float *patches, float *dtau_partial, int do_synth, int precision, const float *r2,
                     float alpha, float *dA, float *workspace, void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (!thin || !map || !frags || !du_out || !dtau_partial || du_out == base) return CDL_EINVAL;
    if (do_synth && !patches) return CDL_EINVAL;
    if (r2 && (!dA || !workspace)) return CDL_EINVAL;
    const Flags f = parse_flags(precision);
    if (!f.ok) return CDL_EINVAL;
    FusedParams p = {};
    p.rev = f.rev;
    p.r = thin; p.zin = base; p.map = const_cast<unsigned *>(map); p.zout = du_out; p.dtau = dtau_partial;
    p.frags = reinterpret_cast<const uint4 *>(frags);
    p.patches = patches; p.sgn = 1.0f; p.do_synth = do_synth ? 1 : 0; CDL_DBG_FIELD(p.dbg = debug_flags();)
    p.N = g->N; p.H = g->H; p.W = g->W;
    p.tilesX = tiles_x(g); p.tilesY = tiles_y(g);
    p.r2 = r2; p.da_partial = workspace;
    int rc = dispatch_stage(g, p, MODE_BWD, f, S(stream));
    if (rc || !r2) return rc;
    const int G = stage_grid(g, p);
    const int total = 2 * g->M * g->Ph * g->Pw;
    k_wgrad_reduce<<<(total + 31) / 32, 1024, 0, S(stream)>>>(workspace, G, dA, alpha, nullptr, 0.0f, g->M, g->Ph);
    CDL_LAUNCH_CHECK();
    return 0;
}

int cdl_fused2d_stage_bwd(const cdl_geom *g, const float *thin, const float *base, const unsigned *map,
                          const void *frags, float *du_out, float *patches, float *dtau_partial,
                          int do_synth, int precision, void *stream)
{
    return stage_bwd(g, thin, base, map, frags, du_out, patches, dtau_partial, do_synth, precision, nullptr, 0.0f, nullptr,
                     nullptr, stream);
}

int cdl_fused2d_stage_bwd_da(const cdl_geom *g, const float *thin, const float *base, const unsigned *map,
                             const void *frags, float *du_out, float *patches, float *dtau_partial, int do_synth,
                             const float *r2, float alpha, float *dA, float *workspace, int precision, void *stream)
{
    if (!r2 || !dA || !workspace) return CDL_EINVAL;
    return stage_bwd(g, thin, base, map, frags, du_out, patches, dtau_partial, do_synth, precision, r2, alpha, dA, workspace,
                     stream);
}

int cdl_fused2d_dtau_reduce(const cdl_geom *g, const float *dtau_partial, const float *c, float *dt0,
                            float *dt1, void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (!dtau_partial || !dt0 || !dt1) return CDL_EINVAL;
    k_dtau_reduce<<<(g->M + 3) / 4, 1024, 0, S(stream)>>>(dtau_partial, c, dt0, dt1, g->N, tiles_x(g) * tiles_y(g), g->M);
    CDL_LAUNCH_CHECK();
    return 0;
}

size_t cdl_fused2d_wgrad_workspace_floats(const cdl_geom *g)
{
    if (!fused_shape_ok(g)) return 0;
    const size_t own = (size_t)wgrad_grid(g) * 2 * g->M * 64;
#ifndef CDL_F2D_WY4
    return larger(own, cdl_fused2d_wgrad_workspace_floats_wy4(g));
#else
    return own;
#endif
}

int cdl_fused2d_wgrad(const cdl_geom *g, const float *X0, const float *T0, float alpha0, float *dw0,
                      const float *X1, const float *T1, float alpha1, float *dw1, float *workspace,
                      int precision, void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (!workspace || (!X0 && !X1)) return CDL_EINVAL;
    if ((X0 && (!T0 || !dw0)) || (X1 && (!T1 || !dw1))) return CDL_EINVAL;
    const Flags f = parse_flags(precision);
    if (!f.ok) return CDL_EINVAL;
    WgradParams p = {};
    p.rev = f.rev;
    p.X[0] = X0; p.T[0] = T0; p.X[1] = X1; p.T[1] = T1;
    p.single = (X0 != nullptr) != (X1 != nullptr);
    if (p.single) {                                         // one gradient: both wave groups on it
        if (!X0) { X0 = X1; T0 = T1; alpha0 = alpha1; dw0 = dw1; X1 = nullptr; }
        p.X[0] = p.X[1] = X0; p.T[0] = p.T[1] = T0;
    }
    p.partial = workspace;
    p.N = g->N; p.H = g->H; p.W = g->W;
    p.tilesX = tiles_x(g); p.tilesY = (g->H + GW_TH - 1) / GW_TH;
    p.numTiles = p.N * p.tilesX * p.tilesY;
    int G = wgrad_grid(g);
    if (p.single && G > (p.numTiles + 1) / 2) G = (p.numTiles + 1) / 2;
    int rc;
    if (g->M == 64)
        rc = f.prec == 0 ? launch_wgrad<2, 0>(p, f.lin, G, S(stream))
             : f.prec == 1 ? launch_wgrad<2, 1>(p, f.lin, G, S(stream)) : launch_wgrad<2, 2>(p, f.lin, G, S(stream));
    else
        rc = f.prec == 0 ? launch_wgrad<1, 0>(p, f.lin, G, S(stream))
             : f.prec == 1 ? launch_wgrad<1, 1>(p, f.lin, G, S(stream)) : launch_wgrad<1, 2>(p, f.lin, G, S(stream));
    if (rc) return rc;
    const int total = 2 * g->M * g->Ph * g->Pw;
    k_wgrad_reduce<<<(total + 31) / 32, 1024, 0, S(stream)>>>(workspace, G, X0 ? dw0 : nullptr, alpha0,
                                                             X1 ? dw1 : nullptr, alpha1, g->M, g->Ph, p.single);
    CDL_LAUNCH_CHECK();
    return 0;
}

int cdl_fused2d_assemble(const cdl_geom *g, const float *patches, const float *mask, const float *sub,
                         float alpha, float *out, void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (!patches || !out) return CDL_EINVAL;
    if ((size_t)g->N * g->H * g->W >= ((size_t)1 << 20) && (g->W & 3) == 0 && !cdl_opts().scalar_assemble) {
        dim3 grid((unsigned)((g->W + 255) / 256), (unsigned)((g->H + 7) / 8), (unsigned)g->N);
        k_assemble_v4<<<grid, 256, 0, S(stream)>>>(patches, mask, sub, alpha, out, g->N, g->H, g->W, tiles_x(g), tiles_y(g));
    } else if ((size_t)g->N * g->H * g->W >= ((size_t)1 << 20)) {
        dim3 grid((unsigned)((g->W + 255) / 256), (unsigned)((g->H + 3) / 4), (unsigned)g->N);
        k_assemble<4><<<grid, 256, 0, S(stream)>>>(patches, mask, sub, alpha, out, g->N, g->H, g->W, tiles_x(g), tiles_y(g));
    } else {
        dim3 grid((unsigned)((g->W + 255) / 256), (unsigned)g->H, (unsigned)g->N);
        k_assemble<1><<<grid, 256, 0, S(stream)>>>(patches, mask, sub, alpha, out, g->N, g->H, g->W, tiles_x(g), tiles_y(g));
    }
    CDL_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

/* ---- optional per-kernel timing inside the sweeps (HIP events on the launch stream) -------------------
 * bench.py turns it on for a few steps to report the in-step average duration of the three fat kernels
 * (isolated re-launches of one kernel miss the cache state the step leaves behind).  Off: no events. */
namespace {
// Process-wide and mutex-guarded: the one piece of state of this file, used by bench.py only.  Events are
// recorded on whichever stream the timed sweep runs on; pairs of different devices are kept apart by the
// device they were created on (an event can only be recorded on its own device's streams).
struct TimingState {
    std::mutex mu;
    std::atomic<bool> on{false};
    struct Pair { hipEvent_t start, stop; int cls, dev; };
    std::vector<Pair> pairs;               // in launch order; class 0 forward stage (k >= 1), 1 reverse stage,
                                           // 2 filter gradients, 3 first forward stage (k = 0: no fat read)
    std::vector<std::pair<hipEvent_t, int>> spare;     // recycled events with their device
};
TimingState g_timing;

struct TimingScope {
    hipStream_t st;
    hipEvent_t stop = nullptr;
    TimingScope(int c, hipStream_t s) : st(s)
    {
        if (!g_timing.on.load(std::memory_order_relaxed)) return;
        std::lock_guard<std::mutex> lk(g_timing.mu);
        const int dev = cdl_current_device();
        hipEvent_t ev[2] = {nullptr, nullptr};
        for (int i = 0; i < 2; ++i) {
            for (size_t k = 0; k < g_timing.spare.size() && !ev[i]; ++k)
                if (g_timing.spare[k].second == dev) {
                    ev[i] = g_timing.spare[k].first;
                    g_timing.spare.erase(g_timing.spare.begin() + k);
                }
            if (!ev[i] && hipEventCreate(&ev[i]) != hipSuccess) return;
        }
        (void)hipEventRecord(ev[0], st);
        stop = ev[1];
        g_timing.pairs.push_back({ev[0], ev[1], c, dev});
    }
    ~TimingScope()
    {
        if (stop) (void)hipEventRecord(stop, st);
    }
};
}  // namespace

#ifndef CDL_F2D_WY4
extern "C" int cdl_fused2d_timing_wy4(int enable);
extern "C" int cdl_fused2d_timing_read_wy4(double *ms_sum, int *count);
#endif

extern "C" int cdl_fused2d_timing(int enable)
{
#ifndef CDL_F2D_WY4
    if (int rc = cdl_fused2d_timing_wy4(enable)) return rc;     // the sweeps of the other object (bf16 code storage) too
#endif
    std::lock_guard<std::mutex> lk(g_timing.mu);
    g_timing.on.store(enable != 0, std::memory_order_relaxed);
    if (enable) {
        for (auto &pr : g_timing.pairs) {
            g_timing.spare.push_back({pr.start, pr.dev});
            g_timing.spare.push_back({pr.stop, pr.dev});
        }
        g_timing.pairs.clear();
    }
    return 0;
}

extern "C" int cdl_fused2d_timing_read(double *ms_sum, int *count)
{
    if (!ms_sum || !count) return CDL_EINVAL;
    std::lock_guard<std::mutex> lk(g_timing.mu);
    for (int c = 0; c < 4; ++c) { ms_sum[c] = 0.0; count[c] = 0; }
    for (auto &pr : g_timing.pairs) {
        float ms = 0.0f;
        hipError_t e = hipEventSynchronize(pr.stop);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, pr.start, pr.stop);
        if (e != hipSuccess) return -(int)e;
        ms_sum[pr.cls] += ms;
        ++count[pr.cls];
    }
#ifndef CDL_F2D_WY4
    double ms2[4];
    int n2[4];
    if (int rc = cdl_fused2d_timing_read_wy4(ms2, n2)) return rc;
    for (int c = 0; c < 4; ++c) { ms_sum[c] += ms2[c]; count[c] += n2[c]; }
#endif
    return 0;
}

extern "C" {
/* ---- whole sweeps: every launch of a forward / reverse pass enqueued from one C call ----------------
 * Snake order: consecutive fat launches walk the tiles in opposite directions, so each one starts on
 * the bytes the previous one touched last, which are still in the 256 MiB Infinity Cache (it holds
 * the last ~256 MiB loaded or stored).  Tile results do not depend on the order; only the filter
 * gradients' per-workgroup partial sums are grouped differently.  CDL_FUSED_SNAKE=0 turns it off. */
// fragments of K (analysis-like, synthesis-like) pairs, pair k at frags + k * cdl_fused2d_frag_bytes(M)
static int prep_pairs(const float *const *w1, const float *const *w2, int K, int shift2, void *frags, int M, int P,
                      hipStream_t st)
{
    const int MT = M / 32, threads = 8 * MT * 64;
    const int frag_uint4 = (int)(cdl_fused2d_frag_bytes(M) / 16);
    for (int k0 = 0; k0 < K; k0 += PREP_BATCH) {
        const int nb = K - k0 < PREP_BATCH ? K - k0 : PREP_BATCH;
        PrepBatch b = {};
        for (int i = 0; i < nb; ++i) {
            const int k = k0 + i;
            // forward: (A_k, B_{k+1 mod K});  backward: (B_{k+1 mod K}, A_k) -- the shifted bank is index shift2
            b.wA[i] = shift2 == 0 ? w1[(k + 1) % K] : w1[k];
            b.wB[i] = shift2 == 0 ? w2[k] : w2[(k + 1) % K];
        }
        dim3 grid((unsigned)((threads + 255) / 256), (unsigned)nb);
        k_prep_batch<<<grid, 256, 0, st>>>(b, reinterpret_cast<uint4 *>(frags) + (size_t)k0 * frag_uint4, frag_uint4, MT, P);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return -(int)e;
    }
    return 0;
}

static int snake_enabled() { return cdl_opts().fused_snake; }

// Sweeps: CDL_LAYOUT_IN(L) in `precision` selects the layout L of the tensors that stay inside the sweeps --
// z_1 .. z_{K-1} (z[0..K-2]) and the du ping-pong buffers; z_K (z[K-1]) and g_z are always NCHW fp32.

#ifndef CDL_F2D_WY4
int cdl_fused2d_forward_wy4(const cdl_geom *g, int K, const float *yp, const float *mask, const float *tau,
                            const float *const *wA, const float *const *wB, float *const *z, float *const *r,
                            unsigned *const *maps, float *xp, void *frags, float *patches, int precision, void *stream);
int cdl_fused2d_backward_wy4(const cdl_geom *g, int K, const float *yp, const float *mask, const float *c,
                             const float *const *wA, const float *const *wB, const float *const *z,
                             const float *const *r, const unsigned *const *maps, const float *g_xp, const float *g_z,
                             float *const *dA, float *const *dB, float *dt, float *du0, float *du1, float *q, void *frags,
                             float *patches, float *dtau_partial, float *wgrad_ws, int precision, void *stream);
#endif

int cdl_fused2d_forward(const cdl_geom *g, int K, const float *yp, const float *mask, const float *tau,
                        const float *const *wA, const float *const *wB, float *const *z, float *const *r,
                        unsigned *const *maps, float *xp, void *frags, float *patches, int precision,
                        void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (K < 1 || !yp || !tau || !wA || !wB || !z || !xp || !frags || !patches || (K > 1 && !r)) return CDL_EINVAL;
    const size_t nm = (size_t)g->N * g->M;
    const int snake = snake_enabled();
    const Flags f = parse_flags(precision);
    if (!f.ok || f.lout != 0) return CDL_EINVAL;
    const int L = f.lin;
#ifndef CDL_F2D_WY4
    if (L == LAY_BLK16)                      // bf16 code storage: the two-waves-per-SIMD object (see the top of the file)
        return cdl_fused2d_forward_wy4(g, K, yp, mask, tau, wA, wB, z, r, maps, xp, frags, patches, precision, stream);
#endif
    const float *thin = yp;
    const size_t fb = cdl_fused2d_frag_bytes(g->M);
    int rc = prep_pairs(wA, wB, K, 1, frags, g->M, g->Ph, S(stream));       // (A_k, B_{k+1}) for every k, one launch
    if (rc) return rc;
    for (int k = 0; k < K; ++k) {
        const void *fk = static_cast<const char *>(frags) + (size_t)k * fb;
        {
            TimingScope ts(k ? 0 : 3, S(stream));
            rc = cdl_fused2d_iter_fwd(g, thin, k ? z[k - 1] : nullptr, tau + k * nm, fk, k ? -1.0f : 1.0f, z[k],
                                      patches, maps ? maps[k] : nullptr,
                                      f.prec | CDL_LAYOUT_IN(L) | CDL_LAYOUT_OUT(k == K - 1 ? LAY_NCHW : L) |
                                          ((k & 1) && snake ? CDL_TILES_REVERSED : 0),
                                      stream);
        }
        if (rc) return rc;
        if (k < K - 1) {
            rc = cdl_fused2d_assemble(g, patches, mask, yp, 1.0f, r[k], stream);
            thin = r[k];
        } else {
            rc = cdl_fused2d_assemble(g, patches, nullptr, nullptr, 1.0f, xp, stream);
        }
        if (rc) return rc;
    }
    return 0;
}

int cdl_fused2d_backward(const cdl_geom *g, int K, const float *yp, const float *mask, const float *c,
                         const float *const *wA, const float *const *wB, const float *const *z,
                         const float *const *r, const unsigned *const *maps, const float *g_xp,
                         const float *g_z, float *const *dA,
                         float *const *dB, float *dt, float *du0, float *du1, float *q, void *frags,
                         float *patches, float *dtau_partial, float *wgrad_ws, int precision, void *stream)
{
    if (!fused_shape_ok(g)) return CDL_EUNSUPPORTED;
    if (K < 1 || !yp || !wA || !wB || !z || !maps || !g_xp || !dA || !dB || !dt || !du0 || !du1 || !q || !frags ||
        !patches || !dtau_partial || !wgrad_ws || (K > 1 && !r))
        return CDL_EINVAL;
    const int M = g->M;
    float *du[2] = {du0, du1};
    const Flags f = parse_flags(precision);
    if (!f.ok || f.lout != 0) return CDL_EINVAL;
    const int L = f.lin;
#ifndef CDL_F2D_WY4
    if (L == LAY_BLK16)
        return cdl_fused2d_backward_wy4(g, K, yp, mask, c, wA, wB, z, r, maps, g_xp, g_z, dA, dB, dt, du0, du1, q, frags,
                                        patches, dtau_partial, wgrad_ws, precision, stream);
#endif
    // the forward's last launch ran in direction (K-1)&1: stages take that one, filter gradients the other
    const int sdir = snake_enabled() ? (((K - 1) & 1) ? CDL_TILES_REVERSED : 0) : 0;
    const int wdir = snake_enabled() ? (sdir ^ CDL_TILES_REVERSED) : 0;
    const int sprec = f.prec | sdir, wprec = f.prec | wdir;
    int rc = cdl_fused2d_wgrad(g, z[K - 1], g_xp, 1.0f, dB[0], nullptr, nullptr, 0.0f, nullptr, wgrad_ws,
                               wprec, stream);                           // dB_0 = z_K (x) dL/d(D z_K); z_K is NCHW
    if (rc) return rc;
    const float *thin = g_xp, *base = g_z;
    const size_t fb = cdl_fused2d_frag_bytes(M);
    rc = prep_pairs(wB, wA, K, 0, frags, M, g->Ph, S(stream));               // (B_{k+1}, A_k) for every k, one launch
    if (rc) return rc;
    for (int k = K - 1, flip = 0; k >= 0; --k, flip ^= 1) {
        const void *fk = static_cast<const char *>(frags) + (size_t)k * fb;
        float *duk = du[flip];
        // dA_k = -du_k (x) r_k (k = 0: du_0 (x) yp) rides in the reverse stage, which has du_k in its registers: du_k is
        // read once (by stage k-1), not twice -- 5.1 fat passes per iteration instead of 6.1 (CDL_FUSED_DA=0: the two-launch
        // form, for A/B runs)
        const bool ride = TH == GW_TH && cdl_opts().fused_da;
        {
            TimingScope ts(1, S(stream));
            rc = stage_bwd(g, thin, base, maps[k], fk, duk, patches, dtau_partial, k >= 1,
                           sprec | CDL_LAYOUT_IN(k == K - 1 ? LAY_NCHW : L) | CDL_LAYOUT_OUT(L),
                           ride ? (k >= 1 ? r[k - 1] : yp) : nullptr, k >= 1 ? -1.0f : 1.0f, dA[k], wgrad_ws, stream);
        }
        if (rc) return rc;
        rc = cdl_fused2d_dtau_reduce(g, dtau_partial, c, dt + (size_t)k * 2 * M, dt + (size_t)k * 2 * M + M, stream);
        if (rc) return rc;
        if (k >= 1) {
            rc = cdl_fused2d_assemble(g, patches, mask, nullptr, -1.0f, q, stream);
            if (rc) return rc;
            {
                TimingScope ts(2, S(stream));
                rc = ride ? cdl_fused2d_wgrad(g, z[k - 1], q, 1.0f, dB[k], nullptr, nullptr, 0.0f, nullptr, wgrad_ws,
                                              wprec | CDL_LAYOUT_IN(L), stream)
                          : cdl_fused2d_wgrad(g, duk, r[k - 1], -1.0f, dA[k], z[k - 1], q, 1.0f, dB[k], wgrad_ws,
                                              wprec | CDL_LAYOUT_IN(L), stream);
            }
            thin = q;
        } else if (!ride) {
            rc = cdl_fused2d_wgrad(g, duk, yp, 1.0f, dA[0], nullptr, nullptr, 0.0f, nullptr, wgrad_ws,
                                   wprec | CDL_LAYOUT_IN(L), stream);
        }
        if (rc) return rc;
        base = duk;
    }
    return 0;
}

}  // extern "C"
