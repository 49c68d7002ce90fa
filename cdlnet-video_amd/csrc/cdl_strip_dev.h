// Device-side pieces shared by the strip kernels (cdl_strip.hip: one image channel, stride 1 / 2, M <= 192;
// cdl_stripg.hip: channel / depth groups, unit stride, M <= 64).  Included inside each file's anonymous namespace.
#pragma once

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int NWV = 8;                 // waves per workgroup (they share the weights, nothing else)
constexpr int NTS = 64 * NWV;
constexpr int OOB = 0x7fff0000;
constexpr int MAXP = 3;                // channel-tile pairs: M <= 192

enum { MODE_FWD = 0, MODE_FIRST = 1, MODE_BWD = 2 };

// tap (i, j) -> row of the synthesis-like accumulator tiles (as cdl_fusedg.hip): row t of a 32-row tile sits in
// register v = 4*((t>>3)&3) + (t&3) of lane half (t>>2)&1
template <int P> __host__ __device__ constexpr int tap_slot(int i, int j)
{
    if (P == 5) return (i < 4 && j < 4) ? 8 * i + j : (i < 4 ? 8 * i + 4 : (j < 4 ? 8 * j + 5 : 6));
    return 8 * i + j;
}
template <int P> struct Shape {
    static constexpr int RT = (P == 7) ? 2 : 1;
    static constexpr int KS = (P * P + 15) / 16;
    static constexpr int RC = 8;                       // circular rows of the thin buffer (>= P, power of two)
    static constexpr int RBUF = RC + P - 1;            // + mirror rows so that a P-row window never wraps
};
template <int P, int S> struct Strip {
    static constexpr int XW = S * 31 + P;              // image columns under 32 code columns
    static constexpr int XWP = (XW + 3) & ~3;
    static constexpr int JM = (P + S - 1) / S;         // taps per parity class (max)
    static constexpr int PXW = 32 + JM - 1;            // half-resolution patch columns
    static constexpr int NLD = (S * XW + 63) / 64;     // thin loads per lane and code row
    static constexpr int NPRE = (P - S + S - 1) / S;   // row groups before the first code row's own
};

__device__ __forceinline__ float wave_shr1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}

// (cdl_fusedg.hip) totals of 16 per-lane values over the 32 pixel lanes of each half-wave; lanes c and c ^ 16 end up
// with the two halves of total #(c mod 16)
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

// Epilogue of one channel tile (a plain function, not a closure inside the step: a nested closure kept the accumulator
// sets in scratch memory).  acc holds u = zin + sgn A r (forward) or du' + B^T q (reverse); on return it holds the code /
// the gated gradient, which has also been stored.
// PARTIAL: M need not be a multiple of 8 -- the register quad that straddles M addresses its missing channels out of range
template <int MODE, bool MAPPED, bool GENERAL, bool PARTIAL = true>
__device__ __forceinline__ void strip_epilogue(f32x16 &acc, float (&tsum)[16], unsigned &ws, unsigned &wg,
                                               const float *tau_s, int R, int M, int h, int cb_part,
                                               const int (&vo_part)[4], int voff_x, __amdgpu_buffer_rsrc_t rs_out, int so0,
                                               int plane4, unsigned sup, unsigned sgb, bool nostore)
{
#pragma unroll
    for (int qv = 0; qv < 4; ++qv) {
        const int cb = 32 * R + 8 * qv;
        // (tsum of the padding quads is zeroed by the caller, not here: a store in this branch and the store of the
        //  other one get merged into one store through a pointer phi, which keeps both arrays in scratch memory)
        if (cb >= M) continue;                                           // uniform: padding channels, already exactly zero
        float t4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (MODE != MODE_BWD) {
            const float4 tt = *reinterpret_cast<const float4 *>(&tau_s[cb + 4 * h]);
            t4[0] = tt.x; t4[1] = tt.y; t4[2] = tt.z; t4[3] = tt.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int v = 4 * qv + e;
            const float a = acc[v];
            float zz;
            if (MODE == MODE_BWD) {
                const bool on = (sup >> v) & 1u;
                zz = on ? a : 0.0f;
                tsum[v] = ((sgb >> v) & 1u) ? zz : -zz;
            } else {
                // t >= 0: sign(a) relu(|a| - t) == a - clamp(a, -t, t), same rounding, NaN stays NaN
                zz = GENERAL ? cdl_shrink(a, t4[e]) : a - __builtin_amdgcn_fmed3f(a, -t4[e], t4[e]);
                if (MAPPED) {
                    const unsigned bits = __builtin_bit_cast(unsigned, zz);
                    ws |= ((bits & 0x7fffffffu) != 0u ? 1u : 0u) << v;
                    wg |= (bits >> 31) << v;
                }
            }
            const int vo = nostore ? OOB : ((PARTIAL && cb == cb_part) ? vo_part[e] : voff_x);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, zz), rs_out, vo, so0 + (8 * qv + e) * plane4, 0);
            acc[v] = zz;
        }
    }
}

