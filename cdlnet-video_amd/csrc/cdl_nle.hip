// Blind noise-level estimation by the median absolute deviation of the finest diagonal wavelet band
// (reference model/nle.py:17-27 nle_mad with model/wvlt.py:13-41):
//     HHy   = conv2d(y, hh, stride = 2, groups = C)      hh = flip(outer(dec_hi, dec_hi)) of 'bior4.4', 10 x 10
//     sigma = median(|HHy| over (C, H', W') per sample) / 0.6745          (torch.median: the LOWER median)
// Two kernels: the depthwise stride-2 correlation (the 2-D filter is an outer product, applied as such:
// 100 multiply-adds per output from registers) writing |HHy|, and an exact selection of the k-th smallest
// value per sample by a 3-pass radix histogram over the float bit patterns (non-negative floats order like
// their bits), integer counting only: the result is the bit-exact lower median of the band.
//
// The taps are PyWavelets' 'bior4.4' decomposition high-pass filter (pywt.Wavelet('bior4.4').dec_hi).  PyWavelets
// is not installed in the build image, so the table below is restated from the published CDF 9/7 pair; tests
// pin it through the perfect-reconstruction identity with the matching low-pass / reconstruction filters.
#include "cdl_common.h"

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

namespace {

constexpr int TAPS = 10;
// flip(dec_hi): conv2d correlates with the flipped outer product (wvlt.py:41 flips both axes)
__constant__ float c_hi_flipped[TAPS] = {0.0f,
                                         0.0f,
                                         -0.06453888262869706f,
                                         0.04068941760916406f,
                                         0.41809227322161724f,
                                         -0.7884856164055829f,
                                         0.41809227322161724f,
                                         0.04068941760916406f,
                                         -0.06453888262869706f,
                                         0.0f};

// one thread per output (n, c, i, j): |sum_a f[a] * (sum_b f[b] * y[2i + a][2j + b])|
__global__ __launch_bounds__(256) void k_hh_abs(const float *__restrict__ y, float *__restrict__ out, int NC,
                                                int H, int W, int Ho, int Wo)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)NC * Ho * Wo;
    if (idx >= total) return;
    const int j = (int)(idx % Wo);
    const size_t r = idx / Wo;
    const int i = (int)(r % Ho);
    const size_t nc = r / Ho;
    const float *base = y + (nc * H + 2 * i) * (size_t)W + 2 * j;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < TAPS; ++a) {
        float row = 0.0f;
#pragma unroll
        for (int b = 0; b < TAPS; ++b) row = fmaf(c_hi_flipped[b], base[(size_t)a * W + b], row);
        acc = fmaf(c_hi_flipped[a], row, acc);
    }
    out[idx] = fabsf(acc);
}

// k-th smallest (0-based) of `count` non-negative floats per sample: radix select, 11 + 11 + 10 bits.
__global__ __launch_bounds__(1024) void k_select(const float *__restrict__ v, float *__restrict__ out, size_t count,
                                                 size_t kth, float divisor)
{
    __shared__ unsigned hist[2048];
    __shared__ unsigned sel_prefix, sel_rank;
    const unsigned *bits = reinterpret_cast<const unsigned *>(v) + (size_t)blockIdx.x * count;
    unsigned prefix = 0, mask = 0;
    size_t rank = kth;                                     // rank of the target among the elements matching prefix
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int sh = shifts[pass], nb = 1 << widths[pass];
        for (int b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = 0;
        __syncthreads();
        for (size_t e = threadIdx.x; e < count; e += blockDim.x) {
            const unsigned u = bits[e];
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> sh) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            size_t seen = 0;
            int b = 0;
            for (; b < nb; ++b) {
                if (seen + hist[b] > rank) break;
                seen += hist[b];
            }
            sel_prefix = prefix | ((unsigned)b << sh);
            sel_rank = (unsigned)(rank - seen);
        }
        __syncthreads();
        prefix = sel_prefix;
        rank = sel_rank;
        mask |= (unsigned)(nb - 1) << sh;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_bit_cast(float, prefix) / divisor;
}

}  // namespace

extern "C" {

size_t cdl_nle_mad_scratch_floats(int N, int C, int H, int W)
{
    if (N <= 0 || C <= 0 || H < TAPS || W < TAPS) return 0;
    return (size_t)N * C * ((H - TAPS) / 2 + 1) * ((W - TAPS) / 2 + 1);
}

int cdl_nle_mad(const float *y, float *sigma_hat, float *scratch, size_t scratch_floats, int N, int C, int H, int W,
                void *stream)
{
    if (!y || !sigma_hat || !scratch || N <= 0 || C <= 0) return CDL_EINVAL;
    if (H < TAPS || W < TAPS) return CDL_EINVAL;
    const int Ho = (H - TAPS) / 2 + 1, Wo = (W - TAPS) / 2 + 1;
    const size_t total = (size_t)N * C * Ho * Wo, per_n = (size_t)C * Ho * Wo;
    if (scratch_floats < total) return CDL_EINVAL;
    k_hh_abs<<<(unsigned)((total + 255) / 256), 256, 0, S(stream)>>>(y, scratch, N * C, H, W, Ho, Wo);
    CDL_LAUNCH_CHECK();
    // torch.median returns the lower of the two middle values: 0-based rank (n - 1) / 2
    k_select<<<(unsigned)N, 1024, 0, S(stream)>>>(scratch, sigma_hat, per_n, (per_n - 1) / 2, 0.6745f);
    CDL_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
