// GPU-box diagnostic: do the matrix pipe and the vector ALU of a CDNA4 SIMD run concurrently?  A persistent grid of
// 512-thread workgroups (2 waves per SIMD, as the fused kernels run) executes, per loop iteration and wave, NM
// independent 32x32x16 bf16 MFMAs (4 rotating accumulators: no dependent back-to-back issue) and NV independent
// v_fma_f32.  Reported: shader cycles per iteration for (NM, 0), (0, NV) and (NM, NV).  If the two pipes overlapped,
// t(NM, NV) ~ max(t(NM, 0), t(0, NV)); if a SIMD issues one or the other, t(NM, NV) ~ t(NM, 0) + t(0, NV).
//   hipcc --offload-arch=gfx950 -O3 -o probe_issue probe_issue.hip && ./probe_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NM, int NV>
__global__ __launch_bounds__(512) void k_issue(float *out, long long *cycles, int iters)
{
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int v = 0; v < 16; ++v) acc[a][v] = 0.0f;
    bf16x8 x[4], y;                      // a different operand per accumulator: identical products would be merged
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 8; ++i) x[a][i] = (__bf16)(0.001f * (threadIdx.x + i + 3 * a));
    for (int i = 0; i < 8; ++i) y[i] = (__bf16)(0.002f * (threadIdx.x - i));
    float f[16];
    for (int i = 0; i < 16; ++i) f[i] = 0.5f + 0.01f * (threadIdx.x + i);
    const float m = 0.999f, c = 0.001f;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < (NM > NV / 8 ? NM : NV / 8); ++k) {       // interleave: 1 MFMA, then 8 VALU, ...
            if (k < NM) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[k & 3], y, acc[k & 3], 0, 0, 0);
            if (k < NV / 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[(8 * k + j) & 15] = __builtin_amdgcn_fmed3f(f[(8 * k + j) & 15] + c, m, c);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int a = 0; a < 4; ++a)
        for (int v = 0; v < 16; ++v) s += acc[a][v];
    for (int i = 0; i < 16; ++i) s += f[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NM, int NV>
double run(float *out, long long *cyc, int iters)
{
    const int grid = 256;
    k_issue<NM, NV><<<grid, 512>>>(out, cyc, iters);
    CK(hipDeviceSynchronize());
    k_issue<NM, NV><<<grid, 512>>>(out, cyc, iters);
    CK(hipDeviceSynchronize());
    long long h[256];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double s = 0;
    for (int i = 0; i < grid; ++i) s += (double)h[i];
    return s / grid / iters;
}

int main()
{
    float *out; long long *cyc;
    CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    CK(hipMalloc(&cyc, 256 * sizeof(long long)));
    const int iters = 2000;
    const double m = run<16, 0>(out, cyc, iters), v = run<0, 128>(out, cyc, iters), b = run<16, 128>(out, cyc, iters);
    const double m2 = run<8, 0>(out, cyc, iters), v2 = run<0, 256>(out, cyc, iters), b2 = run<8, 256>(out, cyc, iters);
    printf("{\"probe\": \"issue\", \"waves_per_simd\": 2, \"mfma\": 16, \"valu\": 128, \"cycles_mfma_only\": %.0f, \"cycles_valu_only\": %.0f, \"cycles_both\": %.0f, \"sum\": %.0f, \"max\": %.0f}\n", m, v, b, m + v, m > v ? m : v);
    printf("{\"probe\": \"issue\", \"waves_per_simd\": 2, \"mfma\": 8, \"valu\": 256, \"cycles_mfma_only\": %.0f, \"cycles_valu_only\": %.0f, \"cycles_both\": %.0f, \"sum\": %.0f, \"max\": %.0f}\n", m2, v2, b2, m2 + v2, m2 > v2 ? m2 : v2);
    return 0;
}
