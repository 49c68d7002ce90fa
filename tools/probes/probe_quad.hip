// GPU-box diagnostic: the in-quad 4x4 transpose used by the 16-byte fat access path.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void quad_transpose(float (&a)[4], int p)
{
    const bool odd = (p & 1) != 0, hi = (p & 2) != 0;
    {
        const float r01 = dpp_quad<0xB1>(odd ? a[0] : a[1]);
        const float r23 = dpp_quad<0xB1>(odd ? a[2] : a[3]);
        if (odd) { a[0] = r01; a[2] = r23; } else { a[1] = r01; a[3] = r23; }
    }
    {
        const float r02 = dpp_quad<0x4E>(hi ? a[0] : a[2]);
        const float r13 = dpp_quad<0x4E>(hi ? a[1] : a[3]);
        if (hi) { a[0] = r02; a[1] = r13; } else { a[2] = r02; a[3] = r13; }
    }
}
__global__ void k(float *out, float *raw1, float *raw2)
{
    const int lane = threadIdx.x;
    raw1[lane] = dpp_quad<0xB1>((float)lane);
    raw2[lane] = dpp_quad<0x4E>((float)lane);
    float a[4];
    for (int e = 0; e < 4; ++e) a[e] = (float)(lane * 10 + e);
    quad_transpose(a, lane & 3);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = a[e];
}
int main()
{
    float *d, *d1, *d2, h[256], h1[64], h2[64];
    (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&d1, 256); (void)hipMalloc(&d2, 256);
    k<<<1, 64>>>(d, d1, d2);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(h1, d1, 256, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h2, d2, 256, hipMemcpyDeviceToHost);
    printf("quad_perm[1,0,3,2] lanes 0..7: "); for (int i = 0; i < 8; ++i) printf("%g ", h1[i]); printf("\n");
    printf("quad_perm[2,3,0,1] lanes 0..7: "); for (int i = 0; i < 8; ++i) printf("%g ", h2[i]); printf("\n");
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const int src_lane = (lane & ~3) + e;
            if (h[lane * 4 + e] != (float)(src_lane * 10 + (lane & 3))) ++bad;
        }
    printf("lane 5 regs: %g %g %g %g (expect 41 51 61 71)\n", h[20], h[21], h[22], h[23]);
    printf("quad transpose: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    return 0;
}
