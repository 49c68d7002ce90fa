// GPU-box diagnostic: wave_shr:1 DPP and v_permlane32_swap as used by col2im_row (cdl_fused2d.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *o1, float *o2, float *o3)
{
    const float v = (float)threadIdx.x;
    o1[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v + 100.0f), 0x138, 0xf, 0xf, true));
    const unsigned u = __builtin_bit_cast(unsigned, v);
    (void)u;
    float a = v, b = 0.0f;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    o2[threadIdx.x] = b;       // expected {a.hi, 0}
    o3[threadIdx.x] = a;       // expected {a.lo, 0}
}
int main()
{
    float *d1, *d2, *d3, h1[64], h2[64], h3[64];
    (void)hipMalloc(&d1, 256); (void)hipMalloc(&d2, 256); (void)hipMalloc(&d3, 256);
    k<<<1, 64>>>(d1, d2, d3);
    (void)hipMemcpy(h1, d1, 256, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h2, d2, 256, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h3, d3, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        float e1 = i == 0 ? 0.0f : 100.0f + (i - 1);
        float e2 = i < 32 ? (float)(i + 32) : 0.0f;
        if (h1[i] != e1) ++bad;
        if (h2[i] != e2) ++bad;
    }
    printf("asm swap(a=lane,b=0): new b: lane0=%g lane31=%g lane32=%g lane63=%g | new a: lane0=%g lane31=%g lane32=%g lane63=%g\n",
           h2[0], h2[31], h2[32], h2[63], h3[0], h3[31], h3[32], h3[63]);
    printf("wave_shr1: lane0=%g lane1=%g lane32=%g lane63=%g | swap[1]: lane0=%g lane31=%g lane32=%g\n",
           h1[0], h1[1], h1[32], h1[63], h2[0], h2[31], h2[32]);
    printf("expected lane i <- lane i-1 (lane 0 <- 0) and lower <- upper: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    return 0;
}
