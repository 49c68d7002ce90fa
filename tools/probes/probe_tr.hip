// GPU-box diagnostic: prints what ds_read_b64_tr_b16 delivers, so the wgrad kernel's
// transposed LDS reads can be checked against the documented lane/element map.
//   hipcc --offload-arch=gfx950 -O2 probe_tr.hip -o probe_tr && ./probe_tr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

__global__ void k(float *out_row, float *out_col)
{
    __shared__ __attribute__((aligned(16))) __bf16 img_r[16 * 64];
    __shared__ __attribute__((aligned(16))) __bf16 img_c[16 * 64];
    for (int i = threadIdx.x; i < 16 * 64; i += 64) { img_r[i] = (__bf16)(float)(i / 64); img_c[i] = (__bf16)(float)(i % 64); }
    __syncthreads();
    const int lane = threadIdx.x, grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    // group g reads the 4x16 block at rows 4g..4g+3, columns 16g..16g+15 (row stride 64 elements)
    const int off = (4 * grp + q) * 64 + 16 * grp + 4 * p;
    bf16x4 vr = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(img_r + off));
    bf16x4 vc = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(img_c + off));
    for (int e = 0; e < 4; ++e) { out_row[lane * 4 + e] = (float)vr[e]; out_col[lane * 4 + e] = (float)vc[e]; }
}

int main()
{
    float *dr, *dc, hr[256], hc[256];
    hipMalloc(&dr, sizeof(hr)); hipMalloc(&dc, sizeof(hc));
    k<<<1, 64>>>(dr, dc);
    hipMemcpy(hr, dr, sizeof(hr), hipMemcpyDeviceToHost);
    hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        int grp = lane >> 4, i = lane & 15;
        printf("lane %2d:", lane);
        for (int e = 0; e < 4; ++e) {
            printf(" (r%2.0f,c%2.0f)", hr[lane * 4 + e], hc[lane * 4 + e]);
            if ((int)hr[lane * 4 + e] != 4 * grp + e || (int)hc[lane * 4 + e] != 16 * grp + i) ++bad;
        }
        printf("\n");
    }
    printf("expected map (lane i of group g, element e) = (row 4g+e, col 16g+i): %s (%d mismatches)\n",
           bad ? "MISMATCH" : "OK", bad);
    return 0;
}
