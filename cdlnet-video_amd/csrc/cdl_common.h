// Shared helpers for the gfx950 kernels of libcdlnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/cdlnet_hip.h"

#define CDL_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e_ = hipGetLastError();                   \
        if (e_ != hipSuccess) return -(int)e_;               \
    } while (0)

// sign(u) * max(|u| - t, 0): exactly the reference's x.sign()*relu(|x|-t) (model/net.py:11-14),
// including t < 0 (|u|-t > 0 everywhere, sign(0) = 0).
__device__ __forceinline__ float cdl_shrink(float u, float t)
{
    float m = fmaxf(fabsf(u) - t, 0.0f);
    return u > 0.0f ? m : (u < 0.0f ? -m : 0.0f);
}

__host__ __device__ __forceinline__ int cdl_floordiv(int a, int b)
{
    int q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

static inline bool cdl_geom_ok(const cdl_geom *g)
{
    if (!g) return false;
    if (g->N <= 0 || g->C <= 0 || g->M <= 0) return false;
    if (g->D <= 0 || g->H <= 0 || g->W <= 0) return false;
    if (g->Pd <= 0 || g->Ph <= 0 || g->Pw <= 0) return false;
    if (g->sd <= 0 || g->sh <= 0 || g->sw <= 0) return false;
    if (g->pd < 0 || g->ph < 0 || g->pw < 0) return false;
    if (g->D % g->sd || g->H % g->sh || g->W % g->sw) return false;
    // the transpose must map Z*s back onto X exactly (output_padding = s-1): 2p + s - P >= 0 .. s-1
    return true;
}

// register-tiled variants (cdl_generic_tiled.hip): CDL_EUNSUPPORTED means "use the untiled kernel"
int cdl_tiled_analysis(const cdl_geom *g, const float *x, const float *w, float alpha, const float *zin,
                       const float *gate, const float *tau, float *out, void *stream);
int cdl_tiled_synthesis(const cdl_geom *g, const float *z, const float *gate, const float *w, float alpha,
                        const float *mask, const float *sub, float *out, float *ws, size_t ws_floats,
                        void *stream);
size_t cdl_tiled_synthesis_ws_floats(const cdl_geom *g);
int cdl_tiled_wgrad(const cdl_geom *g, const float *z, const float *gate, const float *x, float alpha,
                    float *dw, float *workspace, size_t workspace_floats, void *stream);
