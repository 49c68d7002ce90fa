// Shared helpers for the gfx950 kernels of libcdlnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/cdlnet_hip.h"

#define CDL_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e_ = hipGetLastError();                   \
        if (e_ != hipSuccess) return -(int)e_;               \
    } while (0)

// sign(u) * relu(|u| - t): exactly the reference's x.sign()*relu(|x|-t) (model/net.py:11-14), including
// t < 0 (|u|-t > 0 everywhere, sign(0) = 0) and non-finite values: a NaN in u or t comes out as NaN (torch's
// sign and relu both keep it; fmaxf would turn it into 0 and hide a diverging net from the trainer's nan / inf
// backtracking, train.py:113-142), +-inf stays +-inf.
__device__ __forceinline__ float cdl_shrink(float u, float t)
{
    float m = fabsf(u) - t;
    m = m > 0.0f ? m : (m != m ? m : 0.0f);                          // relu that keeps NaN
    const float s = u > 0.0f ? 1.0f : (u < 0.0f ? -1.0f : (u != u ? u : 0.0f));      // sign that keeps NaN
    return s * m;
}

// ---- CSR proximal maps (reference model/net.py:229-262), shared by cdl_prox.hip and the analysis
// epilogues.  The reference's evaluation order is kept term by term and fma contraction is off: the
// maps are discontinuous for negative thresholds, a one-ulp re-association can move an output by |t|.
__device__ __forceinline__ float cdl_sgn(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

struct cdl_prox1 {           // intermediates of prox_CSR(u, zp, lam, gam)
    float s, ls, a, tg, m, z;
};

__device__ __forceinline__ cdl_prox1 cdl_prox_csr1(float u, float zp, float lam, float gam)
{
#pragma clang fp contract(off)
    cdl_prox1 p;
    p.s = cdl_sgn(zp);
    p.ls = lam * p.s;
    p.a = (u - zp) - p.ls;                                   // u - z_prev - lambd*sign(z_prev)
    p.tg = lam * gam;
    const float inner = cdl_shrink(p.a, p.tg);
    p.m = (inner + zp) + p.ls;
    p.z = cdl_shrink(p.m, lam);
    return p;
}

struct cdl_prox2 {           // intermediates of prox_CSR_f2(u, zp, za, lam, g1, g2)
    float sp, sa_, spa, sap, a, sa, t1, b, t2, m, z;
};

__device__ __forceinline__ cdl_prox2 cdl_prox_csr2(float u, float zp, float za, float lam, float g1, float g2)
{
#pragma clang fp contract(off)
    cdl_prox2 p;
    p.sp = cdl_sgn(zp);
    p.sa_ = cdl_sgn(za);
    p.spa = cdl_sgn(zp - za);
    p.sap = cdl_sgn(za - zp);
    const float l1 = lam * g1, l2 = lam * g2;
    const float ca = (zp + lam * p.sp) + l2 * p.spa;
    const float cb = (za + lam * p.sa_) + l1 * p.sap;
    p.a = u - ca;
    p.sa = cdl_sgn(p.a);
    p.t1 = g1 * lam;
    const float inner = cdl_shrink(p.a, p.t1);
    p.b = (inner - cb) + l1 * p.sa;
    p.t2 = g2 * lam;
    const float mid = cdl_shrink(p.b, p.t2);
    p.m = (mid + cb) - l1 * p.sa;
    p.z = cdl_shrink(p.m, lam);
    return p;
}

// Optional CSR epilogue of the analysis kernels: out = prox(u; zp[, za]) and, for training, u itself.
struct cdl_prox_args {
    const float *zp, *za, *lam, *g1, *g2;   // zp == nullptr: epilogue off
    float *u_out;                           // nullable
};

__device__ __forceinline__ float cdl_prox_apply(const cdl_prox_args &px, float u, size_t idx, int row)
{
    if (px.u_out) px.u_out[idx] = u;
    return px.za ? cdl_prox_csr2(u, px.zp[idx], px.za[idx], px.lam[row], px.g1[row], px.g2[row]).z
                 : cdl_prox_csr1(u, px.zp[idx], px.lam[row], px.g1[row]).z;
}

__host__ __device__ __forceinline__ int cdl_floordiv(int a, int b)
{
    int q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

// ---- timing-ablation switches: compiled OUT of the product library ------------------------------------------
// The kernels carry run-time switches that turn parts of them off for timing experiments (results are then wrong).
// They exist only in the probe build (`make libcdlnet_hip_ablate.so`, -DCDL_ABLATE, loaded by the tools through
// CDLNET_HIP_LIB); in libcdlnet_hip.so CDL_DBG() is the constant 0, the kernels' parameter blocks have no debug
// field, and CDL_FUSED_DEBUG / CDL_DENSE_DEBUG are not read.
#ifdef CDL_ABLATE
#define CDL_DBG(flags, bit) (((flags) & (bit)) != 0)
#define CDL_DBG_FIELD(decl) decl
#define CDL_DBG_COMMA(x) , x
#else
#define CDL_DBG(flags, bit) (false)
#define CDL_DBG_FIELD(decl)
#define CDL_DBG_COMMA(x)
#endif

// ---- process-wide switches and per-device bookkeeping (cdl_options.hip) ---------------------------------
struct cdl_options {
    int mfma_analysis, mfma_synthesis, mfma_wgrad, mfma_dense;   // 0: use the fp32 VALU kernels instead (CDL_MFMA_*=0)
    int no_tiled, no_pipelined_synthesis;                        // CDL_NO_TILED, CDL_NO_PIPELINED_SYNTHESIS
    int fused_snake;                                             // CDL_FUSED_SNAKE=0: no alternating tile direction
    int fused_grid;                                              // CDL_FUSED_GRID=n: fewer persistent workgroups (probes)
    int fused_da;                                                // CDL_FUSED_DA=0: dA_k by k_wgrad2d instead of inside the reverse stage
    int fusedg_bwd_prec;                                         // CDL_FUSEDG_PREC=0|2: forces the arithmetic of the tile kernel's sweeps (experiments)
    int fusedg_strip;                                            // CDL_FUSEDG_STRIP=1: cdl_stripg.hip instead of the tile kernel k_stage_g
    int scalar_assemble;                                         // CDL_SCALAR_ASSEMBLE=1: the one-pixel-per-thread patch assemble (tests)
    int fused_debug, dense_debug;                                // -DCDL_ABLATE builds only (always 0 in the product)
};
const cdl_options &cdl_opts();                                   // snapshot of the environment, read once
bool cdl_exact_fp32();                                           // this host thread asked for the fp32 VALU tier (cdl_set_exact_fp32)
int cdl_current_device();
int cdl_cu_count();                                              // compute units of the CURRENT device
int cdl_ensure_dynamic_lds(const void *kernel, int bytes);       // per (device, kernel): keep the dynamic-LDS limit >= bytes

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
                       const float *gate, const float *tau, float *out, const cdl_prox_args &px, void *stream);
int cdl_tiled_synthesis(const cdl_geom *g, const float *z, const float *gate, const float *w, float alpha,
                        const float *mask, const float *sub, float *out, float *ws, size_t ws_floats,
                        void *stream);
size_t cdl_tiled_synthesis_ws_floats(const cdl_geom *g);
// matrix-core synthesis (cdl_synth_mfma.hip): CDL_EUNSUPPORTED / 0 floats when the shape has no such kernel
int cdl_mfma_synthesis(const cdl_geom *g, const float *z, const float *gate, const float *w, float alpha,
                       const float *mask, const float *sub, float *out, float *ws, size_t ws_floats, void *stream);
size_t cdl_mfma_synthesis_ws_floats(const cdl_geom *g);
// matrix-core analysis (cdl_analysis_mfma.hip), same convention
// cdl_dense_mfma.hip: many-channel unit-stride convolution (C >= 16 on both sides), analysis or synthesis role
size_t cdl_dense_ws_floats(const cdl_geom *g, int transpose);
size_t cdl_dense_wgrad_ws_floats(const cdl_geom *g);
int cdl_dense_wgrad(const cdl_geom *g, const float *F, const float *gate, const float *x, float alpha, float *dw,
                    float *ws, size_t ws_floats, void *stream);
int cdl_dense_conv(const cdl_geom *g, int transpose, const float *x, const float *in_gate, const float *w,
                   float alpha, const float *add, const float *add_gate, const float *mask, const float *sub,
                   const float *tau, int relu, const float *out_gate, float *out, float *ws, size_t ws_floats,
                   void *stream);
int cdl_mfma_analysis(const cdl_geom *g, const float *x, const float *w, float alpha, const float *zin,
                      const float *gate, const float *tau, float *out, const cdl_prox_args &px, float *ws,
                      size_t ws_floats, void *stream);
size_t cdl_mfma_analysis_ws_floats(const cdl_geom *g);
size_t cdl_mfma_analysis_rev_ws_floats(const cdl_geom *g);
int cdl_mfma_analysis_rev(const cdl_geom *g, const float *x, const float *w, float alpha, const float *zin,
                          const float *zsup, const float *c, float *dt0, float *dt1, float *out, float *ws,
                          size_t ws_floats, void *stream);
// matrix-core filter gradients (cdl_wgrad_mfma.hip), same convention
int cdl_mfma_wgrad(const cdl_geom *g, const float *F, const float *gate, const float *x, float alpha, float *dw,
                   float *ws, size_t ws_floats, void *stream);
int cdl_mfma_wgrad_pair(const cdl_geom *g, const float *F0, const float *x0, float alpha0, float *dw0, const float *F1,
                        const float *x1, float alpha1, float *dw1, float *ws, size_t ws_floats, void *stream);
size_t cdl_mfma_wgrad_ws_floats(const cdl_geom *g);
bool cdl_mfma_wgrad_takes(const cdl_geom *g);
int cdl_mfma_wgrad_lay(const cdl_geom *g, const float *F, const float *x, float alpha, float *dw, float *ws,
                       size_t ws_floats, int rsc, void *stream);
int cdl_mfma_wgrad_pair_lay(const cdl_geom *g, const float *F0, const float *x0, float alpha0, float *dw0,
                            const float *F1, const float *x1, float alpha1, float *dw1, float *ws, size_t ws_floats,
                            int rsc, void *stream);
int cdl_tiled_wgrad(const cdl_geom *g, const float *z, const float *gate, const float *x, float alpha,
                    float *dw, float *workspace, size_t workspace_floats, void *stream);
