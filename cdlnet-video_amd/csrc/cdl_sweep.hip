// Whole forward / reverse sweeps of the shape-generic loop from one C call: the same launches that
// cdl_analysis / cdl_synthesis / cdl_wgrad / cdl_tau_grad / cdl_prox_csr* make, enqueued back to back.
// Single-frame inference (the frame-recurrent CSR drivers, reference analyzemri.py:87-182) and crop-sized
// training (traincsr.py) run launches of a few microseconds, where a host round trip per launch
// (interpreter + ctypes, ~30 us) is what bounds the step.
#include "cdl_common.h"

static inline hipStream_t S(void *s) { return (hipStream_t)s; }
static inline size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

#define CDL_TRY(expr)              \
    do {                           \
        const int rc_ = (expr);    \
        if (rc_ != 0) return rc_;  \
    } while (0)

extern "C" {

size_t cdl_ista_scratch_floats(const cdl_geom *g)
{
    if (!cdl_geom_ok(g)) return 0;
    size_t n = (size_t)CDL_TAU_SPLITS * g->N * g->M;             // cdl_tau_grad
    n = max_sz(n, cdl_wgrad_workspace_floats(g));
    n = max_sz(n, cdl_synthesis_workspace_floats(g));
    n = max_sz(n, cdl_prox_csr_scratch_floats(g));
    n = max_sz(n, cdl_analysis_workspace_floats(g));
    n = max_sz(n, cdl_analysis_rev_workspace_floats(g));
    return n;
}

int cdl_ista_forward(const cdl_geom *g, int K, const float *yp, const float *mask, const float *tau,
                     const float *z_prev, const float *z_after, const float *gam1, const float *gam2,
                     const float *const *wA, const float *const *wB, float *const *z, float *const *r,
                     float *const *u, float *xp, float *scratch, size_t scratch_floats, void *stream)
{
    if (!cdl_geom_ok(g) || K < 1 || !yp || !tau || !wA || !wB || !z || !xp || (K > 1 && !r)) return CDL_EINVAL;
    if (z_prev ? (!gam1 || (z_after && !gam2)) : (z_after || u)) return CDL_EINVAL;
    const size_t NM = (size_t)g->N * g->M;
    for (int k = 0; k < K; ++k) {
        const float *zin = nullptr, *x = yp;
        if (k > 0) {
            CDL_TRY(cdl_synthesis_ws(g, z[k - 1], nullptr, wB[k], 1.0f, mask, yp, r[k - 1], scratch, scratch_floats,
                                     stream));
            zin = z[k - 1];
            x = r[k - 1];
        }
        const float alpha = k == 0 ? 1.0f : -1.0f;
        if (z_prev)
            CDL_TRY(cdl_analysis_prox_ws(g, x, wA[k], alpha, zin, z_prev, z_after, tau + k * NM, gam1 + k * NM,
                                         gam2 ? gam2 + k * NM : nullptr, u ? u[k] : nullptr, z[k], scratch,
                                         scratch_floats, stream));
        else
            CDL_TRY(cdl_analysis_ws(g, x, wA[k], alpha, zin, nullptr, tau + k * NM, z[k], scratch, scratch_floats,
                                    stream));
    }
    return cdl_synthesis_ws(g, z[K - 1], nullptr, wB[0], 1.0f, nullptr, nullptr, xp, scratch, scratch_floats, stream);
}

int cdl_ista_backward(const cdl_geom *g, int K, const float *yp, const float *mask, const float *c,
                      const float *z_prev, const float *z_after, const float *lam, const float *gam1,
                      const float *gam2, const float *const *wA, const float *const *wB,
                      const float *const *z, const float *const *r, const float *const *u, const float *g_xp,
                      const float *g_z, float *const *dA, float *const *dB, float *dt, float *dg1, float *dg2,
                      float *gz_prev, float *gz_after, float *gbuf0, float *gbuf1, float *q, float *scratch,
                      size_t scratch_floats, void *stream)
{
    if (!cdl_geom_ok(g) || K < 1 || !yp || !wA || !wB || !z || !dA || !dB || !dt || !gbuf0 || !gbuf1 || !q ||
        !scratch || (K > 1 && !r))
        return CDL_EINVAL;
    if (!g_xp && !g_z) return CDL_EINVAL;
    if (z_prev ? (!u || !lam || !gam1 || !dg1 || (z_after && (!gam2 || !dg2))) : (z_after != nullptr)) return CDL_EINVAL;
    if (scratch_floats < cdl_ista_scratch_floats(g)) return CDL_EINVAL;
    const size_t NM = (size_t)g->N * g->M, M = g->M;
    const size_t code = NM * (size_t)(g->D / g->sd) * (g->H / g->sh) * (g->W / g->sw);
    const size_t flen = (size_t)g->M * g->C * g->Pd * g->Ph * g->Pw;
    float *gk = gbuf0, *other = gbuf1;
    // Plain loop (no neighbour codes): the analysis that produces dL/dz_k also gates it by the support of z_k and
    // reduces the threshold gradients of iteration k (cdl_analysis_rev_ws) -- the separate gate / threshold pass re-read
    // and re-wrote the tensor the analysis had just written (3 of the reverse sweep's fat passes per iteration).
    bool gated = false;                                        // gk already gated, dt_k already written
    if (g_xp) {
        CDL_TRY(cdl_wgrad(g, z[K - 1], nullptr, g_xp, 1.0f, dB[0], scratch, scratch_floats, stream));
        if (!z_prev) {
            CDL_TRY(cdl_analysis_rev_ws(g, g_xp, wB[0], 1.0f, g_z, z[K - 1], c, dt + (size_t)(K - 1) * 2 * M,
                                        dt + (size_t)(K - 1) * 2 * M + M, gk, scratch, scratch_floats, stream));
            gated = true;
        } else
        CDL_TRY(cdl_analysis_ws(g, g_xp, wB[0], 1.0f, g_z, nullptr, nullptr, gk, scratch, scratch_floats, stream));   // B_0^T g_xp (+ g_z)
    } else {
        hipError_t e = hipMemsetAsync(dB[0], 0, flen * sizeof(float), S(stream));
        if (e != hipSuccess) return -(int)e;
        e = hipMemcpyAsync(gk, g_z, code * sizeof(float), hipMemcpyDeviceToDevice, S(stream));
        if (e != hipSuccess) return -(int)e;
    }
    for (int k = K - 1; k >= 0; --k) {
        const float *gate = nullptr;
        if (z_prev) {            // gk: dL/dz_{k+1} -> dL/du_k in place; neighbour and threshold gradients
            CDL_TRY(cdl_prox_csr_bwd(g, gk, u[k], z_prev, z_after, lam + k * NM, gam1 + k * NM,
                                     gam2 ? gam2 + k * NM : nullptr, c, gk, gz_prev, gz_after, dt + k * 2 * M,
                                     dg1 + k * 2 * M, dg2 ? dg2 + k * 2 * M : nullptr, scratch, scratch_floats,
                                     stream));
        } else if (!gated) {
            // threshold gradients, and gk gated in place by the support of z_{k+1} in the same pass: the synthesis,
            // the filter gradient and the analysis below then read no gate (3 fat reads less per iteration)
            CDL_TRY(cdl_tau_grad_gate(g, gk, z[k], c, dt + k * 2 * M, dt + k * 2 * M + M, scratch, stream));
        }
        if (k == 0) {
            CDL_TRY(cdl_wgrad(g, gk, gate, yp, 1.0f, dA[0], scratch, scratch_floats, stream));
            break;
        }
        CDL_TRY(cdl_synthesis_ws(g, gk, gate, wA[k], -1.0f, mask, nullptr, q, scratch, scratch_floats, stream));
        CDL_TRY(cdl_wgrad_pair(g, gk, r[k - 1], -1.0f, dA[k], z[k - 1], q, 1.0f, dB[k], scratch, scratch_floats, stream));   // gk is gated in place above
        if (!z_prev) {                                       // dL/dz_{k-1}, gated, with the thresholds' gradient of iteration k-1
            CDL_TRY(cdl_analysis_rev_ws(g, q, wB[k], 1.0f, gk, z[k - 1], c, dt + (size_t)(k - 1) * 2 * M,
                                        dt + (size_t)(k - 1) * 2 * M + M, other, scratch, scratch_floats, stream));
            gated = true;
        } else
        CDL_TRY(cdl_analysis_ws(g, q, wB[k], 1.0f, gk, gate, nullptr, other, scratch, scratch_floats, stream));
        float *t = gk;
        gk = other;
        other = t;
    }
    return 0;
}

}  // extern "C"
