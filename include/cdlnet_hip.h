/* cdlnet_hip.h -- C ABI of libcdlnet_hip.so: the MI355X (gfx950) kernels behind the
 * unrolled-ISTA hot path of RQLuo/CDLNet-video.
 *
 * The reference has no FFI / plugin layer (it is 100 % Python on ATen); its boundary for
 * this path is the nn.Module surface of `model/net.py`.  The Python host package
 * (`cdlnet-video_amd/`) mirrors that surface and calls the entry points below through
 * ctypes; a maintainer of the reference would bind exactly these (INTEGRATION.md shows
 * the stub).  Each entry point names the reference lines whose ATen calls it replaces.
 *
 * Conventions
 *   - plain pointers and ints only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch's caching allocator in practice), fp32 unless a name says otherwise;
 *   - tensors are contiguous, row-major, in the reference's own layouts:
 *       image-like   (N, C, D, H, W)      "thin"  (C = 1 or 3)
 *       code-like    (N, M, Dz, Hz, Wz)   "fat"   (M = 32..169), Dz = D/sd, ...
 *       filter bank  (M, C, Pd, Ph, Pw)   -- Conv{2,3}d.weight and ConvTranspose{2,3}d.weight
 *                                            share this shape (net.py:32-33, 137-142)
 *     2-D nets are the D = Pd = sd = 1, pd = 0 special case;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*, of the CURRENT device) and do
 *     not synchronise.  The compute entry points keep no state between calls and are re-entrant
 *     across streams, devices and ranks; what is process-wide is lock-protected and off the data
 *     path: a snapshot of the experiment switches read once from the environment
 *     (cdl_options_reload), per-device launch attributes, and the opt-in kernel timing of
 *     cdl_fused2d_timing (bench.py only);
 *   - return 0 on success, a negative value on error: -(hipError_t) for runtime errors,
 *     CDL_EINVAL / CDL_EUNSUPPORTED for argument errors.  Nothing throws.
 *   - in-place is allowed only where a parameter says "inout".
 */
#ifndef CDLNET_HIP_H
#define CDLNET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDL_EINVAL        (-10001)
#define CDL_EUNSUPPORTED  (-10002)

/* Geometry of one analysis/synthesis operator pair.  D,H,W are the extents of the
 * (reflect-padded) image, multiples of the stride; the code extents are D/sd, H/sh, W/sw.
 * Zero padding p = (P-1)/2 for CDLNet (net.py:32), P/2 per axis for CDLNetVideo (net.py:138). */
typedef struct cdl_geom {
    int N, C, M;
    int D, H, W;
    int Pd, Ph, Pw;
    int pd, ph, pw;
    int sd, sh, sw;
} cdl_geom;

const char *cdl_version(void);

/* Re-read the CDL_* experiment switches from the environment (they are read once, at first use, into an
 * immutable snapshot; tests and tools that flip a variable mid-process call this afterwards). */
int cdl_options_reload(void);

/* Arithmetic of the shape-generic entry points (cdl_analysis*, cdl_synthesis*, cdl_wgrad*, cdl_ista_* sweeps) called from
 * THIS host thread afterwards: on != 0 keeps them on the fp32 VALU kernels (plain fp32 FMAs: ~3e-6 of an fp64 evaluation even
 * where sums cancel), on == 0 (the default) lets them use the matrix cores (split-bf16 operands: 16-17 significant bits, which
 * a cancelling filter gradient amplifies -- DESIGN.md section 6).  Returns the previous setting.  The fused entry points
 * (cdl_fused2d_*, cdl_fusedg_*) are matrix-core kernels by construction and are not affected: a caller that wants fp32
 * end to end composes the generic ones (the Python host side does: loop.precision_scope("fp32")). */
int cdl_set_exact_fp32(int on);

/* ---- boundary of the loop: model/utils.py:5-22 (pre_process), :70-87 (pre_process_3d) -------
 * mean[n] = sum(y[n]) / (mask ? sum(mask[n]) : numel);  yp = reflect_pad(mask * (y - mean));
 * mask_p = reflect_pad(mask).  pads = {d_lo, d_hi, h_lo, h_hi, w_lo, w_hi} (floor/ceil split,
 * utils.py:35-51).  y: (N,C,D,H,W) unpadded; yp/mask_p: (N,C,D+..,H+..,W+..). */
int cdl_preprocess(const float *y, const float *mask /*nullable*/, float *yp,
                   float *mask_p /*nullable iff mask is*/, float *mean /*N*/,
                   int N, int C, int D, int H, int W, const int pads[6], void *stream);

/* model/utils.py:24-33 (post_process), :89-101: xhat = crop(xp) + mean. D,H,W = unpadded. */
int cdl_postprocess(const float *xp, const float *mean, float *xhat,
                    int N, int C, int D, int H, int W, const int pads[6], void *stream);

/* Adjoint of the crop in cdl_postprocess (autograd of utils.py:26-27): gxp = zero outside the
 * crop window, gxhat inside.  D,H,W = unpadded extents of gxhat; gxp has the padded extents. */
int cdl_postprocess_bwd(const float *gxhat, float *gxp, int N, int C, int D, int H, int W,
                        const int pads[6], void *stream);

/* tau[k,n,m] = t[k,0,m] + c[n] * t[k,1,m]   (net.py:82,85,87; c = sigma/255 or absent = 0) */
int cdl_thresholds(const float *t /*K,2,M*/, const float *c /*N, nullable*/, float *tau /*K,N,M*/,
                   int K, int N, int M, void *stream);

/* Stand-alone ST(x, t) = sign(x)*relu(|x|-t) (net.py:11-14) over a code-like tensor:
 * out[i] = ST(x[i], tau[i / per_m]) for `rows` = N*M rows of `per_m` elements.  In the nets the
 * shrinkage is fused into cdl_analysis; this exists for callers of the reference's helper. */
int cdl_shrink(const float *x, const float *tau /*rows*/, float *out, int rows, size_t per_m,
               void *stream);

/* ---- analysis half: F.conv2d / F.conv3d at net.py:85,87,200,205 and gabor.py:55 ------------
 *   acc  = alpha * corr(x ; w)                         x thin, w (M,C,P..), acc fat
 *   base = zin ? (gate ? (gate != 0 ? zin : 0) : zin) : 0
 *   out  = tau ? ST(base + acc, tau[n,m]) : base + acc,  ST(u,t) = sign(u)*max(|u|-t,0) (net.py:11-14)
 * Forward k=0: (alpha=+1, zin=NULL, tau).  Forward k>=1: (alpha=-1, zin=z_k, tau) with
 * x = mask*B_k z_k - yp.  Backward: g_k = [z_{k+1}!=0]*g_{k+1} + corr(q ; B_k) is
 * (alpha=+1, zin=g_{k+1}, gate=z_{k+1}, tau=NULL).  out must not alias zin. */
int cdl_analysis(const cdl_geom *g, const float *x, const float *w, float alpha,
                 const float *zin /*nullable*/, const float *gate /*nullable*/,
                 const float *tau /*N*M, nullable*/, float *out, void *stream);
/* cdl_analysis with scratch (cdl_analysis_workspace_floats(g) floats, 0 when no kernel wants it): lets the
 * matrix-core kernel of the shape-generic path stage its prepared filter fragments.  workspace == NULL: VALU. */
int cdl_analysis_ws(const cdl_geom *g, const float *x, const float *w, float alpha,
                    const float *zin /*nullable*/, const float *gate /*nullable*/,
                    const float *tau /*N*M, nullable*/, float *out, float *workspace, size_t workspace_floats,
                    void *stream);
size_t cdl_analysis_workspace_floats(const cdl_geom *g);
/* One step of the reverse sweep (autograd of net.py:87 / 205 through the shrinkage of the previous iteration):
 *   out = [zsup != 0] * (zin + alpha * A x),   (dt0, dt1) = cdl_tau_grad(out, zsup, c)
 * i.e. cdl_analysis_ws followed by cdl_tau_grad_gate, as ONE fat launch where the matrix-core analysis covers the
 * geometry (gate and threshold partials in its epilogue).  out must differ from zin and zsup. */
int cdl_analysis_rev_ws(const cdl_geom *g, const float *x, const float *w, float alpha, const float *zin /*nullable*/,
                        const float *zsup, const float *c /*nullable*/, float *dt0, float *dt1, float *out,
                        float *workspace, size_t workspace_floats, void *stream);
size_t cdl_analysis_rev_workspace_floats(const cdl_geom *g);

/* ---- synthesis half: F.conv_transpose2d/3d at net.py:87,90,205,210 and gabor.py:64 ----------
 *   v   = alpha * corrT( gate ? [gate!=0]*z : z ; w )   z fat, v thin, output_padding = s-1
 *   out = (mask ? mask*v : v) - (sub ? sub : 0)
 * Forward residual r_k = mask*B_k z_k - yp: (alpha=1, mask, sub=yp).  Dictionary synthesis
 * D z_K: (alpha=1, NULL, NULL).  Backward q_k = mask * (-A_k^T du): (z=g, gate=z_{k+1}, alpha=-1, mask). */
int cdl_synthesis(const cdl_geom *g, const float *z, const float *gate /*nullable*/,
                  const float *w, float alpha, const float *mask /*nullable*/,
                  const float *sub /*nullable*/, float *out, void *stream);

/* cdl_synthesis with scratch for launches too small to fill the GPU (single frames, crops): the code
 * channels are then split over extra workgroups whose partial images (cdl_synthesis_workspace_floats(g)
 * floats, 0 when no split is worthwhile) are added in a fixed order.  workspace == NULL: no split. */
int cdl_synthesis_ws(const cdl_geom *g, const float *z, const float *gate, const float *w, float alpha,
                     const float *mask, const float *sub, float *out, float *workspace,
                     size_t workspace_floats, void *stream);
size_t cdl_synthesis_workspace_floats(const cdl_geom *g);

/* ---- filter gradients (autograd of the two conv calls above; train.py:98) --------------------
 *   dw[m,c,kd,ki,kj] = alpha * sum_{n,zd,zy,zx} zg[n,m,zd,zy,zx] * x[n,c,zd*sd-pd+kd, zy*sh-ph+ki, zx*sw-pw+kj]
 * with zg = gate ? [gate!=0]*z : z.  dA_k: (z=g_{k+1}, gate=z_{k+1}, x=r_k, alpha=-1);
 * dB_k: (z=z_k, x=q_k, alpha=+1).  Deterministic (no atomics).  Pw <= 16. dw is overwritten. */
int cdl_wgrad(const cdl_geom *g, const float *z, const float *gate /*nullable*/, const float *x,
              float alpha, float *dw, float *workspace /*nullable*/, size_t workspace_floats, void *stream);
/* Two UNGATED filter gradients of one geometry -- the dA_k / dB_k pair of an iteration of the reverse sweep
 * (train.py:98 through net.py:87 / 204-207) -- as one launch where the matrix-core kernel covers the shape; otherwise two
 * cdl_wgrad calls.  Same results as two calls, same workspace (cdl_wgrad_workspace_floats). */
int cdl_wgrad_pair(const cdl_geom *g, const float *z0, const float *x0, float alpha0, float *dw0, const float *z1,
                   const float *x1, float alpha1, float *dw1, float *workspace, size_t workspace_floats,
                   void *stream);
/* Scratch (floats) that lets cdl_wgrad split the pixel range over more workgroups (two-stage,
 * fixed-order reduction).  Without a workspace a slower single-stage kernel is used. */
size_t cdl_wgrad_workspace_floats(const cdl_geom *g);

/* Threshold gradients of one iteration: with du = [zout!=0]*g,
 *   dt0[m] = -sum_{n,pix} sign(zout)*du,   dt1[m] = -sum_n c[n] * sum_pix sign(zout)*du
 * (c NULL -> dt1 = 0).  scratch: CDL_TAU_SPLITS*N*M floats (rows are split over workgroups and folded in a fixed
 * order). dt0/dt1 are overwritten. */
#define CDL_TAU_SPLITS 16
int cdl_tau_grad(const cdl_geom *g, const float *gup, const float *zout, const float *c /*N, nullable*/,
                 float *dt0 /*M*/, float *dt1 /*M*/, float *scratch /*CDL_TAU_SPLITS*N*M*/, void *stream);
/* Same, and gup is gated IN PLACE (gup[i] = 0 where zout[i] == 0) in the same pass: the reverse sweep's three
 * consumers of the gated gradient then need no gate (one fat read each less). */
int cdl_tau_grad_gate(const cdl_geom *g, float *gup /*inout*/, const float *zout, const float *c /*N, nullable*/,
                      float *dt0 /*M*/, float *dt1 /*M*/, float *scratch /*CDL_TAU_SPLITS*N*M*/, void *stream);

/* ---- CSR temporal variants --------------------------------------
 * prox_CSR / prox_CSR_f2 of model/net.py:229-262, the shrinkage that CDLNet_CSR.forward
 * (net.py:438-455) and CDLNet_CSRf2.forward (net.py:545-563) apply instead of ST when a neighbour
 * frame's code is supplied.  u, z_prev, z_after, out are fat (N,M,D/sd,H/sh,W/sw); lam, gam1, gam2 are
 * per (sample, channel) like tau (N*M, from cdl_thresholds).  z_after == NULL selects
 *   out = ST(ST(u - z_prev - lam*sign(z_prev), lam*gam1) + z_prev + lam*sign(z_prev), lam)
 * and z_after != NULL the three-level map with both neighbours.  The reference's evaluation order is
 * kept term by term (no fma): the maps are discontinuous for negative thresholds.  out may alias u. */
int cdl_prox_csr(const cdl_geom *g, const float *u, const float *z_prev, const float *z_after /*nullable*/,
                 const float *lam, const float *gam1, const float *gam2 /*nullable iff z_after is*/,
                 float *out, void *stream);

/* cdl_analysis with the CSR map as its epilogue (one fat write + read per iteration less than
 * cdl_analysis followed by cdl_prox_csr):  u = zin + alpha*corr(x ; w),  out = prox(u; z_prev[, z_after]),
 * and u itself to u_out when it is not NULL (the reverse sweep needs it).  Same bits as the two-call form. */
int cdl_analysis_prox(const cdl_geom *g, const float *x, const float *w, float alpha,
                      const float *zin /*nullable*/, const float *z_prev, const float *z_after /*nullable*/,
                      const float *lam, const float *gam1, const float *gam2 /*nullable*/,
                      float *u_out /*nullable*/, float *out, void *stream);
int cdl_analysis_prox_ws(const cdl_geom *g, const float *x, const float *w, float alpha,
                         const float *zin /*nullable*/, const float *z_prev, const float *z_after /*nullable*/,
                         const float *lam, const float *gam1, const float *gam2 /*nullable*/,
                         float *u_out /*nullable*/, float *out, float *workspace, size_t workspace_floats,
                         void *stream);

/* Reverse of cdl_prox_csr as autograd differentiates the reference expression (sign() has zero
 * gradient): gu = dL/du (may alias gz); gz_prev / gz_after (nullable) are ACCUMULATED into, because a
 * neighbour code feeds all K iterations; dlam, dgam1, dgam2 are (2,M) and receive
 * [sum_n s, sum_n c[n]*s] of the per-(n,m) threshold sums, i.e. the gradients of t[k], g1[k], g2[k]
 * under lam = t[k,0] + c*t[k,1] (net.py:444).  scratch: cdl_prox_csr_scratch_floats(g) floats. */
int cdl_prox_csr_bwd(const cdl_geom *g, const float *gz, const float *u, const float *z_prev,
                     const float *z_after /*nullable*/, const float *lam, const float *gam1,
                     const float *gam2 /*nullable*/, const float *c /*N, nullable*/, float *gu,
                     float *gz_prev /*nullable*/, float *gz_after /*nullable*/, float *dlam /*2*M*/,
                     float *dgam1 /*2*M*/, float *dgam2 /*2*M, nullable*/, float *scratch,
                     size_t scratch_floats, void *stream);
size_t cdl_prox_csr_scratch_floats(const cdl_geom *g);

/* ---- blind noise-level estimate --------------------------------------
 * model/nle.py:17-27 (nle_mad) with the filter of model/wvlt.py:13-41: sigma_hat[n] = median over (C,H',W') of
 * |conv2d(y, flip(outer(dec_hi, dec_hi)) of 'bior4.4', stride 2, groups C)| / 0.6745, the lower median as
 * torch.median returns it (exact selection, no sort).  y (N,C,H,W) in the reference's [0,1] scale, H,W >= 10;
 * callers multiply by 255 (analyze.py:139).  scratch: cdl_nle_mad_scratch_floats(N,C,H,W) floats. */
size_t cdl_nle_mad_scratch_floats(int N, int C, int H, int W);
int cdl_nle_mad(const float *y, float *sigma_hat /*N*/, float *scratch, size_t scratch_floats, int N, int C,
                int H, int W, void *stream);

/* ---- ResidualBlock of CDLNetVideo(residual=True) -------------------------
 * model/net.py:105-120, applied to the code after every iteration (net.py:199-207):
 *     h = relu(conv1(x)),  out = relu(conv2(h) + x),   conv*: Conv3d(M, M, P, stride 1, padding P/2, bias=False)
 * g: a geometry with C == M (the block's channels), unit strides, (D,H,W) the code extents; w1, w2 (M,M,Pd,Ph,Pw).
 * forward writes h (kept for the reverse pass) and out; backward takes g_out = dL/dout and writes dx, dw1, dw2,
 * using dh (code-sized scratch).  x, h, out, dh, dx are (N,M,D,H,W).  scratch: cdl_residual_scratch_floats(g). */
size_t cdl_residual_scratch_floats(const cdl_geom *g);
int cdl_residual_forward(const cdl_geom *g, const float *x, const float *w1, const float *w2, float *h, float *out,
                         float *scratch, size_t scratch_floats, void *stream);
int cdl_residual_backward(const cdl_geom *g, const float *x, const float *h, const float *out, const float *w1,
                          const float *w2, const float *g_out, float *dx, float *dw1, float *dw2, float *dh,
                          float *scratch, size_t scratch_floats, void *stream);

/* ---- whole sweeps of the shape-generic loop in one call ------------------------------------------
 * The same launches as K x (cdl_synthesis_ws, cdl_analysis | cdl_analysis_prox) + the final synthesis,
 * resp. the reverse sweep (cdl_tau_grad | cdl_prox_csr_bwd, cdl_synthesis_ws, 2 x cdl_wgrad, cdl_analysis
 * per iteration), enqueued from C: single frames and crops are launch-bound, a host round trip per
 * launch would dominate.  Pointer tables are HOST arrays of device pointers.
 * Forward (net.py:76-92 / 192-212 / 426-463 / 525-568): tau (K,N,M) thresholds (CSR: lam); z_prev NULL =
 * plain ST loop, else the CSR map with gam1 (K,N,M) [and z_after, gam2]; z[k] receives z_{k+1}, r[k]
 * receives r_{k+1} (k < K-1), u[k] (CSR, nullable table) receives u_k; entries may alias ping-pong
 * buffers when nothing is kept, as long as z[k] != z[k-1].  xp receives D z_K.  scratch (nullable):
 * cdl_ista_scratch_floats(g) floats.
 * Backward: z, r, u as saved by the forward; g_xp = dL/d(D z_K) and / or g_z = dL/dz_K; writes dA[k],
 * dB[k], dt (K,2,M) [, dg1, dg2 (K,2,M)], accumulates gz_prev / gz_after (nullable); gbuf0, gbuf1 fat
 * scratch, q thin scratch, scratch as above (required). */
size_t cdl_ista_scratch_floats(const cdl_geom *g);
int cdl_ista_forward(const cdl_geom *g, int K, const float *yp, const float *mask /*nullable*/,
                     const float *tau, const float *z_prev /*nullable*/, const float *z_after /*nullable*/,
                     const float *gam1 /*nullable*/, const float *gam2 /*nullable*/,
                     const float *const *wA, const float *const *wB, float *const *z, float *const *r,
                     float *const *u /*nullable*/, float *xp, float *scratch, size_t scratch_floats,
                     void *stream);
int cdl_ista_backward(const cdl_geom *g, int K, const float *yp, const float *mask /*nullable*/,
                      const float *c /*N, nullable*/, const float *z_prev /*nullable*/,
                      const float *z_after /*nullable*/, const float *lam, const float *gam1, const float *gam2,
                      const float *const *wA, const float *const *wB, const float *const *z,
                      const float *const *r, const float *const *u /*CSR*/, const float *g_xp /*nullable*/,
                      const float *g_z /*nullable*/, float *const *dA, float *const *dB, float *dt,
                      float *dg1 /*CSR*/, float *dg2 /*CSR f2*/, float *gz_prev /*nullable*/,
                      float *gz_after /*nullable*/, float *gbuf0, float *gbuf1, float *q, float *scratch,
                      size_t scratch_floats, void *stream);

/* model/solvers.py:24-28 (uball_project) applied by net.py:72-73,189-190: every filter
 * (consecutive `flen` floats) with l2 norm > 1 is scaled onto the unit sphere.  w inout. */
int cdl_project_filters(float *w, int nfilters, int flen, void *stream);
/* The same for `nbanks` banks of identical shape (HOST array of device pointers) in one launch per 64 banks:
 * net.py:72-73 projects A[k] and B[k] for every k. */
int cdl_project_filter_banks(float *const *w, int nbanks, int nfilters, int flen, void *stream);

/* model/gabor.py:7-28,46-51 (gabor_kernel, get_filter): w[m,c,i,j] = sum_o alpha[o,m,c] *
 * exp(-(a0*(i-x0))^2 - (a1*(j-x0))^2) * cos(sgn*(w00*(i-x0) + w01*(j-x0) + psi)),
 * x0=(P-1)/2, sgn=-1 for the analysis ("transpose") filter.  alpha (O,M,C), a,w0 (O,M,C,2), psi (O,M,C). */
int cdl_gabor_filters(const float *alpha, const float *a, const float *w0, const float *psi,
                      float *w /*M,C,P,P*/, int order, int M, int C, int P, int transpose, void *stream);

/* Every bank of a net in ONE launch (GDLNet.forward re-synthesises its 2K banks per call, net.py:659-675):
 * bank k reads alpha[k], a[k], w0[k], psi[k] (aliased pointers are fine: shared parameters, net.py:607-622), writes
 * w[k]; transpose[k] != 0 selects the analysis filter.  _bwd: grads[k] receives the block
 * [dalpha (order*M*C) | da (2*order*M*C) | dw0 (2*order*M*C) | dpsi (order*M*C)] for upstream dw[k] (zeros when
 * dw[k] is NULL); the caller sums the blocks of aliased parameters. */
int cdl_gabor_filter_banks(int nbanks, const float *const *alpha, const float *const *a, const float *const *w0,
                           const float *const *psi, const int *transpose, float *const *w, int order, int M, int C,
                           int P, void *stream);
int cdl_gabor_filter_banks_bwd(int nbanks, const float *const *alpha, const float *const *a, const float *const *w0,
                               const float *const *psi, const int *transpose, const float *const *dw,
                               float *const *grads, int order, int M, int C, int P, void *stream);
/* Backward of cdl_gabor_filters: given dw (M,C,P,P) writes dalpha, da, dw0, dpsi (overwritten). */
int cdl_gabor_filters_bwd(const float *alpha, const float *a, const float *w0, const float *psi,
                          const float *dw, float *dalpha, float *da, float *dw0, float *dpsi,
                          int order, int M, int C, int P, int transpose, void *stream);

/* OR-ed into the `precision` argument of cdl_fused2d_iter_fwd / _stage_bwd / _wgrad: walk the tiles from
 * the last to the first.  Stage results are identical; the filter gradients' per-workgroup partial sums
 * are grouped differently (still deterministic).  The whole-sweep entry points alternate it per launch. */
#define CDL_TILES_REVERSED 16

/* Layouts of the fat (code-like) operands of the fused path, OR-ed into `precision` as well:
 *   CDL_LAY_NCHW   (N, M, H, W) fp32 -- the reference's layout; what leaves or enters a sweep (z_K, user gradients)
 *   CDL_LAY_BLK    pixel-blocked fp32 [n][y][ceil(W/32)][M/4][32 px][4 ch]: 16 contiguous bytes per lane and
 *                  register quad, 8 KiB contiguous per 32-pixel row block (measured 5.65 vs 5.08 TB/s for a pure
 *                  load -> store stream of the cfg2 code tensor, tools/probes/probe_stream.hip); for tensors that
 *                  stay INSIDE a sweep (z_1..z_{K-1}, du_k).  Bit-identical values to CDL_LAY_NCHW.
 *   CDL_LAY_BLK16  the same blocking with bf16 elements: opt-in reduced-precision STORAGE (half the fat bytes;
 *                  arithmetic and accumulation stay as `precision` says; the 1e-5 parity gate does not apply).
 * cdl_fused2d_iter_fwd / _stage_bwd: CDL_LAYOUT_IN = zin / base, CDL_LAYOUT_OUT = zout / du_out (pairs: equal
 * layouts, or NCHW on exactly one side).  cdl_fused2d_wgrad: CDL_LAYOUT_IN = X0 and X1.  cdl_fused2d_forward /
 * _backward: CDL_LAYOUT_IN = the layout of z[0..K-2] and of the du ping-pong buffers (z[K-1], g_z: NCHW). */
#define CDL_LAY_NCHW  0
#define CDL_LAY_BLK   1
#define CDL_LAY_BLK16 2
/* cdl_fusedg_* on the strip kernel's shapes (one image channel, stride 1 / 2, M <= 192) only: row-strip channel-major
 * fp32 [n][code row][ceil(Wz/32)][M][32 columns] -- what a wave moves per code row is one contiguous run of M * 128
 * bytes.  Same values as CDL_LAY_NCHW; for the codes that stay inside a sweep. */
#define CDL_LAY_RSC   3
#define CDL_LAYOUT_IN(l)  ((l) << 5)
#define CDL_LAYOUT_OUT(l) ((l) << 7)

/* ==== fused MFMA path (cdl_fused2d.hip): 2-D, C = 1, stride 1, odd P <= 7, M in {32, 64} =========
 * One launch per unrolled iteration replaces the whole body of net.py:87
 *     z = ST(z - A_k(mask*B_k(z) - yp), tau_k)
 * with the fusion boundary moved to the one-channel residual:
 *     cdl_fused2d_iter_fwd :  r_k, z_k  ->  z_{k+1} = ST(z_k + sgn * A_k r_k, tau_k)   (fat, written once)
 *                                           patches = partial B_next z_{k+1}           (thin)
 *     cdl_fused2d_assemble :  patches  ->  r_{k+1} = mask * (B_next z_{k+1}) - yp      (thin)
 * For k = 0 pass r = yp, zin = NULL, sgn = +1 (net.py:85); for k >= 1 sgn = -1.  For the last
 * iteration B_next is D = B_0 and assemble(mask = sub = NULL) yields D z_K (net.py:90).
 * precision: 0 = split-bf16 (hi+lo operands, 3 MFMAs per product, fp32-grade), 1 = plain bf16, 2 = split-bf16 with all
 * four products (exact fp32 products; for objectives that difference two forward passes, e.g. MC-SURE). */
int cdl_fused2d_supported(const cdl_geom *g);              /* 1 if this geometry has a fused kernel */
size_t cdl_fused2d_frag_bytes(int M);                      /* bytes of one prepared (A_k, B_next) pair */
size_t cdl_fused2d_patch_floats(const cdl_geom *g);        /* floats in the patch workspace */
size_t cdl_fused2d_code_bytes(const cdl_geom *g, int layout);   /* bytes of one code tensor in CDL_LAY_* */
/* fp32 filters (M,1,P,P) -> bf16 hi/lo MFMA operand fragments for one launch (A_k with B_next). */
int cdl_fused2d_prep(const float *wA, const float *wB, void *frags, int M, int P, void *stream);
/* map_out (nullable, cdl_fused2d_map_words(g) words): support / sign bit planes of z_{k+1} for the reverse
 * sweep (2 bits per code element, written by the same launch). */
int cdl_fused2d_iter_fwd(const cdl_geom *g, const float *r, const float *zin /*nullable*/,
                         const float *tau /*N*M*/, const void *frags, float sgn, float *zout,
                         float *patches, unsigned *map_out /*nullable*/, int precision, void *stream);
/* The map of a code tensor, (N,4,H,W) 32-bit words: plane 2h holds [z != 0], plane 2h+1 the sign bit, bit
 * 16R + v of a word = channel 32R + 8(v>>2) + 4h + (v&3) of that pixel (the MFMA accumulator layout).  The
 * reverse stage reads it instead of the fat z_{k+1}: 16 B per pixel against 4*M.  cdl_fused2d_support_map
 * builds it from a tensor for callers that did not get it from cdl_fused2d_iter_fwd. */
size_t cdl_fused2d_map_words(const cdl_geom *g);
int cdl_fused2d_support_map(const cdl_geom *g, const float *z, unsigned *map, void *stream);
/* out = (mask ? mask : 1) * alpha * (sum of the overlapping patches) - (sub ? sub : 0) */
int cdl_fused2d_assemble(const cdl_geom *g, const float *patches, const float *mask /*nullable*/,
                         const float *sub /*nullable*/, float alpha, float *out, void *stream);

/* ---- fused reverse sweep (what loss.backward() does through ATen in the reference, train.py:98) ----
 * One stage per iteration k = K-1 .. 0, same kernel skeleton as the forward launch:
 *     g_{k+1} = base + corr(thin ; W1)          base = du_{k+1} (or dL/dz_K, nullable), thin = q_{k+1} (or
 *                                               dL/d(D z_K)), W1 = B_{k+1} (or B_0)
 *     du_k    = [z_{k+1} != 0] * g_{k+1}        support and sign from `map` -> du_out (fat, written once)
 *     dtau    : per-workgroup partials of -sum sign(z_{k+1}) * du_k     -> dtau_partial (tiles x M)
 *     patches : partial W2^T du_k (W2 = A_k)    -> cdl_fused2d_assemble(alpha = -1, mask) gives q_k
 * frags = cdl_fused2d_prep(W1, W2).  do_synth = 0 for k = 0 (no q_0 is needed). */
size_t cdl_fused2d_tiles(const cdl_geom *g);               /* workgroups (= dtau_partial rows) per launch */
int cdl_fused2d_stage_bwd(const cdl_geom *g, const float *thin, const float *base /*nullable*/,
                          const unsigned *map /*of z_{k+1}*/, const void *frags, float *du_out,
                          float *patches, float *dtau_partial, int do_synth, int precision, void *stream);
/* The same stage with the analysis-filter gradient of the iteration riding in it (what autograd of net.py:87 computes from
 * du_k and r_k): dA = alpha * sum_px du_out (x) im2col(r2) -- cdl_fused2d_wgrad's product, taken from the stage's registers
 * instead of a second fat read of du_out.  du_out / patches / dtau_partial are bit-identical to cdl_fused2d_stage_bwd's;
 * workspace: cdl_fused2d_wgrad_workspace_floats(g).  The whole-sweep entry point cdl_fused2d_backward uses this form
 * (5.1 instead of 6.1 fat passes per iteration; CDL_FUSED_DA=0 restores the two-launch form for A/B runs). */
int cdl_fused2d_stage_bwd_da(const cdl_geom *g, const float *thin, const float *base /*nullable*/,
                             const unsigned *map, const void *frags, float *du_out, float *patches,
                             float *dtau_partial, int do_synth, const float *r2 /*(N,1,H,W)*/, float alpha,
                             float *dA /*(M,1,P,P)*/, float *workspace, int precision, void *stream);
/* dt0[m] = sum over workgroups; dt1[m] = sum_n c[n] * (sum over the workgroups of image n); c nullable */
int cdl_fused2d_dtau_reduce(const cdl_geom *g, const float *dtau_partial, const float *c /*N*/,
                            float *dt0 /*M*/, float *dt1 /*M*/, void *stream);

/* Filter gradients on the matrix cores: up to two independent reductions per launch
 *     dw_a[m,i,j] = alpha_a * sum_{n,y,x} Xa[n,m,y,x] * Ta[n,y-p+i,x-p+j]     (a = 0, 1; either may be absent)
 * e.g. (Xa,Ta,alpha) = (du_k, r_k, -1) -> dA_k and (z_k, q_k, +1) -> dB_k.  Two-stage and deterministic:
 * per-workgroup partial sums in `workspace`, then a fixed-order reduction.  workspace_floats() floats. */
size_t cdl_fused2d_wgrad_workspace_floats(const cdl_geom *g);
int cdl_fused2d_wgrad(const cdl_geom *g, const float *X0, const float *T0, float alpha0, float *dw0,
                      const float *X1, const float *T1, float alpha1, float *dw1,
                      float *workspace, int precision, void *stream);

/* ---- whole sweeps in one call (same launches as above, enqueued from C: no per-launch host cost) ----
 * Pointer tables are HOST arrays of device pointers.  Forward: z[k] receives z_{k+1} (K entries; entries
 * may alias two ping-pong buffers when nothing is kept for training, as long as z[k] != z[k-1]); r[k]
 * receives r_{k+1} for k < K-1 (same aliasing rule); maps (nullable table) receives the bit map of z_{k+1};
 * xp receives D z_K.  frags: K * cdl_fused2d_frag_bytes(M) bytes (every pair is prepared up front, one launch).
 * Backward (net.py forward lines in reverse): z[k] = z_{k+1}, r[k] = r_{k+1}, maps[k] as saved by the forward,
 * g_xp = dL/d(D z_K), g_z = dL/dz_K or NULL; writes dA[k], dB[k] (filter shapes) and dt (K,2,M);
 * du0/du1 fat scratch, q thin scratch, dtau_partial (tiles x M), wgrad_ws (workspace_floats). */
int cdl_fused2d_forward(const cdl_geom *g, int K, const float *yp, const float *mask /*nullable*/,
                        const float *tau /*K,N,M*/, const float *const *wA, const float *const *wB,
                        float *const *z, float *const *r, unsigned *const *maps /*nullable*/, float *xp,
                        void *frags, float *patches, int precision, void *stream);
int cdl_fused2d_backward(const cdl_geom *g, int K, const float *yp, const float *mask /*nullable*/,
                         const float *c /*N, nullable*/, const float *const *wA, const float *const *wB,
                         const float *const *z, const float *const *r, const unsigned *const *maps,
                         const float *g_xp, const float *g_z /*nullable*/, float *const *dA, float *const *dB,
                         float *dt,
                         float *du0, float *du1, float *q, void *frags, float *patches,
                         float *dtau_partial, float *wgrad_ws, int precision, void *stream);

/* ==== fused MFMA path for the other shapes (cdl_fusedg.hip): any C, 2-D / 3-D, unit stride, square planes ====
 * P in {3,5,7}, odd Pd with C*Pd in {1,3,5,7}, M <= 64 -- CDLNetVideo.forward's loop body (net.py:204-207) and
 * CDLNet.forward's (net.py:86-87) with C = 3 + mask (JDD).  Same fusion boundary and call structure as the
 * cdl_fused2d_* family above: one launch per iteration takes (r_k thin, z_k fat) to (z_{k+1} fat, patches of
 * B_next z_{k+1}); cdl_fusedg_assemble sums the patches over tiles and depth taps and applies alpha, mask, -sub.
 * Codes in the reference's (N,M,D,H,W) layout; map = (N,4,D,H,W) words as cdl_fused2d_support_map describes.
 * precision: 0 (split-bf16 x3), or 2 (split-bf16 x4: all four products) on the tile kernel; CDL_TILES_REVERSED may be
 * OR-ed in. */
int cdl_fusedg_supported(const cdl_geom *g);
/* Profiling hook (not part of the reference surface): while buf != NULL every cdl_fusedg stage launch records
 * s_memtime stamps of its workgroup 0 into buf[8 waves][256] (16 KB of device memory).  NULL switches it off. */
int cdl_fusedg_set_timeline(void *buf);
size_t cdl_fusedg_frag_bytes(const cdl_geom *g);           /* bytes of one prepared (A_k, B_next) pair */
size_t cdl_fusedg_patch_floats(const cdl_geom *g);
size_t cdl_fusedg_tiles(const cdl_geom *g);                /* workgroup tiles (= dtau_partial rows) per launch */
size_t cdl_fusedg_map_words(const cdl_geom *g);
/* Layout for the codes that stay inside a sweep (z[0..K-2], the du buffers): CDL_LAY_RSC where a strip kernel takes the
 * geometry and -- when a reverse sweep will follow (training != 0) -- the matrix-core filter-gradient kernel does too,
 * CDL_LAY_NCHW otherwise; floats of one code tensor in a
 * layout.  cdl_fusedg_forward / _backward take it as CDL_LAYOUT_IN(layout) in `precision`; the single-stage entry points
 * take CDL_LAYOUT_IN (zin / base) and CDL_LAYOUT_OUT (zout / du_out). */
int cdl_fusedg_code_layout(const cdl_geom *g, int training);
size_t cdl_fusedg_code_floats(const cdl_geom *g, int layout);
int cdl_fusedg_prep(const cdl_geom *g, const float *wA, const float *wB, void *frags, void *stream);
int cdl_fusedg_iter_fwd(const cdl_geom *g, const float *r, const float *zin /*nullable*/, const float *tau /*N,M*/,
                        const void *frags, float sgn, float *zout, float *patches, unsigned *map_out /*nullable*/,
                        int precision, void *stream);
int cdl_fusedg_stage_bwd(const cdl_geom *g, const float *thin, const float *base /*nullable*/, const unsigned *map,
                         const void *frags, float *du_out, float *patches /*nullable iff !do_synth*/,
                         float *dtau_partial /*tiles,M*/, int do_synth, int precision, void *stream);
int cdl_fusedg_assemble(const cdl_geom *g, const float *patches, const float *mask /*nullable*/,
                        const float *sub /*nullable*/, float alpha, float *out, void *stream);
int cdl_fusedg_dtau_reduce(const cdl_geom *g, const float *dtau_partial, const float *c /*N, nullable*/, float *dt0,
                           float *dt1, void *stream);
/* Whole sweeps (arguments as cdl_fused2d_forward / _backward).  The reverse sweep takes the filter gradients
 * dA_k = -du_k (x) r_k, dB_k = z_k (x) q_k from cdl_wgrad: wgrad_ws = cdl_wgrad_workspace_floats(g) floats. */
int cdl_fusedg_forward(const cdl_geom *g, int K, const float *yp, const float *mask /*nullable*/, const float *tau,
                       const float *const *wA, const float *const *wB, float *const *z, float *const *r,
                       unsigned *const *maps /*nullable*/, float *xp, void *frags, float *patches, int precision,
                       void *stream);
int cdl_fusedg_backward(const cdl_geom *g, int K, const float *yp, const float *mask /*nullable*/,
                        const float *c /*nullable*/, const float *const *wA, const float *const *wB,
                        const float *const *z, const float *const *r, const unsigned *const *maps, const float *g_xp,
                        const float *g_z /*nullable*/, float *const *dA, float *const *dB, float *dt, float *du0,
                        float *du1, float *q, void *frags, float *patches, float *dtau_partial, float *wgrad_ws,
                        size_t wgrad_ws_floats, int precision, void *stream);

/* Per-kernel timing inside the fused sweeps: cdl_fused2d_timing(1) starts collecting HIP-event pairs around
 * every forward stage (class 0; the k = 0 launch, which reads no code, is class 3), reverse stage (1) and
 * filter-gradient (2) launch of the sweeps on their own stream, cdl_fused2d_timing(0) stops; _read synchronises and returns the summed milliseconds and launch counts
 * per class.  For measurement (bench.py); off by default, no events are recorded then. */
int cdl_fused2d_timing(int enable);
int cdl_fused2d_timing_read(double *ms_sum /*4*/, int *count /*4*/);

#ifdef __cplusplus
}
#endif
#endif /* CDLNET_HIP_H */
