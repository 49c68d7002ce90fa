/* TEST INFRASTRUCTURE -- plain-C restatement of the unrolled-ISTA forward loop.
 *
 * Never linked into the product library.  Independent of ATen: direct loops with
 * double accumulation, used by tests/ to cross-check oracle/cdl_oracle.py (which is
 * itself pinned to the reference by tests/golden) at small sizes, and as the checker
 * behind __graft_entry__.smoke().
 *
 * Follows /root/reference:
 *   model/net.py:11-14   ST(x,t) = sign(x)*relu(|x|-t)
 *   model/net.py:85-90   z1 = ST(A0 yp, tau0); z_{k+1} = ST(z_k - A_k(mask*B_k z_k - yp), tau_k);
 *                        xp = B_0 z_K
 *   model/net.py:32-33, 137-142   A = strided correlation C->M, zero padding p;
 *                        B = its transpose M->C with output_padding s-1 (so B: Z*s -> X exactly)
 * 2-D nets are the D=1, Pd=1, pd=0, sd=1 special case of the 3-D loops.
 *
 * Layouts (all contiguous, row-major): image (N,C,D,H,W), code (N,M,Dz,Hz,Wz) with
 * Dz=D/sd etc., filters (K,M,C,Pd,Ph,Pw) for both banks (Conv out-major / ConvT in-major
 * store the same shape), tau (K,N,M).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int N, C, M;
    int D, H, W;        /* padded image extent (multiples of the stride) */
    int Pd, Ph, Pw;     /* filter extent */
    int pd, ph, pw;     /* zero padding  */
    int sd, sh, sw;     /* stride        */
} geom_t;

static float shrink(float x, float t)
{
    float m = fabsf(x) - t;
    if (m < 0.0f) m = 0.0f;
    return (x > 0.0f) ? m : ((x < 0.0f) ? -m : 0.0f);
}

/* out[n,m,zd,zy,zx] = sum_{c,kd,ki,kj} x[n,c,zd*sd-pd+kd, zy*sh-ph+ki, zx*sw-pw+kj] * w[m,c,kd,ki,kj] */
static void analysis(const geom_t *g, const float *x, const float *w, float *out)
{
    int Dz = g->D / g->sd, Hz = g->H / g->sh, Wz = g->W / g->sw;
    for (int n = 0; n < g->N; ++n)
    for (int m = 0; m < g->M; ++m)
    for (int zd = 0; zd < Dz; ++zd)
    for (int zy = 0; zy < Hz; ++zy)
    for (int zx = 0; zx < Wz; ++zx) {
        double acc = 0.0;
        for (int c = 0; c < g->C; ++c)
        for (int kd = 0; kd < g->Pd; ++kd) {
            int d = zd * g->sd - g->pd + kd;
            if (d < 0 || d >= g->D) continue;
            for (int ki = 0; ki < g->Ph; ++ki) {
                int y = zy * g->sh - g->ph + ki;
                if (y < 0 || y >= g->H) continue;
                for (int kj = 0; kj < g->Pw; ++kj) {
                    int xx = zx * g->sw - g->pw + kj;
                    if (xx < 0 || xx >= g->W) continue;
                    size_t xi = ((((size_t)n * g->C + c) * g->D + d) * g->H + y) * g->W + xx;
                    size_t wi = ((((size_t)m * g->C + c) * g->Pd + kd) * g->Ph + ki) * g->Pw + kj;
                    acc += (double)x[xi] * (double)w[wi];
                }
            }
        }
        out[((((size_t)n * g->M + m) * Dz + zd) * Hz + zy) * Wz + zx] = (float)acc;
    }
}

/* out[n,c,d,y,x] = sum_{m,kd,ki,kj : (d+pd-kd) % sd == 0 ...} z[n,m,(d+pd-kd)/sd,..] * w[m,c,kd,ki,kj] */
static void synthesis(const geom_t *g, const float *z, const float *w, float *out)
{
    int Dz = g->D / g->sd, Hz = g->H / g->sh, Wz = g->W / g->sw;
    for (int n = 0; n < g->N; ++n)
    for (int c = 0; c < g->C; ++c)
    for (int d = 0; d < g->D; ++d)
    for (int y = 0; y < g->H; ++y)
    for (int x = 0; x < g->W; ++x) {
        double acc = 0.0;
        for (int kd = 0; kd < g->Pd; ++kd) {
            int td = d + g->pd - kd;
            if (td < 0 || td % g->sd) continue;
            int zd = td / g->sd;
            if (zd >= Dz) continue;
            for (int ki = 0; ki < g->Ph; ++ki) {
                int ty = y + g->ph - ki;
                if (ty < 0 || ty % g->sh) continue;
                int zy = ty / g->sh;
                if (zy >= Hz) continue;
                for (int kj = 0; kj < g->Pw; ++kj) {
                    int tx = x + g->pw - kj;
                    if (tx < 0 || tx % g->sw) continue;
                    int zx = tx / g->sw;
                    if (zx >= Wz) continue;
                    for (int m = 0; m < g->M; ++m) {
                        size_t zi = ((((size_t)n * g->M + m) * Dz + zd) * Hz + zy) * Wz + zx;
                        size_t wi = ((((size_t)m * g->C + c) * g->Pd + kd) * g->Ph + ki) * g->Pw + kj;
                        acc += (double)z[zi] * (double)w[wi];
                    }
                }
            }
        }
        out[((((size_t)n * g->C + c) * g->D + d) * g->H + y) * g->W + x] = (float)acc;
    }
}

/* Full loop. mask may be NULL (scalar 1). Returns 0, or -1 on allocation failure.
 * z_out: (N,M,Dz,Hz,Wz) final code; xp_out: (N,C,D,H,W) = B_0 z_K (before unpad / +mean). */
int cdl_oracle_forward(int K, const int *geom15, const float *yp, const float *mask,
                       const float *wA, const float *wB, const float *tau,
                       float *z_out, float *xp_out)
{
    geom_t g;
    memcpy(&g, geom15, sizeof(g));
    size_t Dz = g.D / g.sd, Hz = g.H / g.sh, Wz = g.W / g.sw;
    size_t zsz = (size_t)g.N * g.M * Dz * Hz * Wz, per_m = Dz * Hz * Wz;
    size_t xsz = (size_t)g.N * g.C * g.D * g.H * g.W;
    size_t wsz = (size_t)g.M * g.C * g.Pd * g.Ph * g.Pw;
    float *u = (float *)malloc(zsz * sizeof(float));
    float *r = (float *)malloc(xsz * sizeof(float));
    if (!u || !r) { free(u); free(r); return -1; }

    analysis(&g, yp, wA, u);
    for (size_t i = 0; i < zsz; ++i)
        z_out[i] = shrink(u[i], tau[i / per_m]);               /* tau[0][n][m] */
    for (int k = 1; k < K; ++k) {
        synthesis(&g, z_out, wB + k * wsz, r);
        for (size_t i = 0; i < xsz; ++i)
            r[i] = (mask ? mask[i] * r[i] : r[i]) - yp[i];
        analysis(&g, r, wA + k * wsz, u);
        const float *tk = tau + (size_t)k * g.N * g.M;
        for (size_t i = 0; i < zsz; ++i)
            z_out[i] = shrink(z_out[i] - u[i], tk[i / per_m]);
    }
    synthesis(&g, z_out, wB, xp_out);
    free(u); free(r);
    return 0;
}

/* Stand-alone operators, so tests can check each half on its own. */
int cdl_oracle_analysis(const int *geom15, const float *x, const float *w, float *out)
{
    geom_t g; memcpy(&g, geom15, sizeof(g)); analysis(&g, x, w, out); return 0;
}

int cdl_oracle_synthesis(const int *geom15, const float *z, const float *w, float *out)
{
    geom_t g; memcpy(&g, geom15, sizeof(g)); synthesis(&g, z, w, out); return 0;
}

void cdl_oracle_shrink(const float *x, float t, float *out, int n)
{
    for (int i = 0; i < n; ++i) out[i] = shrink(x[i], t);
}
