// Proximal maps of the CSR temporal variants: prox_CSR and prox_CSR_f2
// of the reference's model/net.py:229-262, forward and reverse, as pointwise passes over the fat
// (N, M, D', H', W') code tensors with per-(sample, channel) thresholds.
//
// The maps are discontinuous when a threshold is negative (ST(x, t<0) jumps by 2|t| at x = 0), so the
// forward kernel keeps the reference's evaluation order term by term and forbids fma contraction: a
// re-association that moves an intermediate across 0 by one ulp would change the result by |t|.
#include "cdl_common.h"

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

namespace {

__device__ __forceinline__ float sgn(float x) { return cdl_sgn(x); }

// d ST(x,t) / dx and d ST(x,t) / dt as autograd sees sign(x) * relu(|x| - t)  (sign has zero gradient)
__device__ __forceinline__ float st_dx(float x, float t) { return (x != 0.0f && fabsf(x) - t > 0.0f) ? 1.0f : 0.0f; }
__device__ __forceinline__ float st_dt(float x, float t) { return (fabsf(x) - t > 0.0f) ? -sgn(x) : 0.0f; }

using Prox1 = cdl_prox1;
using Prox2 = cdl_prox2;
__device__ __forceinline__ Prox1 prox1(float u, float zp, float lam, float gam) { return cdl_prox_csr1(u, zp, lam, gam); }
__device__ __forceinline__ Prox2 prox2(float u, float zp, float za, float lam, float g1, float g2)
{
    return cdl_prox_csr2(u, zp, za, lam, g1, g2);
}

__global__ __launch_bounds__(256) void k_prox_fwd(const float *__restrict__ u, const float *__restrict__ zp,
                                                  const float *__restrict__ za, const float *__restrict__ lam,
                                                  const float *__restrict__ g1, const float *__restrict__ g2,
                                                  float *__restrict__ out, size_t total, size_t per_m)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t row = i / per_m;
    out[i] = za ? prox2(u[i], zp[i], za[i], lam[row], g1[row], g2[row]).z : prox1(u[i], zp[i], lam[row], g1[row]).z;
}

// Reverse of the map for one (row, split): gu, the neighbour-code gradients (accumulated: a neighbour
// feeds every iteration) and this split's three threshold sums into part[(row*S + split)*3 + {0,1,2}].
__global__ __launch_bounds__(256) void k_prox_bwd(const float *__restrict__ gz, const float *__restrict__ u,
                                                  const float *__restrict__ zp, const float *__restrict__ za,
                                                  const float *__restrict__ lam, const float *__restrict__ g1,
                                                  const float *__restrict__ g2, float *__restrict__ gu,
                                                  float *__restrict__ gzp, float *__restrict__ gza,
                                                  float *__restrict__ part, size_t per_m, int S)
{
    __shared__ float red[3][4];
    const int row = blockIdx.x / S, sp = blockIdx.x % S;
    const size_t chunk = (per_m + S - 1) / S;
    const size_t lo = (size_t)sp * chunk, hi = lo + chunk < per_m ? lo + chunk : per_m;
    const size_t base = (size_t)row * per_m;
    const float l = lam[row], a1 = g1[row], a2 = g2 ? g2[row] : 0.0f;
    float sl = 0.0f, s1 = 0.0f, s2 = 0.0f;
    for (size_t j = lo + threadIdx.x; j < hi; j += 256) {
        const size_t i = base + j;
        const float g = gz[i], uu = u[i], p = zp[i];
        if (!za) {
            const Prox1 q = prox1(uu, p, l, a1);
            const float gm = g * st_dx(q.m, l);
            const float ga = gm * st_dx(q.a, q.tg);
            const float gtg = gm * st_dt(q.a, q.tg);
            gu[i] = ga;
            if (gzp) gzp[i] += gm - ga;
            sl += g * st_dt(q.m, l) + gtg * a1 + (gm - ga) * q.s;
            s1 += gtg * l;
        } else {
            const float a = za[i];
            const Prox2 q = prox2(uu, p, a, l, a1, a2);
            const float gm = g * st_dx(q.m, l);
            const float gb = gm * st_dx(q.b, q.t2);
            const float gt2 = gm * st_dt(q.b, q.t2);
            const float ga = gb * st_dx(q.a, q.t1);
            const float gt1 = gb * st_dt(q.a, q.t1);
            const float gcb = gm - gb;
            gu[i] = ga;
            if (gzp) gzp[i] -= ga;
            if (gza) gza[i] += gcb;
            const float p1 = (gb - gm) * q.sa + gcb * q.sap + gt1;      // gradient w.r.t. the product lam*g1
            const float p2 = -ga * q.spa + gt2;                          // ... lam*g2
            sl += g * st_dt(q.m, l) - ga * q.sp + gcb * q.sa_ + p1 * a1 + p2 * a2;
            s1 += p1 * l;
            s2 += p2 * l;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sl += __shfl_down(sl, off, 64);
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    if (threadIdx.x % 64 == 0) {
        red[0][threadIdx.x / 64] = sl;
        red[1][threadIdx.x / 64] = s1;
        red[2][threadIdx.x / 64] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        part[(size_t)blockIdx.x * 3 + threadIdx.x] =
            (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// d(t[k,0,m]) = sum_n s[n,m], d(t[k,1,m]) = sum_n c[n] s[n,m] for each of the three threshold families
// (lam = t[k,0] + c t[k,1] etc., net.py:444-452).  One thread per (family, m); fixed order.
__global__ void k_prox_fold(const float *__restrict__ part, const float *__restrict__ c, float *__restrict__ dlam,
                            float *__restrict__ dg1, float *__restrict__ dg2, int N, int M, int S)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * M) return;
    const int fam = i / M, m = i % M;
    float *dst = fam == 0 ? dlam : (fam == 1 ? dg1 : dg2);
    if (!dst) return;
    float a0 = 0.0f, a1 = 0.0f;
    for (int n = 0; n < N; ++n) {
        float v = 0.0f;
        for (int s = 0; s < S; ++s) v += part[((size_t)(n * M + m) * S + s) * 3 + fam];
        a0 += v;
        if (c) a1 = fmaf(c[n], v, a1);
    }
    dst[m] = a0;
    dst[M + m] = a1;
}

int splits_for(int rows, size_t per_m)
{
    int S = (2048 + rows - 1) / rows;                 // ~8 blocks per CU
    const size_t most = (per_m + 1023) / 1024;        // at least 1024 elements per block
    if ((size_t)S > most) S = (int)most;
    return S < 1 ? 1 : S;
}

size_t code_elems(const cdl_geom *g) { return (size_t)(g->D / g->sd) * (g->H / g->sh) * (g->W / g->sw); }

}  // namespace

extern "C" {

int cdl_prox_csr(const cdl_geom *g, const float *u, const float *z_prev, const float *z_after, const float *lam,
                 const float *gam1, const float *gam2, float *out, void *stream)
{
    if (!cdl_geom_ok(g) || !u || !z_prev || !lam || !gam1 || !out) return CDL_EINVAL;
    if (z_after && !gam2) return CDL_EINVAL;
    const size_t per_m = code_elems(g), total = (size_t)g->N * g->M * per_m;
    k_prox_fwd<<<(unsigned)((total + 255) / 256), 256, 0, S(stream)>>>(u, z_prev, z_after, lam, gam1, gam2, out,
                                                                         total, per_m);
    CDL_LAUNCH_CHECK();
    return 0;
}

size_t cdl_prox_csr_scratch_floats(const cdl_geom *g)
{
    if (!cdl_geom_ok(g)) return 0;
    return (size_t)g->N * g->M * splits_for(g->N * g->M, code_elems(g)) * 3;
}

int cdl_prox_csr_bwd(const cdl_geom *g, const float *gz, const float *u, const float *z_prev,
                     const float *z_after, const float *lam, const float *gam1, const float *gam2,
                     const float *c, float *gu, float *gz_prev, float *gz_after, float *dlam, float *dgam1,
                     float *dgam2, float *scratch, size_t scratch_floats, void *stream)
{
    if (!cdl_geom_ok(g) || !gz || !u || !z_prev || !lam || !gam1 || !gu || !dlam || !dgam1 || !scratch)
        return CDL_EINVAL;
    if (z_after && (!gam2 || !dgam2)) return CDL_EINVAL;
    if (!z_after && gz_after) return CDL_EINVAL;
    if (scratch_floats < cdl_prox_csr_scratch_floats(g)) return CDL_EINVAL;
    const size_t per_m = code_elems(g);
    const int rows = g->N * g->M, Sp = splits_for(rows, per_m);
    k_prox_bwd<<<(unsigned)(rows * Sp), 256, 0, S(stream)>>>(gz, u, z_prev, z_after, lam, gam1, gam2, gu, gz_prev,
                                                             gz_after, scratch, per_m, Sp);
    CDL_LAUNCH_CHECK();
    k_prox_fold<<<(3 * g->M + 63) / 64, 64, 0, S(stream)>>>(scratch, c, dlam, dgam1, z_after ? dgam2 : nullptr,
                                                             g->N, g->M, Sp);
    CDL_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
