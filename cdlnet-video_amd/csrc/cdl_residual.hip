// ResidualBlock of CDLNetVideo(residual=True) (reference model/net.py:105-120,
// applied to the code after every iteration, net.py:199-207):
//     h   = relu(conv1(x))                 conv1, conv2: Conv3d(M, M, 3x3x3, stride 1, padding 1, bias=False)
//     out = relu(conv2(h) + x)
// A Conv3d(M, M) is the analysis operator of a geometry with C = M "image" channels, its data gradient the
// synthesis operator and its filter gradient cdl_wgrad, so the block is five launches of those operators (which
// pick the dense matrix-core tier for this geometry, cdl_dense_mfma.hip) plus the element-wise passes here:
//     forward    h = A(x; w1), relu;  out = x + A(h; w2), relu
//     backward   g2 = g_out [out > 0]    dh = S(g2; w2)    g1 = dh [h > 0]
//                dw2 = wgrad(g2, h)      dw1 = wgrad(g1, x)    dx = S(g1; w1) + g2
// The relu gates are the `gate` arguments of the operators (pass where gate != 0; relu outputs are >= 0).
#include "cdl_common.h"

static inline hipStream_t S(void *s) { return (hipStream_t)s; }
static inline size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

#define CDL_TRY(expr)              \
    do {                           \
        const int rc_ = (expr);    \
        if (rc_ != 0) return rc_;  \
    } while (0)

namespace {

__global__ __launch_bounds__(256) void k_relu(float *__restrict__ v, size_t n4, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        float4 a = reinterpret_cast<float4 *>(v)[i];
        a.x = fmaxf(a.x, 0.0f);
        a.y = fmaxf(a.y, 0.0f);
        a.z = fmaxf(a.z, 0.0f);
        a.w = fmaxf(a.w, 0.0f);
        reinterpret_cast<float4 *>(v)[i] = a;
    }
    if (i < n - 4 * n4) v[4 * n4 + i] = fmaxf(v[4 * n4 + i], 0.0f);
}

// dst = g where gate != 0, else 0
__global__ __launch_bounds__(256) void k_gate(float *__restrict__ dst, const float *__restrict__ gsrc,
                                              const float *__restrict__ gate, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = gate[i] != 0.0f ? gsrc[i] : 0.0f;
}

// acc += g where gate != 0
__global__ __launch_bounds__(256) void k_add_gated(float *__restrict__ acc, const float *__restrict__ gsrc,
                                                   const float *__restrict__ gate, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && gate[i] != 0.0f) acc[i] += gsrc[i];
}

bool block_geom_ok(const cdl_geom *g)
{
    return cdl_geom_ok(g) && g->C == g->M && g->sd == 1 && g->sh == 1 && g->sw == 1;
}

int relu_inplace(float *v, size_t n, hipStream_t st)
{
    const size_t n4 = ((reinterpret_cast<size_t>(v) & 15) == 0) ? n / 4 : 0;
    const size_t threads = n4 > n - 4 * n4 ? n4 : n - 4 * n4;
    k_relu<<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(v, n4, n);
    CDL_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

size_t cdl_residual_scratch_floats(const cdl_geom *g)
{
    if (!block_geom_ok(g)) return 0;
    size_t n = cdl_wgrad_workspace_floats(g);
    n = max_sz(n, cdl_synthesis_workspace_floats(g));
    n = max_sz(n, cdl_analysis_workspace_floats(g));
    return n;
}

int cdl_residual_forward(const cdl_geom *g, const float *x, const float *w1, const float *w2, float *h, float *out,
                         float *scratch, size_t scratch_floats, void *stream)
{
    if (!block_geom_ok(g) || !x || !w1 || !w2 || !h || !out || h == x || out == x || out == h) return CDL_EINVAL;
    if (scratch_floats < cdl_residual_scratch_floats(g) || (scratch_floats && !scratch)) return CDL_EINVAL;
    const size_t n = (size_t)g->N * g->M * g->D * g->H * g->W;
    // dense matrix-core tier: the relu is the convolution's epilogue (2 launches + 2 fragment preps per block)
    if (cdl_opts().mfma_dense && !cdl_opts().no_tiled) {
        const int rc = cdl_dense_conv(g, 0, x, nullptr, w1, 1.0f, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, h,
                                      scratch, scratch_floats, stream);
        if (rc == 0)
            return cdl_dense_conv(g, 0, h, nullptr, w2, 1.0f, x, nullptr, nullptr, nullptr, nullptr, 1, nullptr, out, scratch,
                                  scratch_floats, stream);
        if (rc != CDL_EUNSUPPORTED) return rc;
    }
    CDL_TRY(cdl_analysis_ws(g, x, w1, 1.0f, nullptr, nullptr, nullptr, h, scratch, scratch_floats, stream));
    CDL_TRY(relu_inplace(h, n, S(stream)));
    CDL_TRY(cdl_analysis_ws(g, h, w2, 1.0f, x, nullptr, nullptr, out, scratch, scratch_floats, stream));
    return relu_inplace(out, n, S(stream));
}

int cdl_residual_backward(const cdl_geom *g, const float *x, const float *h, const float *out, const float *w1,
                          const float *w2, const float *g_out, float *dx, float *dw1, float *dw2, float *dh,
                          float *scratch, size_t scratch_floats, void *stream)
{
    if (!block_geom_ok(g) || !x || !h || !out || !w1 || !w2 || !g_out || !dx || !dw1 || !dw2 || !dh)
        return CDL_EINVAL;
    if (scratch_floats < cdl_residual_scratch_floats(g) || (scratch_floats && !scratch)) return CDL_EINVAL;
    const size_t n = (size_t)g->N * g->M * g->D * g->H * g->W;
    {   // dense tier: each relu gate is applied ONCE, where the gated gradient is produced (g2 by one element-wise
        // pass into dx, which is free until the last launch; g1 by the epilogue of the launch that computes dh), so
        // the two data-gradient and two filter-gradient launches read no gate (each re-reads its input ~3x)
        if (cdl_opts().mfma_dense && !cdl_opts().no_tiled && cdl_dense_ws_floats(g, 1) && cdl_dense_wgrad_ws_floats(g)) {
            float *g2 = dx;
            k_gate<<<(unsigned)((n + 255) / 256), 256, 0, S(stream)>>>(g2, g_out, out, n);
            CDL_LAUNCH_CHECK();
            CDL_TRY(cdl_dense_conv(g, 1, g2, nullptr, w2, 1.0f, nullptr, nullptr, nullptr, nullptr, nullptr, 0, h, dh,
                                   scratch, scratch_floats, stream));                     // dh = S(g2; w2) [h > 0] = g1
            CDL_TRY(cdl_wgrad(g, g2, nullptr, h, 1.0f, dw2, scratch, scratch_floats, stream));
            CDL_TRY(cdl_wgrad(g, dh, nullptr, x, 1.0f, dw1, scratch, scratch_floats, stream));
            // dx = S(g1; w1) + g2, g2 read from dx itself: every element is read and then written by the same thread
            return cdl_dense_conv(g, 1, dh, nullptr, w1, 1.0f, g2, nullptr, nullptr, nullptr, nullptr, 0, nullptr, dx,
                                  scratch, scratch_floats, stream);
        }
    }
    CDL_TRY(cdl_synthesis_ws(g, g_out, out, w2, 1.0f, nullptr, nullptr, dh, scratch, scratch_floats, stream));
    CDL_TRY(cdl_wgrad(g, g_out, out, h, 1.0f, dw2, scratch, scratch_floats, stream));
    CDL_TRY(cdl_wgrad(g, dh, h, x, 1.0f, dw1, scratch, scratch_floats, stream));
    CDL_TRY(cdl_synthesis_ws(g, dh, h, w1, 1.0f, nullptr, nullptr, dx, scratch, scratch_floats, stream));
    k_add_gated<<<(unsigned)((n + 255) / 256), 256, 0, S(stream)>>>(dx, g_out, out, n);
    CDL_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
