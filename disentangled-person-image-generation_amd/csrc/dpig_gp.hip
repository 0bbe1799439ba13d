// dpig_gp_double_backward: the WGAN-GP gradient-penalty term of the DCGAN critic and its gradient w.r.t. every critic
// parameter in ONE C-ABI call (trainer.py:222-236, wgan_gp.py:407-440, 605-619).
//
// TensorFlow gets d(penalty)/d(theta) by differentiating its own backward graph (tf.gradients of tf.gradients).  The critic is a
// fixed chain -- Conv5x5s2 -> LReLU -> [Conv5x5s2 -> LayerNorm -> LReLU] x3 -> reshape(NCHW order) -> Linear -- so the second
// derivative is written out analytically here as three sweeps of the library's own kernels, no autograd tape:
//   sweep 1 (forward on xhat, then the input gradient g = d sum D(xhat) / d xhat),
//   penalty + seed u0 = d penalty / d g                                   (dpig_gp_penalty),
//   sweep 2 "up"  : the adjoint of sweep 1's backward half, from g back to the seed: every dgrad's adjoint is a forward
//                   conv (w.r.t. its dy) and a wgrad (w.r.t. the filter); LayerNorm's backward has the second-order kernel
//                   dpig_ln_bwd2 (adjoints w.r.t. dy, x and scale); LReLU masks are piecewise constant,
//   sweep 3 "down": the x-adjoints dpig_ln_bwd2 produced flow down the FORWARD graph as an ordinary backward pass.
// All tensors NHWC fp32; convolution arithmetic follows desc->compute (fp32 MFMA by default).
#include "dpig_common.h"

namespace dpig {

__global__ __launch_bounds__(256) void gp_seed_kernel(const float* __restrict__ w_out, float* __restrict__ out, long total, int F) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) out[i] = w_out[i % F];
}
// out = beta * out + a (+ b)
__global__ __launch_bounds__(256) void gp_acc_kernel(float* __restrict__ out, float beta, const float* __restrict__ a,
                                                     const float* __restrict__ b, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = (beta != 0.f) ? beta * out[i] : 0.f;
        if (a) v += a[i];
        if (b) v += b[i];
        out[i] = v;
    }
}
static inline int gp_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

struct GpGeom {
    int B, H[5], W[5], C[5];
    long n[5];          // elements of level l (0 = the image)
    int F, R;           // linear fan-in (8*4*8*dim, the hard-coded reshape of wgan_gp.py:433) and logit rows
    DpigConvDesc cd[5]; // cd[l]: conv l (level l-1 -> l)
};

static int gp_geom(const DpigCriticDesc* d, GpGeom* g) {
    if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->dim <= 0) return fail(DPIG_EINVAL, "gp: bad critic descriptor");
    g->B = d->B;
    g->H[0] = d->H; g->W[0] = d->W; g->C[0] = d->Cin;
    for (int l = 1; l <= 4; ++l) {
        g->H[l] = (g->H[l - 1] + 1) / 2;
        g->W[l] = (g->W[l - 1] + 1) / 2;
        g->C[l] = d->dim << (l - 1);
    }
    for (int l = 0; l <= 4; ++l) g->n[l] = (long)g->B * g->H[l] * g->W[l] * g->C[l];
    g->F = 8 * 4 * 8 * d->dim;
    if (g->n[4] % g->F) return fail(DPIG_EINVAL, "gp: %ld critic features do not reshape to rows of %d (wgan_gp.py:433)", g->n[4], g->F);
    g->R = (int)(g->n[4] / g->F);
    for (int l = 1; l <= 4; ++l) {
        DpigConvDesc& c = g->cd[l];
        c = DpigConvDesc{};
        c.N = g->B; c.H = g->H[l - 1]; c.W = g->W[l - 1]; c.C = g->C[l - 1]; c.K = g->C[l];
        c.R = c.S = 5; c.stride = 2; c.pad_t = c.pad_l = -1;
        c.ldx = c.C; c.ldy = c.K; c.ldres = c.K; c.ldmask = c.C; c.ldy2 = c.K;
        c.act = DPIG_ACT_NONE; c.alpha = d->lrelu_alpha; c.compute = d->compute;
    }
    return DPIG_OK;
}

struct GpPlan {
    size_t conv_ws, ln_ws, ln2_ws, cs_ws, total;
    size_t off_scratch;
};
static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// per-level tensor slots (each n[l] floats): z, a, da (dL/d a_l of sweep 1), dz, v, ub, zb, t
enum { S_Z = 0, S_A, S_DA, S_DZ, S_V, S_UB, S_ZB, S_T, S_COUNT };

static size_t gp_layout(const GpGeom& g, GpPlan* p) {
    size_t cw = 0;
    for (int l = 1; l <= 4; ++l)
        for (int which = 0; which < 3; ++which) {
            const size_t w = dpig_conv2d_workspace_bytes(&g.cd[l], which);
            cw = w > cw ? w : cw;
        }
    size_t lw = 0, l2 = 0, cs = dpig_colsum_workspace_bytes(g.R, g.F);
    for (int l = 2; l <= 4; ++l) {
        const int P = g.H[l] * g.W[l];
        const size_t a = dpig_ln_workspace_bytes(g.B, P, g.C[l]), b = dpig_ln_bwd2_workspace_bytes(g.B, P, g.C[l]);
        lw = a > lw ? a : lw;
        l2 = b > l2 ? b : l2;
    }
    p->conv_ws = up256(cw); p->ln_ws = up256(lw); p->ln2_ws = up256(l2); p->cs_ws = up256(cs);
    size_t tot = 0;
    tot += up256(g.n[0] * 4) * 3;                                   // xhat, g, u0
    for (int l = 1; l <= 4; ++l) tot += up256(g.n[l] * 4) * S_COUNT;
    tot += up256((size_t)g.B * 4) * 2 * 3;                          // LN mean / rstd, levels 2..4
    tot += up256((size_t)g.C[4] * 4) * 3;                           // per-channel temporaries
    p->off_scratch = tot;
    tot += p->conv_ws + p->ln_ws + p->ln2_ws + p->cs_ws;
    p->total = tot;
    return tot;
}

#define GP_TRY(expr)             \
    do {                         \
        int rc__ = (expr);       \
        if (rc__) return rc__;   \
    } while (0)

}  // namespace dpig

using namespace dpig;

extern "C" size_t dpig_gp_double_backward_workspace_bytes(const DpigCriticDesc* d) {
    GpGeom g;
    if (gp_geom(d, &g)) return 0;
    GpPlan p;
    return gp_layout(g, &p);
}

extern "C" int dpig_gp_double_backward(const DpigCriticDesc* d, const DpigCriticParams* P, const float* real, const float* fake,
                                       const float* alpha, float beta, const DpigCriticGrads* G, float* penalty, float* slopes,
                                       void* ws, size_t ws_bytes, void* stream) {
    GpGeom g;
    GP_TRY(gp_geom(d, &g));
    if (!P || !real || !fake || !alpha || !penalty || !slopes) return fail(DPIG_EINVAL, "gp_double_backward: null pointer");
    for (int l = 0; l < 4; ++l)
        if (!P->w[l] || !P->b[l] || (G && (!G->w[l] || !G->b[l]))) return fail(DPIG_EINVAL, "gp_double_backward: null conv parameter");
    for (int l = 0; l < 3; ++l)
        if (!P->ln_scale[l] || !P->ln_offset[l] || (G && (!G->ln_scale[l] || !G->ln_offset[l])))
            return fail(DPIG_EINVAL, "gp_double_backward: null LayerNorm parameter");
    if (!P->w_out || (G && !G->w_out)) return fail(DPIG_EINVAL, "gp_double_backward: null linear parameter");
    GpPlan pl;
    gp_layout(g, &pl);
    if (!ws || ws_bytes < pl.total) return fail(DPIG_ENOMEM, "gp_double_backward: workspace too small (%zu < %zu)", ws_bytes, pl.total);
    if (!aligned16(ws)) return fail(DPIG_EINVAL, "gp_double_backward: workspace must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);

    // ---- carve the workspace ------------------------------------------------------------------------------------------
    char* cur = static_cast<char*>(ws);
    auto take = [&](size_t bytes) { float* r = reinterpret_cast<float*>(cur); cur += up256(bytes); return r; };
    float* xhat = take(g.n[0] * 4);
    float* gin = take(g.n[0] * 4);
    float* u0 = take(g.n[0] * 4);
    float* T[5][S_COUNT];
    for (int l = 1; l <= 4; ++l)
        for (int s = 0; s < S_COUNT; ++s) T[l][s] = take(g.n[l] * 4);
    float *mean[5], *rstd[5];
    for (int l = 2; l <= 4; ++l) { mean[l] = take((size_t)g.B * 4); rstd[l] = take((size_t)g.B * 4); }
    float* c0 = take((size_t)g.C[4] * 4);
    float* c1 = take((size_t)g.C[4] * 4);
    float* c2 = take((size_t)g.C[4] * 4);
    void* cws = cur; cur += pl.conv_ws;
    void* lws = cur; cur += pl.ln_ws;
    void* l2ws = cur; cur += pl.ln2_ws;
    void* csws = cur; cur += pl.cs_ws;
    const float la = d->lrelu_alpha, eps = d->ln_eps;
    auto PX = [&](int l) { return g.H[l] * g.W[l]; };

    // ---- sweep 1: forward on xhat ---------------------------------------------------------------------------------------
    GP_TRY(dpig_gp_interpolate(real, fake, alpha, g.B, g.n[0] / g.B, xhat, stream));
    {
        DpigConvDesc c = g.cd[1];
        c.act = DPIG_ACT_LRELU;                                         // wgan_gp.py:414-415: conv -> LeakyReLU, one epilogue
        GP_TRY(dpig_conv2d_fwd(&c, xhat, P->w[0], P->b[0], nullptr, T[1][S_A], nullptr, cws, pl.conv_ws, stream));
    }
    for (int l = 2; l <= 4; ++l) {
        GP_TRY(dpig_conv2d_fwd(&g.cd[l], T[l - 1][S_A], P->w[l - 1], P->b[l - 1], nullptr, T[l][S_Z], nullptr, cws, pl.conv_ws, stream));
        GP_TRY(dpig_ln_fwd(T[l][S_Z], g.B, PX(l), g.C[l], P->ln_scale[l - 2], P->ln_offset[l - 2], eps, DPIG_ACT_LRELU, la,
                           T[l][S_A], mean[l], rstd[l], stream));
    }
    // ---- sweep 1: input gradient of sum(D(xhat)).  d out / d feature = w_out for every logit row (linear.py:132-146);
    //      the reshape is over the logical NCHW tensor (wgan_gp.py:433), the data is NHWC -> one transpose.
    hipLaunchKernelGGL(gp_seed_kernel, dim3(gp_blocks(g.n[4])), dim3(256), 0, st, P->w_out, T[4][S_T], g.n[4], g.F);
    GP_TRY(check_launch("gp_seed"));
    GP_TRY(dpig_transpose12(T[4][S_T], T[4][S_DA], g.B, g.C[4], PX(4), 4, stream));
    for (int l = 4; l >= 2; --l) {
        GP_TRY(dpig_ln_bwd(T[l][S_DA], T[l][S_Z], T[l][S_A], g.B, PX(l), g.C[l], P->ln_scale[l - 2], mean[l], rstd[l], DPIG_ACT_LRELU, la,
                           T[l][S_DZ], c0, c1, lws, pl.ln_ws, stream));
        DpigConvDesc c = g.cd[l];
        if (l == 2) { c.act = DPIG_ACT_LRELU; }                         // level 1 has no norm: LReLU' rides the dgrad epilogue
        GP_TRY(dpig_conv2d_dgrad(&c, T[l][S_DZ], P->w[l - 1], nullptr, l == 2 ? T[1][S_A] : nullptr,
                                 l == 2 ? T[1][S_DZ] : T[l - 1][S_DA], cws, pl.conv_ws, stream));
    }
    GP_TRY(dpig_conv2d_dgrad(&g.cd[1], T[1][S_DZ], P->w[0], nullptr, nullptr, gin, cws, pl.conv_ws, stream));
    // ---- penalty and the seed of the second sweep ---------------------------------------------------------------------------
    GP_TRY(dpig_gp_penalty(gin, g.B, g.n[0] / g.B, d->lambda, penalty, u0, slopes, stream));
    if (!G) return DPIG_OK;

    // ---- sweep 2 "up": adjoint of the backward half -----------------------------------------------------------------------------
    // g = dgrad(w1, dz1): adjoint w.r.t. dz1 = conv(u0, w1); w.r.t. w1 = wgrad(x = u0, dy = dz1)
    GP_TRY(dpig_conv2d_fwd(&g.cd[1], u0, P->w[0], nullptr, nullptr, T[1][S_V], nullptr, cws, pl.conv_ws, stream));
    GP_TRY(dpig_conv2d_wgrad(&g.cd[1], u0, T[1][S_DZ], G->w[0], beta, nullptr, 0.f, cws, pl.conv_ws, stream));
    GP_TRY(dpig_act_bwd(T[1][S_V], g.C[1], T[1][S_A], g.C[1], T[1][S_UB], g.C[1], g.n[1] / g.C[1], g.C[1], DPIG_ACT_LRELU, la, stream));
    for (int l = 2; l <= 4; ++l) {
        GP_TRY(dpig_conv2d_fwd(&g.cd[l], T[l - 1][S_UB], P->w[l - 1], nullptr, nullptr, T[l][S_V], nullptr, cws, pl.conv_ws, stream));
        GP_TRY(dpig_conv2d_wgrad(&g.cd[l], T[l - 1][S_UB], T[l][S_DZ], G->w[l - 1], beta, nullptr, 0.f, cws, pl.conv_ws, stream));
        // dz_l = LNbwd(dy = da_l; x = z_l, scale): adjoints w.r.t. dy (-> ub_l), x (-> zb_l) and scale
        float* dsc = (l == 4) ? c0 : (l == 3 ? c1 : c2);                // kept until the down-sweep adds its share
        GP_TRY(dpig_ln_bwd2(T[l][S_V], T[l][S_DA], T[l][S_Z], T[l][S_A], g.B, PX(l), g.C[l], P->ln_scale[l - 2], mean[l], rstd[l],
                            DPIG_ACT_LRELU, la, T[l][S_UB], T[l][S_ZB], dsc, l2ws, pl.ln2_ws, stream));
    }
    // da_4 = reshape(w_out): adjoint w.r.t. w_out[j] = sum over logit rows of ub_4 in NCHW order
    GP_TRY(dpig_transpose12(T[4][S_UB], T[4][S_T], g.B, PX(4), g.C[4], 4, stream));
    GP_TRY(dpig_colsum(T[4][S_T], g.F, g.R, g.F, G->w_out, beta, csws, pl.cs_ws, stream));

    // ---- sweep 3 "down": zb_l flows down the forward graph ------------------------------------------------------------------------
    // level 4: LayerNorm 4's scale gets only the second-order share; its offset none.
    hipLaunchKernelGGL(gp_acc_kernel, dim3(gp_blocks(g.C[4])), dim3(256), 0, st, G->ln_scale[2], beta, (const float*)c0, (const float*)nullptr, (long)g.C[4]);
    hipLaunchKernelGGL(gp_acc_kernel, dim3(gp_blocks(g.C[4])), dim3(256), 0, st, G->ln_offset[2], beta, (const float*)nullptr, (const float*)nullptr, (long)g.C[4]);
    GP_TRY(check_launch("gp_acc"));
    for (int l = 4; l >= 2; --l) {
        // z_l = conv(a_{l-1}, w_l) + b_l
        GP_TRY(dpig_conv2d_wgrad(&g.cd[l], T[l - 1][S_A], T[l][S_ZB], G->w[l - 1], 1.0f, G->b[l - 1], beta, cws, pl.conv_ws, stream));
        if (l > 2) {
            GP_TRY(dpig_conv2d_dgrad(&g.cd[l], T[l][S_ZB], P->w[l - 1], nullptr, nullptr, T[l - 1][S_T], cws, pl.conv_ws, stream));
            // a_{l-1} = LReLU(LN(z_{l-1})): first-order LayerNorm backward; dx joins zb_{l-1}
            float* dsc2 = (l - 1 == 3) ? c1 : c2;
            // first-order dscale / doffset of this level into V (dead after the up-sweep; used as per-channel scratch)
            float* ds = T[l - 1][S_V];
            float* dof = T[l - 1][S_V] + g.C[l - 1];
            GP_TRY(dpig_ln_bwd(T[l - 1][S_T], T[l - 1][S_Z], T[l - 1][S_A], g.B, PX(l - 1), g.C[l - 1], P->ln_scale[l - 3], mean[l - 1],
                               rstd[l - 1], DPIG_ACT_LRELU, la, T[l - 1][S_DA], ds, dof, lws, pl.ln_ws, stream));
            hipLaunchKernelGGL(gp_acc_kernel, dim3(gp_blocks(g.C[l - 1])), dim3(256), 0, st, G->ln_scale[l - 3], beta, (const float*)dsc2,
                               (const float*)ds, (long)g.C[l - 1]);
            hipLaunchKernelGGL(gp_acc_kernel, dim3(gp_blocks(g.C[l - 1])), dim3(256), 0, st, G->ln_offset[l - 3], beta, (const float*)dof,
                               (const float*)nullptr, (long)g.C[l - 1]);
            hipLaunchKernelGGL(gp_acc_kernel, dim3(gp_blocks(g.n[l - 1])), dim3(256), 0, st, T[l - 1][S_ZB], 1.0f,
                               (const float*)T[l - 1][S_DA], (const float*)nullptr, g.n[l - 1]);
            GP_TRY(check_launch("gp_acc"));
        } else {
            // a_1 = LReLU(z_1): the mask rides the dgrad epilogue; then z_1 = conv(xhat, w_1) + b_1
            DpigConvDesc c = g.cd[2];
            c.act = DPIG_ACT_LRELU;
            GP_TRY(dpig_conv2d_dgrad(&c, T[2][S_ZB], P->w[1], nullptr, T[1][S_A], T[1][S_ZB], cws, pl.conv_ws, stream));
            GP_TRY(dpig_conv2d_wgrad(&g.cd[1], xhat, T[1][S_ZB], G->w[0], 1.0f, G->b[0], beta, cws, pl.conv_ws, stream));
        }
    }
    return DPIG_OK;
}
