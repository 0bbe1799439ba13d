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
// Images, parameters and parameter gradients are fp32.  desc->compute selects the arithmetic AND the storage of the critic's own
// activations inside the workspace: DPIG_COMPUTE_F32 / _BF16 / _BF16X3 keep them fp32 (the conv entry points' modes);
// DPIG_COMPUTE_BF16_STORE (the 'bf16' storage mode of BASELINE configs[2]-[4]) stores every level tensor as bf16 and runs the
// bf16-storage kernels -- conv levels 2-4 on dpig_conv2d_{fwd,dgrad,wgrad}_bf16 with filter shadows derived inside the call,
// level 1 (3 input channels) on the vector-ALU kernels with a bf16 wide side, LayerNorm on dpig_ln_*_bf16 -- with fp32
// accumulation, statistics and parameter gradients throughout.  Same sweeps, same order.
#include "dpig_common.h"

namespace dpig {

typedef unsigned short gp_bf16;
__device__ __forceinline__ void gp_store(float* p, long i, float v) { p[i] = v; }
__device__ __forceinline__ void gp_store(gp_bf16* p, long i, float v) {
    const __bf16 b = (__bf16)v;
    p[i] = __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float gp_load(const float* p, long i) { return p[i]; }
__device__ __forceinline__ float gp_load(const gp_bf16* p, long i) { return __uint_as_float((unsigned)p[i] << 16); }

template <typename T>
__global__ __launch_bounds__(256) void gp_seed_kernel(const float* __restrict__ w_out, T* __restrict__ out, long total, int F) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) gp_store(out, i, w_out[i % F]);
}
// out = beta * out + a (+ b)
template <typename T>
__global__ __launch_bounds__(256) void gp_acc_kernel(T* __restrict__ out, float beta, const T* __restrict__ a, const T* __restrict__ b,
                                                     long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = (beta != 0.f) ? beta * gp_load(out, i) : 0.f;
        if (a) v += gp_load(a, i);
        if (b) v += gp_load(b, i);
        gp_store(out, i, v);
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
    bool bf16;          // DPIG_COMPUTE_BF16_STORE
};

static int gp_geom(const DpigCriticDesc* d, GpGeom* g) {
    if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->dim <= 0) return fail(DPIG_EINVAL, "gp: bad critic descriptor");
    if (d->compute < DPIG_COMPUTE_F32 || d->compute > DPIG_COMPUTE_BF16_STORE) return fail(DPIG_EINVAL, "gp: unknown compute mode %d", d->compute);
    g->bf16 = d->compute == DPIG_COMPUTE_BF16_STORE;
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
        c.act = DPIG_ACT_NONE; c.alpha = d->lrelu_alpha; c.compute = g->bf16 ? DPIG_COMPUTE_F32 : d->compute;
    }
    if (g->bf16) {
        // what the bf16-storage kernels take: a 3-channel image into a power-of-two number of channel quads (the vector-ALU first
        // layer), 16-byte channel vectors above it
        const int lp = d->dim / 4;
        if (d->Cin != 3 || d->dim < 32 || lp > 64 || (lp & (lp - 1)))
            return fail(DPIG_EINVAL, "gp: bf16 storage needs Cin = 3 and dim in {32, 64, 128, 256}");
        for (int l = 2; l <= 4; ++l)
            if (!dpig_conv2d_bf16_supported(&g->cd[l], 0)) return fail(DPIG_EINVAL, "gp: conv level %d is not a bf16-storage layer", l);
    }
    return DPIG_OK;
}

struct GpPlan {
    size_t conv_ws, ln_ws, ln2_ws, cs_ws, pen_ws, total;
    size_t off_scratch;
    size_t shadow_elems[5];   // bf16 storage: elements of filter l's plain (= transposed) shadow, l = 2..4
};
static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline size_t maxz(size_t a, size_t b) { return a > b ? a : b; }

// per-level tensor slots (each n[l] elements): z, a, da (dL/d a_l of sweep 1), dz, v, ub, zb (second-order x-adjoint), t (what arrives
// from the level above in the down-sweep), f (its first-order LayerNorm gradient), zs (zb + f: what flows further down).  No slot is
// written twice, so after the call the workspace holds every link of the three sweeps (dpig_gp_double_backward_slot; tests check the
// chain link by link).  Exceptions: level 4 has zs == zb (nothing arrives from above: zs aliases zb), level 1 has no z / da / f.
enum { S_Z = 0, S_A, S_DA, S_DZ, S_V, S_UB, S_ZB, S_T, S_F, S_ZS, S_COUNT };

static size_t gp_layout(const GpGeom& g, GpPlan* p) {
    const size_t es = g.bf16 ? 2 : 4;
    size_t cw = 0;
    for (int which = 0; which < 3; ++which) cw = maxz(cw, dpig_conv2d_workspace_bytes(&g.cd[1], which));   // level 1: fp32 / thin kernels
    for (int l = 2; l <= 4; ++l)
        for (int which = 0; which < 3; ++which)
            cw = maxz(cw, g.bf16 ? dpig_conv2d_bf16_workspace_bytes(&g.cd[l], which) : dpig_conv2d_workspace_bytes(&g.cd[l], which));
    size_t lw = 0, l2 = 0, cs = dpig_colsum_workspace_bytes(g.R, g.F);
    for (int l = 2; l <= 4; ++l) {
        const int P = g.H[l] * g.W[l];
        lw = maxz(lw, maxz(dpig_ln_workspace_bytes(g.B, P, g.C[l]), dpig_ln_fwd_workspace_bytes(g.B, P, g.C[l])));
        l2 = maxz(l2, dpig_ln_bwd2_workspace_bytes(g.B, P, g.C[l]));
    }
    p->conv_ws = up256(cw); p->ln_ws = up256(lw); p->ln2_ws = up256(l2); p->cs_ws = up256(cs);
    p->pen_ws = up256(dpig_gp_penalty_workspace_bytes(g.B, g.n[0] / g.B));
    size_t tot = 0;
    tot += up256(g.n[0] * 4) * 3;                                   // xhat, g, u0 (fp32 images)
    for (int l = 1; l <= 4; ++l) tot += up256(g.n[l] * es) * S_COUNT;
    tot += up256((size_t)g.B * 4) * 2 * 3;                          // LN mean / rstd, levels 2..4
    tot += up256((size_t)g.C[4] * 4) * 3;                           // per-channel temporaries
    tot += up256((size_t)g.C[4] * 4) * 2;                           // first-order dscale / doffset of the down-sweep
    for (int l = 2; l <= 4; ++l) {
        p->shadow_elems[l] = g.bf16 ? (size_t)25 * g.C[l - 1] * g.C[l] : 0;
        tot += 2 * up256(p->shadow_elems[l] * 2);
    }
    if (g.bf16) tot += up256(g.n[4] * 4);                           // fp32 copy of ub_4 for the output weight's column sum
    p->off_scratch = tot;
    tot += p->conv_ws + p->ln_ws + p->ln2_ws + p->cs_ws + p->pen_ws;
    p->total = tot;
    return tot;
}

#define GP_TRY(expr)             \
    do {                         \
        int rc__ = (expr);       \
        if (rc__) return rc__;   \
    } while (0)

// ---- the kernels of one storage type behind one set of names --------------------------------------------------------------------
template <typename T> struct GpOps;
template <> struct GpOps<float> {
    static int ln_fwd(const float* x, int N, int P, int C, const float* sc, const float* of, float eps, int act, float a, float* y, float* m,
                      float* r, void* ws, size_t wsn, void* st) { return dpig_ln_fwd(x, N, P, C, sc, of, eps, act, a, y, m, r, ws, wsn, st); }
    static int ln_bwd(const float* dy, const float* x, const float* y, int N, int P, int C, const float* sc, const float* m, const float* r,
                      int act, float a, float* dx, float* ds, float* dof, void* ws, size_t wsn, void* st) {
        return dpig_ln_bwd(dy, x, y, N, P, C, sc, m, r, act, a, dx, ds, dof, ws, wsn, st);
    }
    static int ln_bwd2(const float* u, const float* dy, const float* x, const float* y, int N, int P, int C, const float* sc, const float* m,
                       const float* r, int act, float a, float* ddy, float* dx, float* ds, void* ws, size_t wsn, void* st) {
        return dpig_ln_bwd2(u, dy, x, y, N, P, C, sc, m, r, act, a, ddy, dx, ds, ws, wsn, st);
    }
    static int act_bwd(const float* dy, const float* y, float* dz, long rows, int C, int act, float a, void* st) {
        return dpig_act_bwd(dy, C, y, C, dz, C, rows, C, act, a, st);
    }
};
template <> struct GpOps<gp_bf16> {
    static int ln_fwd(const gp_bf16* x, int N, int P, int C, const float* sc, const float* of, float eps, int act, float a, gp_bf16* y, float* m,
                      float* r, void* ws, size_t wsn, void* st) { return dpig_ln_fwd_bf16(x, N, P, C, sc, of, eps, act, a, y, m, r, ws, wsn, st); }
    static int ln_bwd(const gp_bf16* dy, const gp_bf16* x, const gp_bf16* y, int N, int P, int C, const float* sc, const float* m, const float* r,
                      int act, float a, gp_bf16* dx, float* ds, float* dof, void* ws, size_t wsn, void* st) {
        return dpig_ln_bwd_bf16(dy, x, y, N, P, C, sc, m, r, act, a, dx, ds, dof, ws, wsn, st);
    }
    static int ln_bwd2(const gp_bf16* u, const gp_bf16* dy, const gp_bf16* x, const gp_bf16* y, int N, int P, int C, const float* sc,
                       const float* m, const float* r, int act, float a, gp_bf16* ddy, gp_bf16* dx, float* ds, void* ws, size_t wsn, void* st) {
        return dpig_ln_bwd2_bf16(u, dy, x, y, N, P, C, sc, m, r, act, a, ddy, dx, ds, ws, wsn, st);
    }
    static int act_bwd(const gp_bf16* dy, const gp_bf16* y, gp_bf16* dz, long rows, int C, int act, float a, void* st) {
        return dpig_act_bwd_bf16(dy, C, y, C, dz, C, rows, C, act, a, st);
    }
};

template <typename T>
static int gp_run(const DpigCriticDesc* d, const GpGeom& g, const GpPlan& pl, const DpigCriticParams* P, const float* real, const float* fake,
                  const float* alpha, float beta, const DpigCriticGrads* G, float* penalty, float* slopes, void* ws, void* stream) {
    constexpr bool BF = sizeof(T) == 2;
    using O = GpOps<T>;
    hipStream_t st = static_cast<hipStream_t>(stream);

    // ---- carve the workspace ------------------------------------------------------------------------------------------
    char* cur = static_cast<char*>(ws);
    auto takeb = [&](size_t bytes) { char* r = cur; cur += up256(bytes); return r; };
    float* xhat = reinterpret_cast<float*>(takeb(g.n[0] * 4));
    float* gin = reinterpret_cast<float*>(takeb(g.n[0] * 4));
    float* u0 = reinterpret_cast<float*>(takeb(g.n[0] * 4));
    T* Tn[5][S_COUNT];
    for (int l = 1; l <= 4; ++l)
        for (int s = 0; s < S_COUNT; ++s) Tn[l][s] = reinterpret_cast<T*>(takeb(g.n[l] * sizeof(T)));
    float *mean[5], *rstd[5];
    for (int l = 2; l <= 4; ++l) { mean[l] = reinterpret_cast<float*>(takeb((size_t)g.B * 4)); rstd[l] = reinterpret_cast<float*>(takeb((size_t)g.B * 4)); }
    float* c0 = reinterpret_cast<float*>(takeb((size_t)g.C[4] * 4));
    float* c1 = reinterpret_cast<float*>(takeb((size_t)g.C[4] * 4));
    float* c2 = reinterpret_cast<float*>(takeb((size_t)g.C[4] * 4));
    float* fds = reinterpret_cast<float*>(takeb((size_t)g.C[4] * 4));
    float* fdo = reinterpret_cast<float*>(takeb((size_t)g.C[4] * 4));
    uint16_t *shp[5] = {}, *sht[5] = {};           // bf16 storage: plain [25][C][K] (dgrad) and per-tap transposed [25][K][C] (forward)
    for (int l = 2; l <= 4; ++l) {
        shp[l] = reinterpret_cast<uint16_t*>(takeb(pl.shadow_elems[l] * 2));
        sht[l] = reinterpret_cast<uint16_t*>(takeb(pl.shadow_elems[l] * 2));
    }
    float* ub32 = BF ? reinterpret_cast<float*>(takeb(g.n[4] * 4)) : nullptr;
    void* cws = cur; cur += pl.conv_ws;
    void* lws = cur; cur += pl.ln_ws;
    void* l2ws = cur; cur += pl.ln2_ws;
    void* csws = cur; cur += pl.cs_ws;
    void* pws = cur; cur += pl.pen_ws;
    const float la = d->lrelu_alpha, eps = d->ln_eps;
    auto PX = [&](int l) { return g.H[l] * g.W[l]; };

    // ---- the convolutions of level l (1: the 3-channel image layer; 2..4: MFMA layers), storage type T ------------------------------
    // x_img: level 1's input is an fp32 image whatever T is
    auto conv_fwd = [&](int l, const void* x, const float* bias, int act, T* y) -> int {
        DpigConvDesc c = g.cd[l];
        c.act = act;
        if (!BF) return dpig_conv2d_fwd(&c, static_cast<const float*>(x), P->w[l - 1], bias, nullptr, reinterpret_cast<float*>(y), nullptr, cws, pl.conv_ws, stream);
        if (l == 1) return dpig_conv2d_fwd_thin_bf16(&c, x, P->w[0], bias, y, stream);
        return dpig_conv2d_fwd_bf16(&c, static_cast<const uint16_t*>(x), sht[l], bias, nullptr, nullptr, reinterpret_cast<uint16_t*>(y), nullptr, cws, pl.conv_ws, stream);
    };
    // dx = dgrad(dy) (* act'(mask) when mask != null)
    auto conv_dgrad = [&](int l, const T* dy, const T* mask, void* dx) -> int {
        DpigConvDesc c = g.cd[l];
        c.act = mask ? DPIG_ACT_LRELU : DPIG_ACT_NONE;
        if (!BF) return dpig_conv2d_dgrad(&c, reinterpret_cast<const float*>(dy), P->w[l - 1], nullptr, reinterpret_cast<const float*>(mask), static_cast<float*>(dx), cws, pl.conv_ws, stream);
        if (l == 1) return dpig_conv2d_dgrad_thin_bf16(&c, dy, P->w[0], dx, stream);
        return dpig_conv2d_dgrad_bf16(&c, reinterpret_cast<const uint16_t*>(dy), shp[l], nullptr, reinterpret_cast<const uint16_t*>(mask), static_cast<uint16_t*>(dx), cws, pl.conv_ws, stream);
    };
    auto conv_wgrad = [&](int l, const void* x, const T* dy, float* dw, float bw, float* db, float bb) -> int {
        const DpigConvDesc& c = g.cd[l];
        if (!BF) return dpig_conv2d_wgrad(&c, static_cast<const float*>(x), reinterpret_cast<const float*>(dy), dw, bw, db, bb, cws, pl.conv_ws, stream);
        if (l == 1) return dpig_conv2d_wgrad_thin_bf16(&c, x, dy, dw, bw, db, bb, cws, pl.conv_ws, stream);
        return dpig_conv2d_wgrad_bf16(&c, static_cast<const uint16_t*>(x), reinterpret_cast<const uint16_t*>(dy), dw, bw, db, bb, cws, pl.conv_ws, stream);
    };
    if (BF)
        for (int l = 2; l <= 4; ++l) GP_TRY(dpig_filter_shadow_bf16(P->w[l - 1], shp[l], sht[l], 25, g.C[l - 1], g.C[l], stream));

    // ---- sweep 1: forward on xhat ---------------------------------------------------------------------------------------
    GP_TRY(dpig_gp_interpolate(real, fake, alpha, g.B, g.n[0] / g.B, xhat, stream));
    GP_TRY(conv_fwd(1, xhat, P->b[0], DPIG_ACT_LRELU, Tn[1][S_A]));          // wgan_gp.py:414-415: conv -> LeakyReLU, one epilogue
    for (int l = 2; l <= 4; ++l) {
        GP_TRY(conv_fwd(l, Tn[l - 1][S_A], P->b[l - 1], DPIG_ACT_NONE, Tn[l][S_Z]));
        GP_TRY(O::ln_fwd(Tn[l][S_Z], g.B, PX(l), g.C[l], P->ln_scale[l - 2], P->ln_offset[l - 2], eps, DPIG_ACT_LRELU, la, Tn[l][S_A], mean[l],
                         rstd[l], lws, pl.ln_ws, stream));
    }
    // ---- sweep 1: input gradient of sum(D(xhat)).  d out / d feature = w_out for every logit row (linear.py:132-146);
    //      the reshape is over the logical NCHW tensor (wgan_gp.py:433), the data is NHWC -> one transpose.
    hipLaunchKernelGGL(gp_seed_kernel<T>, dim3(gp_blocks(g.n[4])), dim3(256), 0, st, P->w_out, Tn[4][S_T], g.n[4], g.F);
    GP_TRY(check_launch("gp_seed"));
    GP_TRY(dpig_transpose12(Tn[4][S_T], Tn[4][S_DA], g.B, g.C[4], PX(4), (int)sizeof(T), stream));
    for (int l = 4; l >= 2; --l) {
        GP_TRY(O::ln_bwd(Tn[l][S_DA], Tn[l][S_Z], Tn[l][S_A], g.B, PX(l), g.C[l], P->ln_scale[l - 2], mean[l], rstd[l], DPIG_ACT_LRELU, la,
                         Tn[l][S_DZ], nullptr, nullptr, lws, pl.ln_ws, stream));
        // level 1 has no norm: LReLU' rides the dgrad epilogue
        GP_TRY(conv_dgrad(l, Tn[l][S_DZ], l == 2 ? Tn[1][S_A] : nullptr, l == 2 ? Tn[1][S_DZ] : Tn[l - 1][S_DA]));
    }
    GP_TRY(conv_dgrad(1, Tn[1][S_DZ], nullptr, gin));
    // ---- penalty and the seed of the second sweep ---------------------------------------------------------------------------
    GP_TRY(dpig_gp_penalty(gin, g.B, g.n[0] / g.B, d->lambda, penalty, u0, slopes, pws, pl.pen_ws, stream));
    if (!G) return DPIG_OK;

    // ---- sweep 2 "up": adjoint of the backward half -----------------------------------------------------------------------------
    // g = dgrad(w1, dz1): adjoint w.r.t. dz1 = conv(u0, w1); w.r.t. w1 = wgrad(x = u0, dy = dz1)
    GP_TRY(conv_fwd(1, u0, nullptr, DPIG_ACT_NONE, Tn[1][S_V]));
    GP_TRY(conv_wgrad(1, u0, Tn[1][S_DZ], G->w[0], beta, nullptr, 0.f));
    GP_TRY(O::act_bwd(Tn[1][S_V], Tn[1][S_A], Tn[1][S_UB], g.n[1] / g.C[1], g.C[1], DPIG_ACT_LRELU, la, stream));
    for (int l = 2; l <= 4; ++l) {
        GP_TRY(conv_fwd(l, Tn[l - 1][S_UB], nullptr, DPIG_ACT_NONE, Tn[l][S_V]));
        GP_TRY(conv_wgrad(l, Tn[l - 1][S_UB], Tn[l][S_DZ], G->w[l - 1], beta, nullptr, 0.f));
        // dz_l = LNbwd(dy = da_l; x = z_l, scale): adjoints w.r.t. dy (-> ub_l), x (-> zb_l) and scale
        float* dsc = (l == 4) ? c0 : (l == 3 ? c1 : c2);                // kept until the down-sweep adds its share
        GP_TRY(O::ln_bwd2(Tn[l][S_V], Tn[l][S_DA], Tn[l][S_Z], Tn[l][S_A], g.B, PX(l), g.C[l], P->ln_scale[l - 2], mean[l], rstd[l],
                          DPIG_ACT_LRELU, la, Tn[l][S_UB], Tn[l][S_ZB], dsc, l2ws, pl.ln2_ws, stream));
    }
    // da_4 = reshape(w_out): adjoint w.r.t. w_out[j] = sum over logit rows of ub_4 in NCHW order
    GP_TRY(dpig_transpose12(Tn[4][S_UB], Tn[4][S_T], g.B, PX(4), g.C[4], (int)sizeof(T), stream));
    if (BF) {
        GP_TRY(dpig_cvt_bf16_to_f32(reinterpret_cast<const uint16_t*>(Tn[4][S_T]), g.F, ub32, g.F, g.R, g.F, stream));
        GP_TRY(dpig_colsum(ub32, g.F, g.R, g.F, G->w_out, beta, csws, pl.cs_ws, stream));
    } else {
        GP_TRY(dpig_colsum(reinterpret_cast<const float*>(Tn[4][S_T]), g.F, g.R, g.F, G->w_out, beta, csws, pl.cs_ws, stream));
    }

    // ---- sweep 3 "down": zb_l flows down the forward graph ------------------------------------------------------------------------
    // level 4: LayerNorm 4's scale gets only the second-order share; its offset none.
    hipLaunchKernelGGL(gp_acc_kernel<float>, dim3(gp_blocks(g.C[4])), dim3(256), 0, st, G->ln_scale[2], beta, (const float*)c0, (const float*)nullptr, (long)g.C[4]);
    hipLaunchKernelGGL(gp_acc_kernel<float>, dim3(gp_blocks(g.C[4])), dim3(256), 0, st, G->ln_offset[2], beta, (const float*)nullptr, (const float*)nullptr, (long)g.C[4]);
    GP_TRY(check_launch("gp_acc"));
    for (int l = 4; l >= 2; --l) {
        T* zs = (l == 4) ? Tn[4][S_ZB] : Tn[l][S_ZS];                   // what arrived at z_l in total
        // z_l = conv(a_{l-1}, w_l) + b_l
        GP_TRY(conv_wgrad(l, Tn[l - 1][S_A], zs, G->w[l - 1], 1.0f, G->b[l - 1], beta));
        if (l > 2) {
            GP_TRY(conv_dgrad(l, zs, nullptr, Tn[l - 1][S_T]));
            // a_{l-1} = LReLU(LN(z_{l-1})): first-order LayerNorm backward; dx joins zb_{l-1}
            float* dsc2 = (l - 1 == 3) ? c1 : c2;
            GP_TRY(O::ln_bwd(Tn[l - 1][S_T], Tn[l - 1][S_Z], Tn[l - 1][S_A], g.B, PX(l - 1), g.C[l - 1], P->ln_scale[l - 3], mean[l - 1],
                             rstd[l - 1], DPIG_ACT_LRELU, la, Tn[l - 1][S_F], fds, fdo, lws, pl.ln_ws, stream));
            hipLaunchKernelGGL(gp_acc_kernel<float>, dim3(gp_blocks(g.C[l - 1])), dim3(256), 0, st, G->ln_scale[l - 3], beta, (const float*)dsc2,
                               (const float*)fds, (long)g.C[l - 1]);
            hipLaunchKernelGGL(gp_acc_kernel<float>, dim3(gp_blocks(g.C[l - 1])), dim3(256), 0, st, G->ln_offset[l - 3], beta, (const float*)fdo,
                               (const float*)nullptr, (long)g.C[l - 1]);
            hipLaunchKernelGGL(gp_acc_kernel<T>, dim3(gp_blocks(g.n[l - 1])), dim3(256), 0, st, Tn[l - 1][S_ZS], 0.0f, (const T*)Tn[l - 1][S_ZB],
                               (const T*)Tn[l - 1][S_F], g.n[l - 1]);
            GP_TRY(check_launch("gp_acc"));
        } else {
            // a_1 = LReLU(z_1): the mask rides the dgrad epilogue; then z_1 = conv(xhat, w_1) + b_1
            GP_TRY(conv_dgrad(2, zs, Tn[1][S_A], Tn[1][S_ZS]));
            GP_TRY(conv_wgrad(1, xhat, Tn[1][S_ZS], G->w[0], 1.0f, G->b[0], beta));
        }
    }
    return DPIG_OK;
}

}  // namespace dpig

using namespace dpig;

extern "C" size_t dpig_gp_double_backward_workspace_bytes(const DpigCriticDesc* d) {
    GpGeom g;
    if (gp_geom(d, &g)) return 0;
    GpPlan p;
    return gp_layout(g, &p);
}

// Where the call leaves a tensor inside the caller's workspace: level 0 = the fp32 images (slot 0 xhat, 1 g = dD/dxhat, 2 u0 = dpenalty/dg),
// levels 1..4 = the critic's activations and adjoints (slots as the enum above; fp32, or bf16 with DPIG_COMPUTE_BF16_STORE).
extern "C" int dpig_gp_double_backward_slot(const DpigCriticDesc* d, int level, int slot, size_t* offset, size_t* bytes) {
    GpGeom g;
    GP_TRY(gp_geom(d, &g));
    if (!offset || !bytes || level < 0 || level > 4 || slot < 0 || slot >= (level == 0 ? 3 : (int)S_COUNT))
        return fail(DPIG_EINVAL, "gp_double_backward_slot: no such tensor");
    const size_t es = g.bf16 ? 2 : 4;
    size_t off = 0;
    if (level == 0) {
        off = up256(g.n[0] * 4) * slot;
        *bytes = g.n[0] * 4;
    } else {
        off = up256(g.n[0] * 4) * 3;
        for (int l = 1; l < level; ++l) off += up256(g.n[l] * es) * S_COUNT;
        off += up256(g.n[level] * es) * slot;
        *bytes = g.n[level] * es;
    }
    *offset = off;
    return DPIG_OK;
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
    if (g.bf16) return gp_run<gp_bf16>(d, g, pl, P, real, fake, alpha, beta, G, penalty, slopes, ws, stream);
    return gp_run<float>(d, g, pl, P, real, fake, alpha, beta, G, penalty, slopes, ws, stream);
}
