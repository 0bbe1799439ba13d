"""TEST INFRASTRUCTURE (never imported by the product path).

CPU restatement of the WGAN-GP gradient-penalty term of the DCGAN critic AS THREE EXPLICIT SWEEPS, the way
`csrc/dpig_gp.hip::dpig_gp_double_backward` evaluates it, with an optional `store` function applied at every point where the
library stores a tensor:

    reference graph   trainer.py:222-236 (interpolates, tf.gradients(D(interpolates)), penalty) over wgan_gp.py:407-440
                      (Conv5x5s2 -> LReLU -> [Conv5x5s2 -> LayerNorm -> LReLU] x3 -> reshape -> Linear) and the part of
                      Optimizer.minimize(disc_cost) (trainer.py:131-140) that differentiates the penalty w.r.t. the critic variables.

Two uses:
  * store = identity: the sweeps must reproduce `oracle.models.gradient_penalty` differentiated by torch's double backward (an independent
    derivation of the same quantity) to fp64 round-off -- `tests/test_oracle.py::test_gp_sweeps_equal_double_backward` pins the
    restatement on the CPU;
  * store = round-to-bf16: what the 'bf16' storage mode of the library computes (DPIG_COMPUTE_BF16_STORE: every level tensor of the three
    sweeps is a bf16 tensor, all arithmetic in between is fp32 / here fp64).  The HIP result is held against THIS at a tight bar
    (tests/test_variants_gpu.py::test_fused_gp_double_backward_bf16_storage); the distance between this and the exact oracle is the
    price of bf16 storage itself, measured on the CPU by the same test.

Every op is its textbook definition evaluated by torch in fp64; the adjoints of the convolutions are taken by torch.autograd.grad of the
forward op (exact transposes), LayerNorm's backward is written out and its second-order terms are torch's derivative of that expression.
"""
import torch

from . import ops as O


def bf16_round(t):
    """Round-to-nearest-even to bfloat16, returned in the input's dtype (what a bf16 tensor store + load does)."""
    return t.detach().float().to(torch.bfloat16).to(t.dtype)


def _conv(x, w, b=None):
    return O.conv2d_same(x, w, b, 2)


def _conv_dgrad(dy, w, x_shape):
    x = torch.zeros(x_shape, dtype=dy.dtype, requires_grad=True)
    (dx,) = torch.autograd.grad(_conv(x, w), x, dy)
    return dx.detach()


def _conv_wgrad(x, dy, w_shape):
    w = torch.zeros(w_shape, dtype=dy.dtype, requires_grad=True)
    (dw,) = torch.autograd.grad(_conv(x, w), w, dy)
    return dw.detach()


def _lrelu_grad(y, alpha):
    return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, alpha))


def _ln_stats(x, eps):
    mean = x.mean(dim=(1, 2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    return mean, 1.0 / torch.sqrt(var + eps)


def _ln_bwd(dy, x, y, scale, mean, rstd, alpha):
    """dx of y = LReLU(LayerNorm(x)): with g = dy * LReLU'(y) * scale, dx = r (g - mean(g) - xh mean(g xh))."""
    g = dy * _lrelu_grad(y, alpha) * scale
    xh = (x - mean) * rstd
    return rstd * (g - g.mean(dim=(1, 2, 3), keepdim=True) - xh * (g * xh).mean(dim=(1, 2, 3), keepdim=True))


def gp_sweeps(params, real, fake, alpha, lam=10.0, dim=64, lrelu=0.2, eps=1e-5, store=None, prefix="", taps=None):
    """-> (penalty, {name: d penalty / d parameter}) for the critic variables `prefix + 'Discriminator.*'` in `params` (fp64 tensors).
    `store`: applied to every tensor the library's call keeps at a level of the critic (None = identity).  `taps` (dict, optional)
    receives the level tensors by the library's slot names (Z, A, DA, DZ, V, UB, ZB per level; xhat, gin, u0)."""
    st = store if store is not None else (lambda t: t)
    p = {k: v.detach() for k, v in params.items()}
    pre = prefix + "Discriminator."
    w = [None] + [p[pre + "%d.Filters" % l] for l in range(1, 5)]
    b = [None] + [p[pre + "%d.Biases" % l] for l in range(1, 5)]
    sc = [None, None] + [p[pre + "BN%d.scale" % l] for l in range(2, 5)]
    of = [None, None] + [p[pre + "BN%d.offset" % l] for l in range(2, 5)]
    w_out = p[pre + "Output.W"]
    B = real.shape[0]
    F = 8 * 4 * 8 * dim
    xhat = real + alpha.reshape(B, 1, 1, 1) * (fake - real)

    # ---- sweep 1: forward, then the input gradient of sum(D(xhat)) -------------------------------------------------------------
    Z, A, mean, rstd = [None] * 5, [None] * 5, [None] * 5, [None] * 5
    A[1] = st(O.leaky_relu(_conv(xhat, w[1], b[1]), lrelu))
    for l in range(2, 5):
        Z[l] = st(_conv(A[l - 1], w[l], b[l]))
        mean[l], rstd[l] = _ln_stats(Z[l], eps)
        A[l] = st(O.leaky_relu((Z[l] - mean[l]) * rstd[l] * sc[l] + of[l], lrelu))
    shapes = [xhat.shape] + [None if a is None else a.shape for a in A[1:]]
    n4 = A[4].numel()
    assert n4 % F == 0
    R = n4 // F
    C4, H4, W4 = A[4].shape[3], A[4].shape[1], A[4].shape[2]
    # d out / d feature: every logit row sees w_out; the rows are the NCHW-flattened tensor (wgan_gp.py:433)
    seed = st(w_out.reshape(1, F).expand(R, F).reshape(B, C4, H4, W4))       # NCHW
    DA, DZ = [None] * 5, [None] * 5
    DA[4] = st(seed.permute(0, 2, 3, 1))
    for l in range(4, 1, -1):
        DZ[l] = st(_ln_bwd(DA[l], Z[l], A[l], sc[l], mean[l], rstd[l], lrelu))
        d = _conv_dgrad(DZ[l], w[l], shapes[l - 1])
        if l == 2:
            DZ[1] = st(d * _lrelu_grad(A[1], lrelu))
        else:
            DA[l - 1] = st(d)
    gin = _conv_dgrad(DZ[1], w[1], shapes[0])
    slopes = gin.reshape(B, -1).pow(2).sum(1).sqrt()
    penalty = lam * ((slopes - 1.0) ** 2).mean()
    coef = torch.where(slopes > 0, lam * 2.0 * (slopes - 1.0) / (B * slopes), torch.zeros_like(slopes))
    u0 = coef.reshape(B, 1, 1, 1) * gin

    grads = {}
    # ---- sweep 2 "up": adjoint of the backward half -----------------------------------------------------------------------------------
    V, UB, ZB = [None] * 5, [None] * 5, [None] * 5
    dsc2 = [None] * 5
    V[1] = st(_conv(u0, w[1]))
    dw = [None] + [None] * 4
    dw[1] = _conv_wgrad(u0, DZ[1], w[1].shape)
    UB[1] = st(V[1] * _lrelu_grad(A[1], lrelu))
    for l in range(2, 5):
        V[l] = st(_conv(UB[l - 1], w[l]))
        dw[l] = _conv_wgrad(UB[l - 1], DZ[l], w[l].shape)
        # second-order LayerNorm: derivative of <ln_bwd(dy, x; scale), u> w.r.t. (dy, x, scale), statistics as functions of x
        dy_ = DA[l].clone().requires_grad_(True)
        x_ = Z[l].clone().requires_grad_(True)
        s_ = sc[l].clone().requires_grad_(True)
        m_, r_ = _ln_stats(x_, eps)
        dx_ = _ln_bwd(dy_, x_, A[l], s_, m_, r_, lrelu)
        d_dy, d_x, d_s = torch.autograd.grad((dx_ * V[l]).sum(), [dy_, x_, s_])
        UB[l], ZB[l], dsc2[l] = st(d_dy), st(d_x), d_s
    grads[pre + "Output.W"] = UB[4].permute(0, 3, 1, 2).reshape(R, F).sum(0).reshape(F, 1)

    # ---- sweep 3 "down": the x-adjoints flow down the forward graph ----------------------------------------------------------------------
    db = [None] * 5
    grads[pre + "BN4.scale"] = dsc2[4]
    grads[pre + "BN4.offset"] = torch.zeros_like(of[4])
    for l in range(4, 1, -1):
        dw[l] = dw[l] + _conv_wgrad(A[l - 1], ZB[l], w[l].shape)
        db[l] = ZB[l].sum(dim=(0, 1, 2))
        d = _conv_dgrad(ZB[l], w[l], shapes[l - 1])
        if l > 2:
            T = st(d)
            dz = T * _lrelu_grad(A[l - 1], lrelu)
            xh = (Z[l - 1] - mean[l - 1]) * rstd[l - 1]
            grads[pre + "BN%d.scale" % (l - 1)] = dsc2[l - 1] + (dz * xh).sum(dim=(0, 1, 2))
            grads[pre + "BN%d.offset" % (l - 1)] = dz.sum(dim=(0, 1, 2))
            first = st(_ln_bwd(T, Z[l - 1], A[l - 1], sc[l - 1], mean[l - 1], rstd[l - 1], lrelu))
            ZB[l - 1] = st(ZB[l - 1] + first)
        else:
            ZB[1] = st(d * _lrelu_grad(A[1], lrelu))
            dw[1] = dw[1] + _conv_wgrad(xhat, ZB[1], w[1].shape)
            db[1] = ZB[1].sum(dim=(0, 1, 2))
    if taps is not None:
        taps.update({"xhat": xhat, "gin": gin, "u0": u0})
        for nm, arr in (("Z", Z), ("A", A), ("DA", DA), ("DZ", DZ), ("V", V), ("UB", UB), ("ZB", ZB)):
            for l in range(1, 5):
                if arr[l] is not None:
                    taps["%s%d" % (nm, l)] = arr[l]
    for l in range(1, 5):
        grads[pre + "%d.Filters" % l] = dw[l]
        grads[pre + "%d.Biases" % l] = db[l]
    return penalty, grads


def _ln_bwd2(u, dy, x, y, scale, eps, alpha):
    """(d_dy, d_x, d_scale) of <ln_bwd(dy, x; scale), u>, the statistics being functions of x: torch's derivative of `_ln_bwd`."""
    dy_ = dy.clone().requires_grad_(True)
    x_ = x.clone().requires_grad_(True)
    s_ = scale.clone().requires_grad_(True)
    m_, r_ = _ln_stats(x_, eps)
    return torch.autograd.grad((_ln_bwd(dy_, x_, y, s_, m_, r_, alpha) * u).sum(), [dy_, x_, s_])


def gp_chain_links(params, t, real, fake, alpha, lam=10.0, dim=64, lrelu=0.2, eps=1e-5, prefix=""):
    """Every link of the three sweeps recomputed in fp64 FROM THE TENSORS THE LIBRARY STORED (`t`: name -> fp64 tensor, the names of
    hip_ops.gp_double_backward_tensors): yields (name of the stored result, what its inputs give in fp64, kind), kind 'store' for a tensor
    the library rounds to its storage type, 'f32' for an fp32 result (images, parameter gradients, the penalty).  Feeding each link the
    library's OWN inputs keeps the LeakyReLU masks of both sides identical -- compared end to end, one rounding flip near a unit's zero
    crossing changes that unit's mask and moves the second-order terms by O(1) (measured: tests/test_oracle.py), which says nothing about
    the kernels.  Gradient names carry the parameter's name."""
    p = {k: v.detach() for k, v in params.items()}
    pre = prefix + "Discriminator."
    w = [None] + [p[pre + "%d.Filters" % l] for l in range(1, 5)]
    b = [None] + [p[pre + "%d.Biases" % l] for l in range(1, 5)]
    sc = [None, None] + [p[pre + "BN%d.scale" % l] for l in range(2, 5)]
    of = [None, None] + [p[pre + "BN%d.offset" % l] for l in range(2, 5)]
    w_out = p[pre + "Output.W"]
    B = real.shape[0]
    F = 8 * 4 * 8 * dim
    yield "xhat", real + alpha.reshape(B, 1, 1, 1) * (fake - real), "f32"
    # sweep 1, forward
    yield "A1", O.leaky_relu(_conv(t["xhat"], w[1], b[1]), lrelu), "store"
    mean, rstd = [None] * 5, [None] * 5
    for l in range(2, 5):
        yield "Z%d" % l, _conv(t["A%d" % (l - 1)], w[l], b[l]), "store"
        mean[l], rstd[l] = _ln_stats(t["Z%d" % l], eps)
        yield "A%d" % l, O.leaky_relu((t["Z%d" % l] - mean[l]) * rstd[l] * sc[l] + of[l], lrelu), "store"
    A4 = t["A4"]
    R = A4.numel() // F
    yield "DA4", w_out.reshape(1, F).expand(R, F).reshape(B, A4.shape[3], A4.shape[1], A4.shape[2]).permute(0, 2, 3, 1), "store"
    # sweep 1, input gradient
    for l in range(4, 1, -1):
        yield "DZ%d" % l, _ln_bwd(t["DA%d" % l], t["Z%d" % l], t["A%d" % l], sc[l], mean[l], rstd[l], lrelu), "store"
        d = _conv_dgrad(t["DZ%d" % l], w[l], t["A%d" % (l - 1)].shape)
        if l == 2:
            yield "DZ1", d * _lrelu_grad(t["A1"], lrelu), "store"
        else:
            yield "DA%d" % (l - 1), d, "store"
    yield "gin", _conv_dgrad(t["DZ1"], w[1], t["xhat"].shape), "f32"
    slopes = t["gin"].reshape(B, -1).pow(2).sum(1).sqrt()
    yield "penalty", lam * ((slopes - 1.0) ** 2).mean(), "f32"
    coef = torch.where(slopes > 0, lam * 2.0 * (slopes - 1.0) / (B * slopes), torch.zeros_like(slopes))
    yield "u0", coef.reshape(B, 1, 1, 1) * t["gin"], "f32"
    # sweep 2
    yield "V1", _conv(t["u0"], w[1]), "store"
    yield "UB1", t["V1"] * _lrelu_grad(t["A1"], lrelu), "store"
    dsc2 = [None] * 5
    for l in range(2, 5):
        yield "V%d" % l, _conv(t["UB%d" % (l - 1)], w[l]), "store"
        d_dy, d_x, dsc2[l] = _ln_bwd2(t["V%d" % l], t["DA%d" % l], t["Z%d" % l], t["A%d" % l], sc[l], eps, lrelu)
        yield "UB%d" % l, d_dy, "store"
        yield "ZB%d" % l, d_x, "store"
    yield pre + "Output.W", t["UB4"].permute(0, 3, 1, 2).reshape(R, F).sum(0).reshape(F, 1), "f32"
    # sweep 3
    zs = {4: t["ZB4"], 3: t["ZS3"], 2: t["ZS2"], 1: t["ZS1"]}
    yield pre + "BN4.scale", dsc2[4], "f32"
    for l in range(4, 1, -1):
        yield pre + "%d.Filters" % l, _conv_wgrad(t["UB%d" % (l - 1)], t["DZ%d" % l], w[l].shape) + _conv_wgrad(t["A%d" % (l - 1)], zs[l], w[l].shape), "f32"
        yield pre + "%d.Biases" % l, zs[l].sum(dim=(0, 1, 2)), "f32"
        d = _conv_dgrad(zs[l], w[l], t["A%d" % (l - 1)].shape)
        if l > 2:
            k = l - 1
            yield "T%d" % k, d, "store"
            T = t["T%d" % k]
            dz = T * _lrelu_grad(t["A%d" % k], lrelu)
            xh = (t["Z%d" % k] - mean[k]) * rstd[k]
            yield pre + "BN%d.scale" % k, dsc2[k] + (dz * xh).sum(dim=(0, 1, 2)), "f32"
            yield pre + "BN%d.offset" % k, dz.sum(dim=(0, 1, 2)), "f32"
            yield "F%d" % k, _ln_bwd(T, t["Z%d" % k], t["A%d" % k], sc[k], mean[k], rstd[k], lrelu), "store"
            yield "ZS%d" % k, t["ZB%d" % k] + t["F%d" % k], "store"
        else:
            yield "ZS1", d * _lrelu_grad(t["A1"], lrelu), "store"
    yield pre + "1.Filters", _conv_wgrad(t["u0"], t["DZ1"], w[1].shape) + _conv_wgrad(t["xhat"], t["ZS1"], w[1].shape), "f32"
    yield pre + "1.Biases", t["ZS1"].sum(dim=(0, 1, 2)), "f32"
