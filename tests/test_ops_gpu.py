"""GPU parity of the non-conv kernels (through the C ABI) against the CPU oracle (fp64).
Tolerance 2e-5 relative to max|ref| unless noted (fp32 kernels vs fp64 oracle)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g, dtype=torch.float64) * (hi - lo) + lo


def _close(got, ref, tol=TOL):
    ref = ref.detach().double()
    err = (got.detach().double().cpu() - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    assert err <= tol * scale, "max err %.3e vs scale %.3e" % (err, scale)


@pytest.mark.parametrize("shape", [(2, 32, 16, 128), (16, 8, 4, 512), (3, 5, 7, 20)])
@pytest.mark.parametrize("act", [0, 2])
def test_batchnorm_fwd_bwd(dev, shape, act):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    C = shape[-1]
    x = (_rand(shape, 1) * 3 + 0.7).requires_grad_(True)
    scale = _rand((C,), 2, 0.5, 1.5).requires_grad_(True)
    offset = _rand((C,), 3).requires_grad_(True)
    y = O.batchnorm_train(x, scale, offset)
    if act == 2:
        y = O.leaky_relu(y, 0.2)
    dy = _rand(shape, 4)
    y.backward(dy)
    xg = x.detach().float().to(dev)
    yg, mean, rstd = H.bn_fwd(xg, scale.detach().float().to(dev), offset.detach().float().to(dev), 1e-5, act, 0.2)
    _close(yg, y)
    dx, ds, do = H.bn_bwd(dy.float().to(dev), xg, yg, scale.detach().float().to(dev), mean, rstd, act, 0.2)
    _close(dx, x.grad, 5e-5)
    _close(ds, scale.grad, 5e-5)
    _close(do, offset.grad, 5e-5)


@pytest.mark.parametrize("shape", [(3, 20, 20, 64, 128, 5, 2), (2, 17, 9, 32, 136, 3, 1), (1, 8, 8, 64, 64, 5, 2)])
@pytest.mark.parametrize("compute", ["f32", "bf16x3"])
def test_conv_epilogue_bn_statistics(dev, shape, compute):
    """conv2d.py:106-120 followed by batchnorm.py:30 with the statistics carried by the conv's epilogue
    (dpig_conv2d_fwd_stats -> dpig_bn_stats_finalize -> dpig_bn_apply): conv + bias, batch mean, rstd and the normalised
    LeakyReLU output against the fp64 oracle conv -> batchnorm_train chain; a large offset in the bias makes a one-pass
    E[x^2]-E[x]^2 lose ~4 digits (the per-tile centred sums do not); ragged last row tile; bit-for-bit repeatable.
    Split-K plans (the heuristic one on few row tiles, forced 2 / 3 way) carry them too: their reduction pass leaves them."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = shape
    x = _rand((N, Hh, W, C), 1)
    w = _rand((k, k, C, K), 2) * 0.1
    b = _rand((K,), 3) * 0.5 + 40.0
    scale = _rand((K,), 4, 0.5, 1.5)
    offset = _rand((K,), 5)
    pre = O.conv2d_same(x, w, b, s)
    ref = O.leaky_relu(O.batchnorm_train(pre, scale, offset), 0.2)
    xg, wg, bg = x.float().to(dev), w.float().to(dev), b.float().to(dev)
    H.set_compute(compute)
    try:
        y, st = H.conv2d_fwd_stats(xg, wg, bg, stride=s, split_k=1)
        assert st is not None and st[0].shape == ((pre.numel() // K + 127) // 128, 2, K)
        _close(y, pre)
        out, mean, rstd = H.bn_fwd(y, scale.float().to(dev), offset.float().to(dev), 1e-5, 2, 0.2, stats=st)
        rows = pre.reshape(-1, K)
        _close(mean, rows.mean(0), 1e-6)
        _close(rstd, 1.0 / torch.sqrt(rows.var(0, unbiased=False) + 1e-5), 2e-5 if compute == "f32" else 1e-4)
        _close(out, ref, 2e-5 if compute == "f32" else 1e-4)
        out2, mean2, rstd2 = H.bn_fwd(y, scale.float().to(dev), offset.float().to(dev), 1e-5, 2, 0.2)     # the op's own passes
        assert float((mean2 - mean).abs().max()) <= 1e-6 * float(mean.abs().max())
        assert float((rstd2 - rstd).abs().max()) <= 1e-5 * float(rstd.abs().max())
        y_b, st_b = H.conv2d_fwd_stats(xg, wg, bg, stride=s, split_k=1)
        assert torch.equal(y_b, y) and torch.equal(st_b[0], st[0])
        # split-K plans (the heuristic one on these few tiles, and forced 2 / 3 way): the reduction pass that sums the partials leaves
        # the same per-tile statistics of the values it stores
        for sk in (0, 2, 3):
            y_s, st_s = H.conv2d_fwd_stats(xg, wg, bg, stride=s, split_k=sk)
            if sk == 0 and st_s is None:            # host policy: a heuristic split over few tiles takes bn_fwd's own passes
                _close(y_s, pre)
                continue
            assert st_s is not None and st_s[0].shape == st[0].shape
            _close(y_s, pre)
            out_s, mean_s, rstd_s = H.bn_fwd(y_s, scale.float().to(dev), offset.float().to(dev), 1e-5, 2, 0.2, stats=st_s)
            _close(mean_s, rows.mean(0), 1e-6)
            _close(rstd_s, 1.0 / torch.sqrt(rows.var(0, unbiased=False) + 1e-5), 2e-5 if compute == "f32" else 1e-4)
            _close(out_s, ref, 2e-5 if compute == "f32" else 1e-4)
            y_s2, st_s2 = H.conv2d_fwd_stats(xg, wg, bg, stride=s, split_k=sk)
            assert torch.equal(y_s2, y_s) and torch.equal(st_s2[0], st_s[0])
    finally:
        H.set_compute("f32")


def test_keypoint_fed_stem_falls_back_to_the_dense_conv_for_widths_the_kernel_does_not_take(dev):
    """conv_hidden_num = 96: K / 4 = 24 lanes per pixel does not divide 256, dpig_pose_stem_wgrad would refuse it at the first backward
    pass -- `tiled_emb_conv` rasterises the map and runs the dense path instead (same oracle bar, forward and gradients)."""
    from dpig_amd import autograd as A

    class W(object):
        shape = (3, 3, 38, 96)
        def data_ptr(self): return 0
    class P(object):
        shape = (3, 24, 16, 18)
    assert not A._keypoint_stem_ok(P(), W())
    test_keypoint_fed_stem_conv_equals_the_dense_conv_on_the_rasterised_map(dev, "f32", K=96)


def test_discriminator_uses_conv_epilogue_bn_statistics(dev):
    """DCGANDiscriminator (wgan_gp.py:407-440) in BatchNorm mode asks its convs for the statistics; on a batch large enough
    for un-split plans the fused path must give the logits (and input gradient) of the path with separate statistics passes."""
    import dpig_amd.hip_ops as H
    import dpig_amd.tflib as lib
    from dpig_amd import slim
    from dpig_amd.wgan_gp import WGAN_GP
    lib.delete_all_params(); slim.reset_scopes()
    lib.set_device(dev)
    wg = WGAN_GP(MODE='dcgan', DIM=32, BATCH_SIZE=16)
    # DeepFashion-sized critic input (trainer_256.py): Discriminator.2 writes 16 x 64 x 64 = 65536 rows = 512 row tiles
    x = torch.randn(16, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).requires_grad_(True)
    calls = []
    orig = H.conv2d_fwd_stats

    def spy(*a, **k):
        r = orig(*a, **k)
        calls.append(r[1] is not None)
        return r
    H.conv2d_fwd_stats = spy
    try:
        out = wg.DCGANDiscriminator(x, dim=32)
        g, = torch.autograd.grad(out.sum(), x)
    finally:
        H.conv2d_fwd_stats = orig
    assert len(calls) == 3 and any(calls), calls          # the three BN-fed convs asked; at least the largest carried them
    H.conv2d_fwd_stats = lambda xx, w, b=None, stride=1, split_k=0: (H.conv2d_fwd(xx, w, b, stride=stride, split_k=split_k), None)
    try:
        out0 = wg.DCGANDiscriminator(x, dim=32)
        g0, = torch.autograd.grad(out0.sum(), x)
    finally:
        H.conv2d_fwd_stats = orig
    assert float((out - out0).abs().max()) <= 1e-4 * float(out0.abs().max())
    # The two statistic paths sum in different orders; a LeakyReLU unit whose normalised input sits within that round-off of zero
    # takes the other slope (tests/test_model_gpu.py docstring): one such unit moves the input gradient of the pixels in its
    # receptive field by a few per cent.  So: all but a sliver of the gradient agrees tightly, and nothing is far off.
    dg = (g - g0).abs()
    assert float((dg <= 1e-3 * float(g0.abs().max())).float().mean()) >= 0.995
    assert float(dg.max()) <= 1e-1 * float(g0.abs().max())


@pytest.mark.parametrize("shape", [(2, 32, 16, 128), (4, 8, 4, 512), (3, 5, 7, 20)])
def test_layernorm_fwd_bwd(dev, shape):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    C = shape[-1]
    x = (_rand(shape, 1) * 2 - 0.3).requires_grad_(True)
    scale = _rand((C,), 2, 0.5, 1.5).requires_grad_(True)
    offset = _rand((C,), 3).requires_grad_(True)
    y = O.leaky_relu(O.layernorm(x, scale, offset), 0.2)
    dy = _rand(shape, 4)
    y.backward(dy)
    xg = x.detach().float().to(dev)
    yg, mean, rstd = H.ln_fwd(xg, scale.detach().float().to(dev), offset.detach().float().to(dev), 1e-5, 2, 0.2)
    _close(yg, y)
    dx, ds, do = H.ln_bwd(dy.float().to(dev), xg, yg, scale.detach().float().to(dev), mean, rstd, 2, 0.2)
    _close(dx, x.grad, 5e-5)
    _close(ds, scale.grad, 5e-5)
    _close(do, offset.grad, 5e-5)


@pytest.mark.parametrize("mkn", [(16, 20480, 128), (14, 5760, 32), (16, 64, 4096), (2, 16384, 1), (64, 224, 512),
                                 (5, 37, 11)])
def test_linear(dev, mkn):
    import dpig_amd.hip_ops as H
    M, K, N = mkn
    x = _rand((M, K), 1).requires_grad_(True)
    w = (_rand((K, N), 2) * 0.1).requires_grad_(True)
    b = _rand((N,), 3)
    y = torch.relu(x @ w + b)
    dy = _rand((M, N), 4)
    (x @ w + b).backward(dy)
    yg = H.linear_fwd(x.detach().float().to(dev), w.detach().float().to(dev), b.float().to(dev), act=1)
    _close(yg, y)
    _close(H.linear_dgrad(dy.float().to(dev), w.detach().float().to(dev)), x.grad)
    _close(H.linear_wgrad(x.detach().float().to(dev), dy.float().to(dev)), w.grad)


def test_crop_and_resize(dev):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C = 3, 32, 16, 24
    img = _rand((N, Hh, W, C), 1).requires_grad_(True)
    # pixel boxes /H,/W (models.py:410-413), incl. the [0,0,1,1] sentinel and a box touching the border
    px = torch.tensor([[0, 0, 1, 1], [3, 2, 20, 9], [10, 5, 31, 15], [0, 0, 31, 15], [7, 7, 8, 8], [5, 1, 30, 14]],
                      dtype=torch.float64)
    boxes = px / torch.tensor([Hh, W, Hh, W], dtype=torch.float64)
    # one out-of-range box (extrapolation 0)
    boxes = torch.cat([boxes, torch.tensor([[-0.2, 0.1, 1.3, 0.9]], dtype=torch.float64)])
    box_ind = torch.tensor([0, 1, 2, 0, 1, 2, 1])
    ref = O.crop_and_resize(img, boxes, box_ind, 12, 12)
    dout = _rand(tuple(ref.shape), 2)
    ref.backward(dout)
    got = H.crop_resize_fwd(img.detach().float().to(dev), boxes.float().to(dev), box_ind.to(dev), 12, 12)
    _close(got, ref)
    dimg = H.crop_resize_bwd(dout.float().to(dev), boxes.float().to(dev), box_ind.to(dev), (N, Hh, W, C))
    _close(dimg, img.grad, 5e-5)


@pytest.mark.parametrize("C", [128, 6])
def test_crop_and_resize_roi_shapes(dev, C):
    """ROI-tower shapes (models.py:410-415): 7 parts x B boxes, 48x48 crops of a 128x64 map; also degenerate
    (zero-height), flipped (y2<y1) and partly outside boxes.  The backward is a gather: bitwise repeatable."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    B, Hh, W = 2, 128, 64
    g = torch.Generator().manual_seed(5)
    img = _rand((B, Hh, W, C), 3).requires_grad_(True)
    y1 = torch.randint(0, 64, (7 * B,), generator=g).double()
    x1 = torch.randint(0, 32, (7 * B,), generator=g).double()
    hh = torch.randint(8, 64, (7 * B,), generator=g).double()
    ww = torch.randint(8, 32, (7 * B,), generator=g).double()
    px = torch.stack([y1, x1, y1 + hh, x1 + ww], 1)
    px[0] = torch.tensor([0., 0., 1., 1.])          # invisible-part sentinel
    px[1] = torch.tensor([20., 5., 20., 30.])       # zero height
    px[2] = torch.tensor([90., 40., 30., 10.])      # flipped
    px[3] = torch.tensor([100., 50., 160., 80.])    # runs off the image
    boxes = px / torch.tensor([Hh, W, Hh, W], dtype=torch.float64)
    box_ind = torch.arange(B).repeat(7)
    ref = O.crop_and_resize(img, boxes, box_ind, 48, 48)
    dout = _rand(tuple(ref.shape), 4)
    ref.backward(dout)
    bd, bi = boxes.float().to(dev), box_ind.to(dev)
    _close(H.crop_resize_fwd(img.detach().float().to(dev), bd, bi, 48, 48), ref)
    d1 = H.crop_resize_bwd(dout.float().to(dev), bd, bi, (B, Hh, W, C))
    d2 = H.crop_resize_bwd(dout.float().to(dev), bd, bi, (B, Hh, W, C))
    assert torch.equal(d1, d2)
    _close(d1, img.grad, 1e-4)


def test_upsample_act_colsum(dev):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    x = _rand((2, 5, 3, 12), 1).requires_grad_(True)
    y = O.upsample2x(x)
    dy = _rand(tuple(y.shape), 2)
    y.backward(dy)
    _close(H.upsample2x_fwd(x.detach().float().to(dev)), y)
    _close(H.upsample2x_bwd(dy.float().to(dev)), x.grad)
    a = _rand((4, 6, 5, 36), 3)
    yy = O.leaky_relu(a, 0.2)
    _close(H.act_fwd(a.float().to(dev), 2, 0.2), yy)
    g = _rand((4, 6, 5, 36), 4)
    _close(H.act_bwd(g.float().to(dev), yy.float().to(dev), 2, 0.2), g * torch.where(yy > 0, 1.0, 0.2))
    _close(H.colsum(a.float().to(dev)), a.reshape(-1, 36).sum(0))
    big = _rand((16, 64, 32, 64), 5)
    _close(H.colsum(big.float().to(dev)), big.reshape(-1, 64).sum(0), 1e-4)


def test_losses(dev):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    x = (_rand((37,), 1) * 6).requires_grad_(True)
    for label in (0.0, 1.0):
        x.grad = None
        ref = O.sigmoid_cross_entropy_with_logits(x, torch.full_like(x, label)).mean()
        ref.backward()
        out, dl = H.sce_mean(x.detach().float().to(dev), label, want_grad=True, scale=1.0)
        _close(out, ref.reshape(1))
        _close(dl, x.grad)
    a = _rand((2, 16, 8, 3), 2).requires_grad_(True)
    b = _rand((2, 16, 8, 3), 3)
    ref = (a - b).abs().mean()
    ref.backward()
    out, da = H.l1_mean(a.detach().float().to(dev), b.float().to(dev), want_grad=True, scale=1.0)
    _close(out, ref.reshape(1))
    _close(da, a.grad)


def test_wgan_and_lsgan_loss_terms(dev):
    """trainer.py:218-220 (wgan: +-mean D) and :246-248 (lsgan: mean (D - 1)^2, mean D^2) with their gradients, and
    `gan_loss` in the four modes against the oracle's restatement."""
    import dpig_amd.autograd as A
    from dpig_amd.trainer import gan_loss
    from dpig_amd.wgan_gp import WGAN_GP
    from oracle import models as OM
    g = torch.Generator().manual_seed(7)
    real = (torch.rand(37, generator=g, dtype=torch.float64) * 6 - 3)
    fake = (torch.rand(37, generator=g, dtype=torch.float64) * 6 - 3)
    for mode in ("wgan", "lsgan", "dcgan"):
        rr, ff = real.clone().requires_grad_(True), fake.clone().requires_grad_(True)
        g_ref, d_ref = OM.gan_loss(mode, rr, ff)
        gr, gf = torch.autograd.grad(d_ref, [rr, ff])
        rg, fg = real.float().to(dev).requires_grad_(True), fake.float().to(dev).requires_grad_(True)
        g_got, d_got = gan_loss(WGAN_GP(MODE=mode), rg, fg)
        assert abs(g_got.item() - g_ref.item()) < 1e-5 * max(1.0, abs(g_ref.item()))
        assert abs(d_got.item() - d_ref.item()) < 1e-5 * max(1.0, abs(d_ref.item()))
        d_got.backward()
        assert (rg.grad.double().cpu() - gr).abs().max().item() < 1e-6
        assert (fg.grad.double().cpu() - gf).abs().max().item() < 1e-6
    x = real.float().to(dev).requires_grad_(True)
    (A.logit_sq_mean(x, 1.0) * 3.0).backward()
    assert (x.grad.double().cpu() - 3.0 * 2.0 * (real - 1.0) / 37).abs().max().item() < 1e-6


def test_tf_adam(dev):
    import dpig_amd.hip_ops as H
    from oracle import naive
    n = 1003
    p0 = _rand((n,), 1).numpy()
    grads = [_rand((n,), 10 + i).numpy() * 0.1 for i in range(3)]
    ref = naive.tf_adam(p0, grads, lr=2e-3, beta1=0.5, beta2=0.999, eps=1e-8)
    pad = (n + 3) // 4 * 4
    p = torch.zeros(pad, device=dev); p[:n] = torch.tensor(p0, dtype=torch.float32)
    m = torch.zeros(pad, device=dev); v = torch.zeros(pad, device=dev)
    lr = torch.full((1,), 2e-3, device=dev)
    for t, g in enumerate(grads, start=1):
        gg = torch.zeros(pad, device=dev); gg[:n] = torch.tensor(g, dtype=torch.float32)
        H.adam_step(p, gg, m, v, lr, 0.5, 0.999, 1e-8, t)
    _close(p[:n], torch.tensor(ref), 1e-5)


def test_tflib_ops_boundary(dev):
    """The reference-signature ops: NCHW logical in/out, shared params by name, fused epilogues."""
    import dpig_amd.tflib as lib
    import dpig_amd.tflib.ops  # noqa
    from oracle import ops as O
    lib.delete_all_params()
    lib.set_device(dev)
    np.random.seed(5)
    x = _rand((2, 16, 8, 6), 1)                       # NHWC
    x_nchw = x.float().to(dev).permute(0, 3, 1, 2)    # logical NCHW view (trainer.py:601)
    lib.ops.conv2d.set_weights_stdev(0.02)
    y = lib.ops.conv2d.Conv2D('T.1', 6, 16, 5, x_nchw, stride=2)
    lib.ops.conv2d.unset_weights_stdev()
    assert tuple(y.shape) == (2, 16, 8, 4)
    w = lib.param('T.1.Filters'); b = lib.param('T.1.Biases')
    assert float(w.abs().max()) <= 0.02 * np.sqrt(3) + 1e-7
    ref = O.conv2d_same(x, w.detach().double().cpu(), b.detach().double().cpu(), 2)
    _close(y.permute(0, 2, 3, 1), ref)
    y2 = lib.ops.conv2d.Conv2D('T.1', 6, 16, 5, x_nchw, stride=2)      # same name -> same weights
    assert torch.equal(y, y2)
    z = lib.ops.batchnorm.Batchnorm('T.BN', [0, 2, 3], y)
    sc = lib.param('T.BN.scale'); of = lib.param('T.BN.offset')
    _close(z.permute(0, 2, 3, 1), O.batchnorm_train(ref, sc.detach().double().cpu(), of.detach().double().cpu()), 1e-4)
    assert 'T.BN.moving_mean' in lib._params and not lib.param('T.BN.moving_mean').requires_grad
    ln = lib.ops.layernorm.Layernorm('T.LN', [1, 2, 3], y)
    _close(ln.permute(0, 2, 3, 1), O.layernorm(ref, torch.ones(16, dtype=torch.float64), torch.zeros(16, dtype=torch.float64)), 1e-4)
    flat = y.reshape(2, -1)                                           # logical (c,h,w) flatten
    lin = lib.ops.linear.Linear('T.Out', 16 * 8 * 4, 3, flat)
    wl = lib.param('T.Out.W').detach().double().cpu()
    _close(lin, ref.permute(0, 3, 1, 2).reshape(2, -1) @ wl)
    dc = lib.ops.deconv2d.Deconv2D('T.D', 16, 5, 5, y)
    assert tuple(dc.shape) == (2, 5, 16, 8)
    wd = lib.param('T.D.Filters').detach().double().cpu()
    _close(dc.permute(0, 2, 3, 1), O.conv2d_transpose_same(ref, wd, None, 2), 1e-4)
    with pytest.raises(Exception):
        lib.ops.deconv2d.Deconv2D('T.D2', 16, 5, 5, y, mask_type=('a', 1))
    with pytest.raises(Exception):
        lib.ops.linear.Linear('T.bad', 4, 4, flat[:, :4], initialization='nope')
    assert len(lib.params_with_name('T.')) == len([n for n in lib._params if 'T.' in n])
    lib.delete_all_params()


def test_rmsprop_and_clip(dev):
    import dpig_amd.hip_ops as H
    from oracle import models as OM
    n = 777
    p = _rand((n,), 1); g1 = _rand((n,), 2) * 0.3; g2 = _rand((n,), 3) * 0.3
    rp, rms, rmom = p.clone(), torch.ones(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    for g in (g1, g2):
        rp, rms, rmom = OM.tf_rmsprop_step(rp, g, rms, rmom, 2e-3)
    gp = p.float().to(dev); ms = torch.ones(n, device=dev); mom = torch.zeros(n, device=dev)
    lr = torch.full((1,), 2e-3, device=dev)
    for g in (g1, g2):
        H.rmsprop_step(gp, g.float().to(dev), ms, mom, lr)
    _close(gp, rp, 1e-5)
    H.clip_(gp, -0.01, 0.01)
    _close(gp, rp.clamp(-0.01, 0.01), 1e-5)
    assert float(gp.max()) <= 0.01 and float(gp.min()) >= -0.01


def test_deconv_backward(dev):
    """Deconv2D (conv2d_transpose) gradients: d/dx = F(dy), d/dw = wgrad_F(dy, x)."""
    import dpig_amd.autograd as A
    from oracle import ops as O
    x = _rand((2, 4, 3, 8), 1).requires_grad_(True)
    w = (_rand((5, 5, 6, 8), 2) * 0.2).requires_grad_(True)     # (k, k, Cout, Cin)
    y = O.conv2d_transpose_same(x, w, None, 2)
    dy = _rand(tuple(y.shape), 3)
    y.backward(dy)
    gx = x.detach().float().to(dev).requires_grad_(True)
    gw = w.detach().float().to(dev).requires_grad_(True)
    gy = A.conv2d_transpose(gx, gw, None)
    _close(gy, y, 1e-4)
    gy.backward(dy.float().to(dev))
    _close(gx.grad, x.grad, 1e-4)
    _close(gw.grad, w.grad, 1e-4)


def test_pose_maps_equal_reference_rasteriser(dev):
    """The device rasteriser against maps produced by the reference's own numpy code (utils.py py_poseInflate, the
    function tester.py:399-400 calls; fixture tests/golden/pose_reference.npz, generated by make_pose_golden.py)."""
    import os
    import dpig_amd.utils as U
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pose_reference.npz"))
    for name in ("pixel_128x64", "normalized_128x64", "normalized_256x256"):
        rcv = z[name + "/rcv"].astype(np.float32)
        norm, Hh, W = (int(v) for v in z[name + "/meta"])
        B, K = rcv.shape[:2]
        want = np.unpackbits(z[name + "/bits"])[:B * Hh * W * K].reshape(B, Hh, W, K).astype(np.float32) * 2 - 1
        t = torch.from_numpy(rcv.reshape(B, K * 3)).to(dev)
        fused = U.pose_target_from_rcv(t, K, bool(norm), Hh, W)
        assert torch.equal(fused.cpu(), torch.from_numpy(want)), name
        chained = U.tf_poseInflate(U.coord2channel_simple_rcv(t, K, bool(norm), Hh, W), K, 4, Hh, W)
        assert torch.equal(chained.cpu(), torch.from_numpy(want)), name


@pytest.mark.parametrize("normalized", [True, False])
def test_pose_maps(dev, normalized):
    """coord2channel_simple_rcv / tf_poseInflate (utils.py:237-318) and the fused rasteriser."""
    import dpig_amd.utils as U
    from oracle import ops as O
    rng = np.random.default_rng(9)
    B, K, Hh, W = 4, 18, 128, 64
    rcv = np.zeros((B, K, 3), dtype=np.float32)
    if normalized:
        rcv[..., 0] = rng.uniform(-1.1, 1.1, (B, K)); rcv[..., 1] = rng.uniform(-1.1, 1.1, (B, K))
    else:
        rcv[..., 0] = rng.integers(0, Hh, (B, K)); rcv[..., 1] = rng.integers(0, W, (B, K))
    rcv[..., 2] = (rng.uniform(size=(B, K)) < 0.85)
    rcv[0, 0] = (-1.0, -1.0, 1.0) if normalized else (0.0, 0.0, 1.0)
    rcv[0, 1] = (1.0, 1.0, 1.0) if normalized else (Hh - 1.0, W - 1.0, 1.0)
    rcv[1, 2, 2] = 0.5
    t = torch.from_numpy(rcv.reshape(B, K * 3))
    ref_pts = O.coord2channel_simple_rcv(t.double(), K, normalized, Hh, W)
    ref = O.tf_poseInflate(ref_pts, K, 4, Hh, W)
    pts = U.coord2channel_simple_rcv(t.to(dev), K, normalized, Hh, W)
    assert torch.equal(pts.cpu().double(), ref_pts)
    assert torch.equal(U.tf_poseInflate(pts, K, 4, Hh, W).cpu().double(), ref)
    assert torch.equal(U.pose_target_from_rcv(t.to(dev), K, normalized, Hh, W).cpu().double(), ref)
    # tf_poseInflate is defined on any [-1,1] map, not only on single points
    dense = torch.from_numpy(rng.choice([-1.0, -0.5, 0.25, 1.0], size=(2, 20, 12, 5))).float()
    _close(U.tf_poseInflate(dense.to(dev), 5, 4, 20, 12), O.tf_poseInflate(dense.double(), 5, 4, 20, 12), 1e-6)
    if not normalized:                              # keypoints outside the image are dropped
        rcv[2, 3] = (-3.0, 5.0, 1.0); rcv[2, 4] = (5.0, W + 2.0, 1.0)
        out = U.pose_target_from_rcv(torch.from_numpy(rcv.reshape(B, K * 3)).to(dev), K, False, Hh, W)
        assert float(out[2, :, :, 3].max()) == -1.0 and float(out[2, :, :, 4].max()) == -1.0


def test_ssim_metric(dev):
    """the SSIM trainer.generate() logs (skimage on gray uint8 images) computed on the device."""
    import dpig_amd.utils as U
    from oracle import ops as O
    rng = np.random.default_rng(4)
    B, Hh, W = 5, 128, 64
    x = rng.uniform(-1, 1, (B, Hh, W, 3)).astype(np.float32)
    G = np.clip((x + 1) * 127.5 + rng.normal(0, 25, x.shape), -20, 280).astype(np.float32)     # incl. values to clip
    G[0] = (x[0] + 1) * 127.5                                                                   # identical image -> 1
    x[1, 40:60, 10:30] = 0.3                                                                     # a flat patch
    ref = O.ssim_G_x(G, x)
    got = U.ssim_G_x(torch.from_numpy(G).to(dev), torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.abs(got - ref).max() < 2e-4, (got, ref)
    assert abs(got[0] - 1.0) < 1e-5
    small = rng.uniform(0, 255, (2, 7, 9, 3)).astype(np.float32)                                 # minimum size: 1x3 windows
    ref2 = O.ssim_G_x(small, small[::-1] / 127.5 - 1)
    got2 = U.ssim_G_x(torch.from_numpy(small).to(dev), torch.from_numpy(small[::-1].copy() / 127.5 - 1).to(dev)).cpu().numpy()
    assert np.abs(got2 - ref2).max() < 2e-4


def test_gp_penalty_fused(dev):
    """trainer.py:222-236: interpolation, and LAMBDA*mean((||g||-1)^2) with its derivative w.r.t. g in one pass."""
    import dpig_amd.hip_ops as H
    B, shape = 5, (5, 8, 4, 3)
    real, fake = _rand(shape, 1), _rand(shape, 2)
    alpha = _rand((B,), 3, 0.0, 1.0)
    xh = H.gp_interpolate(real.float().to(dev), fake.float().to(dev), alpha.float().to(dev))
    _close(xh, real + alpha.reshape(B, 1, 1, 1) * (fake - real))
    g = (_rand(shape, 4) * 0.3).requires_grad_(True)
    slopes = torch.sqrt((g * g).reshape(B, -1).sum(1))
    pen = 10.0 * ((slopes - 1.0) ** 2).mean()
    pen.backward()
    p, dg, sl = H.gp_penalty(g.detach().float().to(dev), 10.0)
    _close(p, pen.reshape(1)); _close(dg, g.grad); _close(sl, slopes)
    z = torch.zeros(2, 7, device=dev)                       # zero slope: no NaN
    p0, dg0, _ = H.gp_penalty(z, 10.0)
    assert float(p0) == 10.0 and float(dg0.abs().max()) == 0.0


def test_glue_kernels_equal_the_elementwise_graph_they_replace(dev):
    """csrc/dpig_glue.hip: mask split (models.py:402-403), ROI box normalisation (:405-413), visibility multiply + concat
    (:433-442, 467-468), the tiled-embedding class sums and their transpose, the critic's NCHW flatten
    (wgan_gp.py:433) -- each against the plain tensor expression of the reference graph, forward and gradient."""
    import dpig_amd.hip_ops as H
    import dpig_amd.autograd as A
    g = torch.Generator().manual_seed(5)
    # mask split, fp32 and bf16, odd channel count (scalar path) and vector path
    for C, dt in ((128, torch.float32), (24, torch.bfloat16), (5, torch.float32)):
        x = torch.randn((2, 6, 5, C), generator=g).to(dt).to(dev).requires_grad_(True)
        m = (torch.rand((2, 6, 5, 1), generator=g) < 0.4).float().to(dev)
        fg, bg = A.mask_split(x, m)
        mm = m.to(dt)
        assert torch.equal(fg, x.detach() * mm) and torch.equal(bg, x.detach() * (1.0 - mm))
        dfg, dbg = torch.randn(fg.shape, generator=g).to(dt).to(dev), torch.randn(bg.shape, generator=g).to(dt).to(dev)
        (dx,) = torch.autograd.grad([fg, bg], x, [dfg, dbg])
        ref = (dfg.float() * m + dbg.float() * (1.0 - m)).to(dt)
        assert torch.equal(dx, ref)
        fg2, _ = A.mask_split(x, m)
        (dx1,) = torch.autograd.grad(fg2, x, dfg)            # only one branch carries a gradient
        assert torch.equal(dx1, (dfg.float() * m).to(dt))
    # ROI boxes
    bbox = torch.randint(0, 60, (3, 7, 4), generator=g, dtype=torch.int32).to(dev)
    boxes, ind = H.roi_boxes(bbox, 5, 128.0, 64.0)
    b = bbox[:, :5, :].float()
    ref = torch.stack([b[..., 0] / 128.0, b[..., 1] / 64.0, b[..., 2] / 128.0, b[..., 3] / 64.0], dim=-1).permute(1, 0, 2).reshape(15, 4)
    assert torch.equal(boxes, ref) and torch.equal(ind.cpu(), torch.arange(3, dtype=torch.int32).repeat(5))
    assert torch.equal(H.roi_boxes(bbox.float(), 5, 128.0, 64.0)[0], ref)
    # visibility multiply + concat
    B, P, z = 3, 7, 32
    fea = torch.randn((P * B, z), generator=g).to(dev).requires_grad_(True)
    bgf = torch.randn((B, 4 * z), generator=g).to(dev).requires_grad_(True)
    vis = (torch.rand((B, P), generator=g) < 0.7).float().to(dev)
    out = A.vis_concat(fea, vis, bgf, P, z)
    ref = torch.cat([fea.reshape(P, B, z)[p] * vis[:, p:p + 1] for p in range(P)] + [bgf], dim=-1)
    assert torch.equal(out, ref)
    dall = torch.randn(out.shape, generator=g).to(dev)
    d1 = torch.autograd.grad(out, [fea, bgf], dall)
    d2 = torch.autograd.grad(ref, [fea, bgf], dall)
    assert torch.equal(d1[0], d2[0]) and torch.equal(d1[1], d2[1])
    assert torch.equal(A.vis_concat(fea, vis, None, P, z), ref[:, :P * z])
    # class sums of the filter taps and their transpose
    E, Pc, K = 40, 18, 24
    w = torch.randn((3, 3, E + Pc, K), generator=g).to(dev)
    wmat = H.emb_class_weights_fwd(w, E)
    vt = (slice(1, 3), slice(0, 3), slice(0, 2))
    we = w[:, :, :E, :]
    wy = torch.stack([we[vt[c]].sum(0) for c in range(3)])
    wc = torch.stack([wy[:, vt[c]].sum(1) for c in range(3)], dim=1)
    ref = wc.permute(2, 0, 1, 3).reshape(E, 9 * K)
    assert (wmat - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()
    dwc = torch.randn((E, 9 * K), generator=g).to(dev)
    outw = torch.full((3, 3, E + Pc, K), 2.0, device=dev)
    H.emb_class_weights_bwd(dwc, E, outw, 1.0)
    d = dwc.view(E, 3, 3, K).permute(1, 2, 0, 3)
    dwy = torch.stack([d[1:].sum(0), d.sum(0), d[:2].sum(0)])
    dwe = torch.stack([dwy[:, 1:].sum(1), dwy.sum(1), dwy[:, :2].sum(1)], dim=1)
    assert (outw[:, :, :E, :] - (dwe + 2.0)).abs().max().item() <= 1e-5 and bool((outw[:, :, E:, :] == 2.0).all())
    dwp = torch.randn((3, 3, 32, K), generator=g).to(dev)
    H.axpby3d(dwp.view(9, 32, K)[:, :Pc, :], outw.view(9, E + Pc, K)[:, E:, :], 0.0)
    assert torch.equal(outw[:, :, E:, :], dwp[:, :, :Pc, :])
    # NCHW flatten of an NHWC tensor
    for dt in (torch.float32, torch.bfloat16):
        xn = torch.randn((4, 8, 4, 96), generator=g).to(dt).to(dev).requires_grad_(True)
        y = A.nchw_flatten(xn.permute(0, 3, 1, 2), 8 * 4 * 96 // 2)
        ref = xn.permute(0, 3, 1, 2).reshape(-1, 8 * 4 * 96 // 2)
        assert torch.equal(y, ref)
        dy = torch.randn(y.shape, generator=g).to(dt).to(dev)
        assert torch.equal(torch.autograd.grad(y, xn, dy)[0], torch.autograd.grad(ref, xn, dy)[0])


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_act_bwd_pool2x_equals_masked_gradient_summed_over_2x2(dev, dtype):
    """dpig_act_bwd_pool2x: the gradient of act(conv1x1(upsample2x(x))) on x's grid = 2x2 block sums of dy * act'(y)
    (ReluGrad + ResizeNearestNeighborGrad of models.py:569-570 / utils.py:61-72 in one pass), dense and channel-sliced operands."""
    import dpig_amd.hip_ops as H
    g = torch.Generator().manual_seed(3)
    N, Hh, W, C = 3, 6, 10, 24
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    wide = torch.randn(N, 2 * Hh, 2 * W, C + 8, generator=g).to(td).to(dev)
    dy_dense = torch.randn(N, 2 * Hh, 2 * W, C, generator=g).to(td).to(dev)
    y = torch.randn(N, 2 * Hh, 2 * W, C, generator=g).to(td).to(dev)
    for dy in (dy_dense, wide[..., 8:]):                      # dense, and a channel slice of a wider buffer (decoder concat)
        for act, alpha in ((0, 0.2), (1, 0.2), (2, 0.2)):
            ref = dy.float()
            if act == 1:
                ref = ref * (y.float() > 0).float()
            elif act == 2:
                ref = ref * torch.where(y.float() > 0, torch.ones_like(ref), torch.full_like(ref, alpha))
            ref = ref.reshape(N, Hh, 2, W, 2, C).sum(dim=(2, 4))
            got = H.act_bwd_pool2x(dy, y, act, alpha)
            assert got.dtype == td and got.shape == (N, Hh, W, C)
            tol = 2.0 ** -7 if dtype == "bf16" else 1e-6
            assert float((got.float() - ref).abs().max()) <= tol * float(ref.abs().max())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_keypoint_fed_stem_conv_equals_the_dense_conv_on_the_rasterised_map(dev, dtype, K=32):
    """autograd._TiledEmbKeypointConvFn (dpig_pose_stem_fwd / _wgrad): relu(conv3x3_SAME(concat([tile(emb), pose_map]), w) + b) and
    its gradients w.r.t. emb, w, b from the KEYPOINTS (trainer.py:556-560 builds pose_map from pose_rcv in the graph; utils.py:237-318),
    against the fp64 oracle's dense conv on the map `oracle.ops` rasterises from the same keypoints.  Keypoints in the corners, on the
    borders, invisible, and two images with the same keypoint."""
    import dpig_amd.hip_ops as H
    from dpig_amd import autograd as A
    from oracle import ops as O
    g = torch.Generator().manual_seed(21)
    B, Hh, W, E, P = 3, 24, 16, 20, 18
    r = torch.randint(0, Hh, (B, P), generator=g).double()
    c = torch.randint(0, W, (B, P), generator=g).double()
    v = (torch.rand(B, P, generator=g) < 0.8).double()
    r[0, :4] = torch.tensor([0, 0, Hh - 1, Hh - 1]).double(); c[0, :4] = torch.tensor([0, W - 1, 0, W - 1]).double(); v[0, :4] = 1
    r[1, :2] = torch.tensor([2.0, Hh - 3.0]); c[1, :2] = torch.tensor([0.0, W - 1.0]); v[1, :2] = 1
    rcv = torch.stack([r, c, v], -1)
    emb = (torch.randn(B, E, generator=g, dtype=torch.float64) * 0.5).requires_grad_(True)
    w = (torch.randn(3, 3, E + P, K, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    b = (torch.randn(K, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    pose_map = O.tf_poseInflate(O.coord2channel_simple_rcv(rcv.reshape(B, -1), P, False, Hh, W), P, 4, Hh, W)
    x = torch.cat([emb.reshape(B, 1, 1, E).expand(B, Hh, W, E), pose_map], -1)
    y_ref = O.relu(O.conv2d_same(x, w, b, 1))
    dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
    d_emb, d_w, d_b = torch.autograd.grad((y_ref * dy).sum(), [emb, w, b])
    H.set_compute(dtype)
    try:
        eg = emb.detach().float().to(dev).requires_grad_(True)
        wg = w.detach().float().to(dev).requires_grad_(True)
        bg = b.detach().float().to(dev).requires_grad_(True)
        kp = A.PoseKeypoints(rcv.reshape(B, -1).float().to(dev), Hh, W, P, is_normalized=False)
        y = A.tiled_emb_conv(eg, kp, wg, bg)
        assert y.dtype == (torch.bfloat16 if dtype == "bf16" else torch.float32)
        tol = 2.0 ** -7 if dtype == "bf16" else 2e-5
        assert float((y.float().cpu().double() - y_ref).abs().max()) <= tol * float(y_ref.abs().max())
        assert torch.equal(kp.dense().cpu().double(), pose_map)                      # the stand-in rasterises the same map on demand
        dyg = dy.float().to(dev)
        ge, gw, gb = torch.autograd.grad((y.float() * dyg).sum(), [eg, wg, bg])
        gtol = 2e-2 if dtype == "bf16" else 1e-4
        for got, ref, name in ((ge, d_emb, "emb"), (gw, d_w, "w"), (gb, d_b, "b")):
            assert float((got.cpu().double() - ref).abs().max()) <= gtol * float(ref.abs().max()), name
        # ... and the pose rows of the filter gradient on their own (the part the sparse gather produces)
        assert float((gw[:, :, E:, :].cpu().double() - d_w[:, :, E:, :]).abs().max()) <= gtol * float(d_w[:, :, E:, :].abs().max())
    finally:
        H.set_compute("f32")


def test_batchnorm_training_flag_updates_the_moving_statistics(dev):
    """Batchnorm(..., is_training=True, stats_iter=k) (tflib/ops/batchnorm.py:51-68; off the hot path, where is_training is None): the output
    is the training-mode normalisation and the moving statistics become the 1/(k+1) running averages of the batch mean and of the
    Bessel-corrected batch variance (what tf.nn.fused_batch_norm returns); update_moving_stats=False and is_training=None leave them alone."""
    import dpig_amd.tflib as lib
    import dpig_amd.tflib.ops  # noqa
    from oracle import ops as O
    lib.delete_all_params()
    lib.set_device(dev)
    x = _rand((4, 9, 7, 12), 3)                                       # NHWC
    xg = x.float().to(dev).permute(0, 3, 1, 2)
    mm0, mv0 = _rand((12,), 4), _rand((12,), 5, 0.5, 1.5)
    lib.param('M.moving_mean', mm0.float().numpy(), trainable=False)
    lib.param('M.moving_variance', mv0.float().numpy(), trainable=False)
    y = lib.ops.batchnorm.Batchnorm('M', [0, 2, 3], xg, is_training=True, stats_iter=3)
    one, zero = torch.ones(12, dtype=torch.float64), torch.zeros(12, dtype=torch.float64)
    _close(y.permute(0, 2, 3, 1), O.batchnorm_train(x, one, zero), 1e-4)
    rm, rv = O.batchnorm_moving_update(x, mm0, mv0, 3)
    _close(lib.param('M.moving_mean'), rm, 1e-5)
    _close(lib.param('M.moving_variance'), rv, 1e-5)
    before = (lib.param('M.moving_mean').clone(), lib.param('M.moving_variance').clone())
    lib.ops.batchnorm.Batchnorm('M', [0, 2, 3], xg, is_training=True, stats_iter=4, update_moving_stats=False)
    lib.ops.batchnorm.Batchnorm('M', [0, 2, 3], xg)                   # the hot path's call: statistics created, never updated
    assert torch.equal(lib.param('M.moving_mean'), before[0]) and torch.equal(lib.param('M.moving_variance'), before[1])
    lib.delete_all_params()

