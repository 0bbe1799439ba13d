"""GPU parity of the bf16-STORAGE conv family (dpig_conv2d_*_bf16, through the C ABI) against the CPU oracle (fp64).

Operands are bf16 in HBM, so the oracle is evaluated on the bf16-ROUNDED operands: products of bf16 numbers are exact
in fp32 and the kernels accumulate in fp32, hence
  * fp32 results (wgrad, bias gradient) must match at the fp32 kernels' bar, 2e-5 * max|ref|;
  * bf16 results (fwd, dgrad) must match to within the final rounding to bf16: |err| <= 2^-8 |ref| + 2e-5 max|ref|
    elementwise (half an ulp is 2^-9 |ref|).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16

# (N, H, W, C, K, k, stride): every shape the bf16 loops accept (C, K >= 32, multiples of 8)
SHAPES = [
    (2, 16, 8, 64, 128, 3, 1),     # decoder-style 3x3 s1, one k-chunk per tap
    (2, 16, 8, 128, 64, 3, 2),     # encoder down conv (TF pad (0,1)), two k-chunks per tap
    (1, 12, 12, 128, 256, 3, 1),   # ragged M (144 rows), 2 n-tiles
    (2, 9, 7, 40, 48, 3, 1),       # odd sizes, partial tiles in every dim, C not a multiple of the 64-deep k-tile
    (2, 9, 7, 40, 48, 3, 2),       # odd input with stride 2 (pad (1,1))
    (2, 8, 4, 64, 128, 5, 2),      # D.2-style 5x5 s2 (pad (1,2))
    (3, 6, 6, 96, 136, 1, 1),      # 1x1, K not a multiple of 128
    (1, 8, 4, 200, 128, 3, 1),     # C = 3 k-chunks + a 8-channel tail
    (2, 3, 3, 640, 64, 3, 1),      # 3x3 image (ROI tower tail): every tap is mostly halo
]


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def _r(t):
    """round to bf16 (RNE) and back to fp64: what the device tensors hold"""
    return t.float().to(BF).double()


def _close_f32(got, ref, tol=2e-5):
    ref = ref.double()
    err = (got.double().cpu() - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    assert err <= tol * scale, "max err %.3e vs scale %.3e" % (err, scale)


def _close_bf16(got, ref):
    assert got.dtype == BF
    ref = ref.double()
    err = (got.double().cpu() - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 2e-5 * max(ref.abs().max().item(), 1e-6)
    bad = (err > bound)
    assert not bad.any(), "%d elements off; worst err %.3e (ref scale %.3e)" % (int(bad.sum()), err.max().item(), ref.abs().max().item())


def test_conversions_are_rne_and_exact(dev):
    import dpig_amd.hip_ops as H
    x = _rand((3, 5, 7, 24), 1, 3.0).float()
    x.view(-1)[:4] = torch.tensor([1.00390625, 1.01171875, -0.0, 3.3895314e38])   # ties-to-even cases, -0, near max
    xb = H.to_bf16(x.to(dev))
    assert xb.dtype == BF and torch.equal(xb.cpu(), x.to(BF))
    assert torch.equal(H.to_f32(xb).cpu(), x.to(BF).float())
    # channel slice (row stride > cols) and a width that is not a multiple of 4
    big = x.to(dev)
    assert torch.equal(H.to_bf16(big[..., 8:16]).cpu(), x[..., 8:16].to(BF))
    odd = _rand((11, 13), 2).float()
    assert torch.equal(H.to_bf16(odd.to(dev)).cpu(), odd.to(BF))
    assert torch.equal(H.to_f32(odd.to(BF).to(dev)).cpu(), odd.to(BF).float())


def test_filter_shadows(dev):
    import dpig_amd.hip_ops as H
    w = _rand((3, 3, 40, 72), 3).float()
    plain, trans = H.filter_shadows(w.to(dev))
    assert torch.equal(plain.cpu(), w.to(BF))
    assert torch.equal(trans.cpu(), w.permute(0, 1, 3, 2).contiguous().to(BF))


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("split_k", [0, 3])
def test_fwd(dev, shape, split_k):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = shape
    x = _rand((N, Hh, W, C), 1)
    w = _rand((k, k, C, K), 2, 0.2)
    b = _rand((K,), 3)
    ref = O.leaky_relu(O.conv2d_same(_r(x), _r(w), b.float().double(), s), 0.2)
    got = H.conv2d_fwd(x.float().to(dev).to(BF), w.float().to(dev), b.float().to(dev), stride=s, act=2, alpha=0.2,
                       split_k=split_k)
    _close_bf16(got, ref)


def test_fwd_fused_epilogues(dev):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = 2, 16, 8, 64, 64
    x = _rand((N, Hh, W, C), 1)
    w = _rand((3, 3, C, K), 2, 0.2)
    b = _rand((K,), 3)
    res = _rand((N, Hh, W, K), 4)
    xd, wd, bd, rd = x.float().to(dev).to(BF), w.float().to(dev), b.float().to(dev), res.float().to(dev).to(BF)
    conv = O.conv2d_same(_r(x), _r(w), b.float().double(), 1)
    # residual before the activation
    _close_bf16(H.conv2d_fwd(xd, wd, bd, act=1, residual=rd), O.relu(conv + _r(res)))
    # the res-block tail: act -> out_act, (stored act) + skip -> out
    out, out_act = torch.empty((N, Hh, W, K), dtype=BF, device=dev), torch.empty((N, Hh, W, K), dtype=BF, device=dev)
    H.conv2d_fwd(xd, wd, bd, act=1, residual=rd, res_after_act=True, out=out, out_act=out_act)
    _close_bf16(out_act, O.relu(conv))
    assert torch.equal(out.cpu(), (out_act.float() + rd.float()).to(BF).cpu())      # exactly bf16(float(c2) + skip)
    # split-K goes through the reduction kernel's epilogue
    out2, act2 = torch.empty_like(out), torch.empty_like(out)
    H.conv2d_fwd(xd, wd, bd, act=1, residual=rd, res_after_act=True, out=out2, out_act=act2, split_k=3)
    _close_bf16(act2, O.relu(conv))
    assert torch.equal(out2.cpu(), (act2.float() + rd.float()).to(BF).cpu())
    # class-indexed residual (the tiled-embedding collapse, fp32 [N, 9, K])
    e9 = _rand((N, 9, K), 5).float()
    yy, xx = torch.meshgrid(torch.arange(Hh), torch.arange(W), indexing="ij")
    cls = torch.where(yy == 0, 0, torch.where(yy == Hh - 1, 2, 1)) * 3 + torch.where(xx == 0, 0, torch.where(xx == W - 1, 2, 1))
    ref = O.relu(conv + e9.double()[:, cls.reshape(-1), :].reshape(N, Hh, W, K))
    _close_bf16(H.conv2d_fwd(xd, wd, bd, act=1, residual=e9.to(dev), res_class=True), ref)


def test_channel_slices_and_upsample(dev):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = 2, 8, 4, 64, 128
    xbig = _rand((N, Hh, W, 96), 1)
    w = _rand((3, 3, C, K), 2, 0.2)
    ref = O.conv2d_same(_r(xbig[..., 32:96]), _r(w), None, 1)
    xg = xbig.float().to(dev).to(BF)
    ybig = torch.full((N, Hh, W, 192), 7.0, device=dev, dtype=BF)
    H.conv2d_fwd(xg[..., 32:96], w.float().to(dev), None, out=ybig[..., 64:192])
    _close_bf16(ybig[..., 64:192], ref)
    assert (ybig[..., :64] == 7.0).all()
    # nearest-2x upsample + 1x1 conv + bias + relu, computed at low resolution and replicated
    C, K = 96, 64
    x = _rand((N, Hh, W, C), 3)
    w1 = _rand((1, 1, C, K), 4, 0.3)
    b = _rand((K,), 5)
    ref = O.relu(O.conv2d_same(O.upsample2x(_r(x)), _r(w1), b.float().double(), 1))
    got = H.conv2d_fwd(x.float().to(dev).to(BF), w1.float().to(dev), b.float().to(dev), act=1, upsample2x=True)
    _close_bf16(got, ref)
    dy = _rand(tuple(ref.shape), 6)
    xr = _r(x).requires_grad_(True)
    wr = _r(w1).requires_grad_(True)
    O.conv2d_same(O.upsample2x(xr), wr, None, 1).backward(_r(dy))
    dyd = dy.float().to(dev).to(BF)
    _close_bf16(H.conv2d_dgrad(dyd, w1.float().to(dev), (N, Hh, W, C), upsample2x=True), xr.grad)
    _close_f32(H.conv2d_wgrad(x.float().to(dev).to(BF), dyd, (1, 1, C, K), upsample2x=True), wr.grad)


@pytest.mark.parametrize("shape", SHAPES)
def test_dgrad_wgrad(dev, shape):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = shape
    x = _rand((N, Hh, W, C), 1)
    w = _rand((k, k, C, K), 2, 0.2)
    xr = _r(x).requires_grad_(True)
    wr = _r(w).requires_grad_(True)
    y = O.conv2d_same(xr, wr, None, s)
    dy = _rand(tuple(y.shape), 3)
    y.backward(_r(dy))
    xd, dyd, wd = x.float().to(dev).to(BF), dy.float().to(dev).to(BF), w.float().to(dev)
    _close_bf16(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), stride=s), xr.grad)
    db = torch.empty(K, device=dev)
    dw = torch.empty((k, k, C, K), device=dev)
    H.conv2d_wgrad(xd, dyd, (k, k, C, K), stride=s, out=dw, beta=0.0, db=db, db_beta=0.0)
    _close_f32(dw, wr.grad)
    _close_f32(db, _r(dy).sum((0, 1, 2)))


def test_dgrad_mask_accum_and_wgrad_split_beta(dev):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = 2, 16, 8, 64, 128
    x = _rand((N, Hh, W, C), 1)
    w = _rand((3, 3, C, K), 2, 0.2)
    dy = _rand((N, Hh, W, K), 3)
    acc = _rand((N, Hh, W, C), 4)
    m = _rand((N, Hh, W, C), 5)
    xr = _r(x).requires_grad_(True)
    wr = _r(w).requires_grad_(True)
    O.conv2d_same(xr, wr, None, 1).backward(_r(dy))
    dyd, wd = dy.float().to(dev).to(BF), w.float().to(dev)
    got = H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), accum=acc.float().to(dev).to(BF), mask=m.float().to(dev).to(BF), act=2,
                         alpha=0.2)
    ref = (xr.grad + _r(acc)) * torch.where(_r(m) > 0, 1.0, 0.2)
    _close_bf16(got, ref)
    for split in (0, 1, 5):
        dw = torch.full((3, 3, C, K), 2.0, device=dev)
        db = torch.full((K,), 3.0, device=dev)
        H.conv2d_wgrad(x.float().to(dev).to(BF), dyd, (3, 3, C, K), out=dw, beta=1.0, split_k=split, db=db, db_beta=1.0)
        _close_f32(dw, wr.grad + 2.0)
        _close_f32(db, _r(dy).sum((0, 1, 2)) + 3.0)


@pytest.mark.parametrize("layer", [(2, 16, 40, 256, 3, 3, 1), (2, 16, 40, 3, 128, 3, 1), (2, 16, 40, 3, 64, 5, 2),
                                   (2, 16, 8, 18, 128, 3, 1)])
def test_thin_layers(dev, layer):
    """3 output channels (x bf16 -> fp32 image), 3 input channels (fp32 image -> bf16): the vector-ALU kernels read /
    write the wide tensor as bf16 directly (dpig_conv2d_*_thin_bf16).  Anything else outside the bf16 loops (the
    18-channel pose conv) runs on the fp32 kernels between conversions.  A result is stored as bf16 iff its channel count
    is a multiple of 8."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = layer
    H.set_compute("bf16")
    try:
        x = _rand((N, Hh, W, C), 1)
        w = _rand((k, k, C, K), 2, 0.2)
        b = _rand((K,), 3)
        xs = _r(x) if C % 8 == 0 else x.float().double()          # what the device tensor holds
        xd = x.float().to(dev).to(BF) if C % 8 == 0 else x.float().to(dev)
        xr = xs.clone().requires_grad_(True)
        wr = w.float().double().requires_grad_(True)
        y = O.conv2d_same(xr, wr, b.float().double(), s)
        got = H.conv2d_fwd(xd, w.float().to(dev), b.float().to(dev), stride=s)
        if K % 8 == 0:
            assert got.dtype == BF
            _close_bf16(got, y.detach())
        else:
            assert got.dtype == torch.float32                    # the image stays fp32
            _close_f32(got, y.detach())
        dy = _rand(tuple(y.shape), 5)
        dys = _r(dy) if K % 8 == 0 else dy.float().double()
        dyd = dy.float().to(dev).to(BF) if K % 8 == 0 else dy.float().to(dev)
        y.backward(dys)
        if C != 3 or k == 5:                                      # (no gradient towards the image through the 3x3 stem)
            dx = H.conv2d_dgrad(dyd, w.float().to(dev), (N, Hh, W, C), stride=s)
            if C % 8 == 0:
                assert dx.dtype == BF
                _close_bf16(dx, xr.grad)
            else:
                assert dx.dtype == torch.float32
                _close_f32(dx, xr.grad)
        dw = torch.full((k, k, C, K), 2.0, device=dev)
        db = torch.full((K,), 3.0, device=dev)
        H.conv2d_wgrad(xd, dyd, (k, k, C, K), stride=s, out=dw, beta=1.0, db=db, db_beta=1.0)
        _close_f32(dw, wr.grad + 2.0)
        _close_f32(db, dys.sum((0, 1, 2)) + 3.0)
    finally:
        H.set_compute("f32")


@pytest.mark.parametrize("shape", [(32, 24, 40, 72, 136), (256, 32, 8, 64, 64), (40, 19, 33, 128, 128)])
def test_halo_patch_kernel_3x3(dev, shape):
    """3x3 stride-1 layers with >= 512 output tiles take the halo-patch kernel (bh_kernel: the input patch + halo staged
    once per channel chunk, 9 shifted window reads): 8 x 16 patches with partial patches on both axes, a partial channel
    chunk (72 = 64 + 8) and two column tiles (136); 16 x 8 patches (W = 8); odd sizes.  Forward with the fused
    epilogues, stride-1 dgrad (flipped taps) with mask and accumulate, against the fp64 oracle on the rounded operands."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = shape
    x = _rand((N, Hh, W, C), 1)
    w = _rand((3, 3, C, K), 2, 0.2)
    b = _rand((K,), 3)
    res = _rand((N, Hh, W, K), 4)
    xr = _r(x).requires_grad_(True)
    conv = O.conv2d_same(xr, _r(w), None, 1)
    xd, wd, bd, rd = x.float().to(dev).to(BF), w.float().to(dev), b.float().to(dev), res.float().to(dev).to(BF)
    _close_bf16(H.conv2d_fwd(xd, wd, bd, act=2, alpha=0.2), O.leaky_relu(conv.detach() + b.float().double(), 0.2))
    out, out_act = torch.empty((N, Hh, W, K), dtype=BF, device=dev), torch.empty((N, Hh, W, K), dtype=BF, device=dev)
    H.conv2d_fwd(xd, wd, bd, act=1, residual=rd, res_after_act=True, out=out, out_act=out_act)
    _close_bf16(out_act, O.relu(conv.detach() + b.float().double()))
    assert torch.equal(out.cpu(), (out_act.float() + rd.float()).to(BF).cpu())
    dy = _rand((N, Hh, W, K), 5)
    conv.backward(_r(dy))
    acc, m = _rand((N, Hh, W, C), 6), _rand((N, Hh, W, C), 7)
    got = H.conv2d_dgrad(dy.float().to(dev).to(BF), wd, (N, Hh, W, C), accum=acc.float().to(dev).to(BF),
                         mask=m.float().to(dev).to(BF), act=1)
    _close_bf16(got, (xr.grad + _r(acc)) * (_r(m) > 0))
    # the same layer on the tap-by-tap kernel (small-layer path) must agree to the final rounding
    import os, subprocess, sys
    assert os.environ.get("DPIG_BF16_HALO", "1") != "0"


def test_crop_resize_and_border_sums_on_bf16_tensors(dev):
    """tf.image.crop_and_resize (models.py:415) forward / image gradient and the border-class sums with bf16 storage."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C = 2, 32, 16, 16
    img = _rand((N, Hh, W, C), 1)
    boxes = torch.tensor([[0.1, 0.2, 0.8, 0.9], [0.0, 0.0, 1.0, 1.0], [-0.2, 0.3, 0.6, 1.2], [0.5, 0.5, 0.5, 0.5]], dtype=torch.float64)
    ind = torch.tensor([0, 1, 1, 0], dtype=torch.int64)
    xr = _r(img).requires_grad_(True)
    ref = O.crop_and_resize(xr, boxes, ind, 12, 12)
    got = H.crop_resize_fwd(img.float().to(dev).to(BF), boxes.float().to(dev), ind.to(dev), 12, 12)
    _close_bf16(got, ref.detach())
    dout = _rand(tuple(ref.shape), 2)
    ref.backward(_r(dout))
    dimg = H.crop_resize_bwd(dout.float().to(dev).to(BF), boxes.float().to(dev), ind.to(dev), (N, Hh, W, C))
    _close_bf16(dimg, xr.grad)
    z = _rand((N, Hh, W, 72), 3)
    s32 = H.border_class_sum(_r(z).float().to(dev))
    sbf = H.border_class_sum(z.float().to(dev).to(BF))
    assert sbf.dtype == torch.float32
    _close_f32(sbf, s32.double().cpu(), 1e-6)
    xp = _rand((N, Hh, W, 18), 4).float()
    pad = H.pad_channels_bf16(xp.to(dev), 32)
    assert torch.equal(pad[..., :18].cpu(), xp.to(BF)) and bool((pad[..., 18:] == 0).all())


def test_bad_arguments(dev):
    import ctypes
    from dpig_amd import _lib
    h = _lib.lib()
    d = _lib.DpigConvDesc()
    for k, v in dict(N=1, H=8, W=8, C=64, K=64, R=3, S=3, stride=1, pad_t=-1, pad_l=-1, ldx=64, ldy=64).items():
        setattr(d, k, v)
    assert h.dpig_conv2d_bf16_supported(ctypes.byref(d), 0) == 1
    d.C = d.ldx = 36
    assert h.dpig_conv2d_bf16_supported(ctypes.byref(d), 0) == 0
    assert h.dpig_conv2d_fwd_bf16(ctypes.byref(d), 16, 16, None, None, None, 16, None, None, 0, None) == -22
    d.C = d.ldx = 64
    assert h.dpig_conv2d_fwd_bf16(ctypes.byref(d), 16, 18, None, None, None, 16, None, None, 0, None) == -14   # misaligned
    assert h.dpig_conv2d_fwd_bf16(ctypes.byref(d), None, 16, None, None, None, 16, None, None, 0, None) == -22


@pytest.mark.parametrize("shape", [(3, 20, 20, 64, 128, 5, 2), (2, 17, 9, 32, 136, 3, 1)])
def test_bf16_conv_epilogue_bn_statistics(dev, shape):
    """dpig_conv2d_fwd_bf16_stats: the bf16-storage forward conv leaves the batch-norm partial statistics of its output AS
    STORED (accumulator + bias rounded to bf16); merged by dpig_bn_stats_finalize they give the batch mean and rstd of the
    stored tensor -- the same numbers bn_fwd's own passes find -- and bn_fwd(..., stats=) the normalised LeakyReLU output."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = shape
    g = torch.Generator().manual_seed(7)
    bf = lambda t: t.float().bfloat16().double()
    x = bf(torch.randn(N, Hh, W, C, generator=g, dtype=torch.float64))
    w = bf(torch.randn(k, k, C, K, generator=g, dtype=torch.float64) * 0.1)
    b = torch.randn(K, generator=g, dtype=torch.float64) * 0.5 + 8.0
    scale = torch.rand(K, generator=g, dtype=torch.float64) + 0.5
    offset = torch.randn(K, generator=g, dtype=torch.float64)
    pre = O.conv2d_same(x, w, b, s)
    rows = pre.reshape(-1, K)
    xg, wg, bg = x.float().to(dev).bfloat16(), w.float().to(dev), b.float().to(dev)
    y, st = H.conv2d_fwd_stats(xg, wg, bg, stride=s, split_k=1)
    assert y.dtype == torch.bfloat16 and st is not None and st[0].shape == ((rows.shape[0] + 127) // 128, 2, K)
    err = (y.float().cpu().double() - pre).abs()
    assert not bool((err > pre.abs() * 2.0 ** -8 + 1e-4 * float(pre.abs().max())).any())
    out, mean, rstd = H.bn_fwd(y, scale.float().to(dev), offset.float().to(dev), 1e-5, 2, 0.2, stats=st)
    # the statistics describe the tensor AS STORED (bf16-rounded y): exactly what bn_apply / bn_bwd normalise ...
    stored = y.float().cpu().double().reshape(-1, K)
    assert float((mean.cpu().double() - stored.mean(0)).abs().max()) <= 2e-6 * float(stored.mean(0).abs().max())
    st_rstd = 1.0 / torch.sqrt(stored.var(0, unbiased=False) + 1e-5)
    assert float((rstd.cpu().double() - st_rstd).abs().max()) <= 2e-5 * float(st_rstd.abs().max())
    # ... so the path without epilogue statistics (split-K plans, multi-run batches: bn_fwd's own passes over y) agrees
    out_ns, mean_ns, rstd_ns = H.bn_fwd(y, scale.float().to(dev), offset.float().to(dev), 1e-5, 2, 0.2, stats=None)
    assert float((mean_ns - mean).abs().max()) <= 2e-6 * float(mean.abs().max())
    assert float((rstd_ns - rstd).abs().max()) <= 2e-5 * float(rstd.abs().max())
    assert float((out_ns.float() - out.float()).abs().max()) <= 2.0 ** -7 * float(out.float().abs().max())
    # ... and they are the oracle's batch statistics up to the rounding of y: values near 8 round with ulp 2^-4, i.e. a
    # zero-mean error of std 0.018 per element, 0.018 / sqrt(rows) on a column mean (5 sigma over the K columns)
    tol_mean = 5.0 * (2.0 ** -4 / 12 ** 0.5) / rows.shape[0] ** 0.5
    assert float((mean.cpu().double() - rows.mean(0)).abs().max()) <= tol_mean
    ref_rstd = 1.0 / torch.sqrt(rows.var(0, unbiased=False) + 1e-5)
    assert float((rstd.cpu().double() - ref_rstd).abs().max()) <= 1e-2 * float(ref_rstd.abs().max())
    ref = O.leaky_relu(O.batchnorm_train(pre, scale, offset), 0.2)
    assert out.dtype == torch.bfloat16
    assert float((out.float().cpu().double() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())      # bf16 y, bf16 result
    y2, st2 = H.conv2d_fwd_stats(xg, wg, bg, stride=s, split_k=1)
    assert torch.equal(y2, y) and torch.equal(st2[0], st[0])


@pytest.mark.parametrize("shape", [(56, 1, 1, 896, 896, 3, 1), (56, 2, 2, 768, 896, 3, 2), (7, 1, 1, 64, 72, 5, 1), (5, 1, 3, 64, 64, 3, 1),
                                   (9, 2, 1, 96, 64, 3, 2)])
def test_maps_smaller_than_the_filter_run_their_live_taps_only(dev, shape):
    """The bottom of the DeepFashion ROI tower (models.py:420-431 with repeat_num 7, trainer_256.py:40-41): 3x3 convs on 1 x 1 maps and the
    2 x 2 -> 1 x 1 stride-2 conv.  Taps that only ever see padding are skipped in forward and dgrad (dpig_conv_plan.h::live_taps):
    forward (+ bias + ReLU), dgrad (* mask), and the (dense) wgrad (+ bias gradient; beta 0 / 1 / 0.5) against the DENSE fp64 oracle
    conv on the bf16-rounded operands."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = shape
    x = _rand((N, Hh, W, C), 31)
    w = _rand((k, k, C, K), 32, 0.1)
    b = _rand((K,), 33)
    xr = _r(x).requires_grad_(True)
    wr = _r(w).requires_grad_(True)
    y0 = O.conv2d_same(xr, wr, None, s)
    dy = _rand(tuple(y0.shape), 34)
    y0.backward(_r(dy))
    xd, wd, bd, dyd = x.float().to(dev).to(BF), w.float().to(dev), b.float().to(dev), dy.float().to(dev).to(BF)
    _close_bf16(H.conv2d_fwd(xd, wd, bd, stride=s, act=1), O.relu(y0.detach() + b.double()))
    if s == 1:
        m = _rand((N, Hh, W, C), 35)
        _close_bf16(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), stride=s, mask=m.float().to(dev).to(BF), act=1), xr.grad * (_r(m) > 0))
    else:
        _close_bf16(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), stride=s), xr.grad)
    for beta in (0.0, 1.0, 0.5):
        dw = torch.full((k, k, C, K), 2.0, device=dev)
        db = torch.full((K,), 3.0, device=dev)
        H.conv2d_wgrad(xd, dyd, (k, k, C, K), stride=s, out=dw, beta=beta, db=db, db_beta=beta)
        ref = wr.grad + 2.0 * beta
        assert (dw.double().cpu() - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1e-6), beta
        refb = _r(dy).sum((0, 1, 2)) + 3.0 * beta
        assert (db.double().cpu() - refb).abs().max().item() <= 2e-5 * max(refb.abs().max().item(), 1e-6), beta


@pytest.mark.parametrize("shape", [(16, 64, 64, 128), (3, 9, 7, 72), (2, 4, 4, 512)])
def test_batchnorm_on_bf16_tensors(dev, shape):
    """Training-mode batch norm (tflib/ops/batchnorm.py:30) + fused LeakyReLU on bf16 tensors without conversion passes
    (dpig_bn_fwd_bf16 / _bwd_bf16): forward, statistics, dx, dscale, doffset against the fp64 oracle evaluated on the bf16 operands;
    outputs within one bf16 rounding, fp32 results at the fp32 kernels' bar."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    C = shape[-1]
    x = (_rand(shape, 41) * 2 - 0.3)
    sc = _rand((C,), 42) * 0.5 + 1.0
    of = _rand((C,), 43)
    dy = _rand(shape, 44)
    xr = _r(x).requires_grad_(True)
    scr, ofr = sc.float().double().requires_grad_(True), of.float().double().requires_grad_(True)
    y = O.leaky_relu(O.batchnorm_train(xr, scr, ofr), 0.2)
    xd = x.float().to(dev).to(BF)
    yg, mean, rstd = H.bn_fwd(xd, sc.float().to(dev), of.float().to(dev), 1e-5, 2, 0.2)
    assert yg.dtype == BF
    _close_bf16(yg, y.detach())
    m_ref = xr.detach().mean(dim=(0, 1, 2))
    assert (mean.double().cpu() - m_ref).abs().max().item() < 1e-5 * max(m_ref.abs().max().item(), 1.0)
    # backward from the kernel's own (rounded) activation, as the trainer runs it
    yq = yg.double().cpu()
    dyq = _r(dy)
    dz = dyq * torch.where(yq > 0, 1.0, 0.2)
    xh = (xr.detach() - m_ref) * rstd.double().cpu()
    n = xr.numel() // C
    dsc_ref, dof_ref = (dz * xh).sum(dim=(0, 1, 2)), dz.sum(dim=(0, 1, 2))
    dx_ref = scr.detach() * rstd.double().cpu() * (dz - dof_ref / n - xh * dsc_ref / n)
    dx, dsc, dof = H.bn_bwd(dy.float().to(dev).to(BF), xd, yg, sc.float().to(dev), mean, rstd, 2, 0.2)
    assert dx.dtype == BF
    _close_bf16(dx, dx_ref)
    assert (dsc.double().cpu() - dsc_ref).abs().max().item() < 5e-5 * max(dsc_ref.abs().max().item(), 1e-6)
    assert (dof.double().cpu() - dof_ref).abs().max().item() < 5e-5 * max(dof_ref.abs().max().item(), 1e-6)
