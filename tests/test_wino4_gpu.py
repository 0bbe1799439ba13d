"""GPU parity of the Winograd F(4x4, 3x3) fp32 kernel (csrc/dpig_conv_wino4.hip) through dpig_conv2d_fwd_wino4 / _dgrad_wino4 and
the 'f32w' mode of hip_ops, against the fp64 oracle (oracle.ops.conv2d_same + its autograd gradient).  Bar: 5e-5 of max|ref| -- the
form's transforms multiply by constants up to 8 (inputs), 1/24 (filters) and 8 (outputs), one decimal digit more rounding than
F(2x2, 3x3)'s 2e-5 (profiles/r06_f43_numerics.txt: 1.3e-5 on a single layer of random data).  Shapes hit every structural edge:
all three block forms (4 x 8, 2 x 16 and 3 x 10 tiles; partial last blocks), blocks that straddle images (zero padding where the stack holds the other image), one
block column / several, one and several channel blocks and chunks, split plans, channel slices and every fused epilogue."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 5e-5


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def _close(got, ref, tol=TOL):
    ref = ref.double()
    err = (got.double().cpu() - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-12)
    assert err <= tol * scale, "max err %.3e vs max|ref| %.3e (%.2e relative)" % (err, scale, err / scale)
    return err / scale


@pytest.fixture
def wino4():
    import dpig_amd.hip_ops as H
    H.set_compute("f32w")
    prev = H.set_wino4_mode(2)          # wherever legal: the cost model would keep the small ones on F(2x2, 3x3)
    prev2 = H.set_wino_mode(2)
    yield H
    H.set_wino_mode(prev2)
    H.set_wino4_mode(prev)
    H.set_compute("f32")


# (N, H, W, C, K)
SHAPES = [
    (1, 32, 16, 64, 64),      # exactly one 4 x 8 block, one chunk-block of 8 chunks
    (2, 16, 16, 64, 128),     # one 4 x 8 block holding two images (tile rows 4 | 4): padding rows inside the block
    (2, 32, 32, 128, 64),     # 2 block columns x 2 block rows, 16 chunks
    (8, 8, 8, 64, 64),        # block form 2 x 16: eight images' tile rows in one block (2 rows each), every tile on two borders
    (16, 4, 8, 64, 64),       # one tile row per image: top and bottom padding in every tile
    (1, 64, 24, 64, 192),     # W / 4 = 6 tile columns: block form 2 x 16, three block columns, three channel blocks
    (3, 32, 48, 128, 128),    # 12 tile columns = 3 blocks of 4; 24 tile rows = 3 block rows straddling image borders
    (2, 16, 16, 448, 64),     # 56 chunks on 2 workgroups: an input-channel split plan
    (7, 24, 24, 64, 64),      # 6 x 6 tiles per image, 42 tile rows (no multiple of 16): the 3-wide form, 3 x 10 tiles in 30 of the 32 slots, partial last block
    (5, 12, 12, 64, 128),     # the ROI tower's 12 x 12 level: three tile columns, blocks of 10 tile rows over 3.3 images each
    (3, 8, 16, 64, 64),       # 4 tile columns, 6 tile rows: one partial 4 x 8 block
    (1, 4, 8, 64, 64),        # 2 tile columns, ONE tile row: a 2 x 16 block with a single live row
    (2, 12, 36, 64, 64),      # 9 tile columns: three 3-wide block columns
    (2, 6, 6, 64, 64),        # sides no multiples of 4: no F(4x4) form, stays on F(2x2, 3x3)
]


def _has_form(H, N, Hh, W, C, K):
    d = H._desc(N, Hh, W, C, K, 3, 3, 1, C, K)
    return bool(H.lib().dpig_conv2d_wino4_eligible(ctypes.byref(d), 0))


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_and_dgrad_against_oracle(dev, wino4, shape):
    H = wino4
    from oracle import ops as O
    N, Hh, W, C, K = shape
    x = _rand((N, Hh, W, C), 1).requires_grad_(True)
    w = _rand((3, 3, C, K), 2, 1.5 / (9 * C) ** 0.5)
    b = _rand((K,), 3)
    ref = O.conv2d_same(x, w, b, 1)
    dy = _rand(tuple(ref.shape), 4)
    xd, wd, bd, dyd = x.detach().float().to(dev), w.float().to(dev), b.float().to(dev), dy.float().to(dev)
    form = _has_form(H, N, Hh, W, C, K)
    assert form == (shape != (2, 6, 6, 64, 64))
    H.PROFILE = []
    try:
        y = H.conv2d_fwd(xd, wd, bd)
        dx = H.conv2d_dgrad(dyd, wd, (N, Hh, W, C))
        kinds = [r[0] for r in H.PROFILE]
    finally:
        H.PROFILE = None
    if form:
        assert kinds == ["conv_fwd_wino4", "conv_dgrad_wino4"], kinds      # no silent fall-back
    else:
        assert kinds == ["conv_fwd_wino", "conv_dgrad_wino"], kinds        # shapes without the block form stay on F(2x2, 3x3)
    if shape == (2, 16, 16, 448, 64):
        dsc = H._desc(N, Hh, W, C, K, 3, 3, 1, C, K)
        assert H.lib().dpig_conv2d_wino4_workspace_bytes(ctypes.byref(dsc), 0) >= 2 * N * Hh * W * K * 4
    # operands are fp32-rounded on the device: compare with the oracle on the same rounded values
    ref32 = O.conv2d_same(xd.cpu().double(), wd.cpu().double(), bd.cpu().double(), 1)
    _close(y, ref32)
    xr = xd.cpu().double().requires_grad_(True)
    (rdx32,) = torch.autograd.grad(O.conv2d_same(xr, wd.cpu().double(), None, 1), xr, dyd.cpu().double())
    _close(dx, rdx32)
    assert torch.equal(H.conv2d_fwd(xd, wd, bd), y)                        # repeatable
    assert torch.equal(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C)), dx)


@pytest.mark.parametrize("geom", [(2, 16, 16), (8, 8, 8), (5, 12, 12)], ids=["block4x8", "block2x16", "block3x10"])
def test_fused_epilogues_and_channel_slices(dev, wino4, geom):
    H = wino4
    from oracle import ops as O
    (N, Hh, W), C, K = geom, 64, 128
    x, w, b = _rand((N, Hh, W, C), 1), _rand((3, 3, C, K), 2, 0.1), _rand((K,), 3)
    res = _rand((N, Hh, W, K), 4)
    f = lambda t: t.float().to(dev)
    r = lambda t: t.float().double()
    xd, wd, bd, rd = f(x), f(w), f(b), f(res)
    xr = r(x).requires_grad_(True)
    conv0 = O.conv2d_same(xr, r(w), None, 1)
    conv = conv0.detach() + r(b)
    H.PROFILE = []
    try:
        _close(H.conv2d_fwd(xd, wd, None), conv0.detach())
        _close(H.conv2d_fwd(xd, wd, bd, act=1), O.relu(conv))
        _close(H.conv2d_fwd(xd, wd, bd, act=2, alpha=0.2), O.leaky_relu(conv, 0.2))
        _close(H.conv2d_fwd(xd, wd, bd, act=1, residual=rd), O.relu(conv + r(res)))
        out, out_act = torch.empty((N, Hh, W, K), device=dev), torch.empty((N, Hh, W, K), device=dev)
        H.conv2d_fwd(xd, wd, bd, act=1, residual=rd, res_after_act=True, out=out, out_act=out_act)
        _close(out_act, O.relu(conv))
        _close(out, O.relu(conv) + r(res))
        # channel slices of wider buffers on both sides (how the decoder's concats are realised)
        xbig = torch.zeros((N, Hh, W, C + 64), device=dev)
        xbig[..., 64:] = xd
        ybig = torch.full((N, Hh, W, K + 64), 7.0, device=dev)
        H.conv2d_fwd(xbig[..., 64:], wd, bd, act=1, out=ybig[..., :K])
        _close(ybig[..., :K], O.relu(conv))
        assert bool((ybig[..., K:] == 7.0).all())
        # dgrad: plain, * mask, (+ accum) * mask, + accum
        dy = _rand((N, Hh, W, K), 6)
        conv0.backward(r(dy))
        dyd = f(dy)
        acc, m = _rand((N, Hh, W, C), 7), _rand((N, Hh, W, C), 8)
        ad, md = f(acc), f(m)
        _close(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C)), xr.grad)
        _close(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), mask=md, act=1), xr.grad * (r(m) > 0))
        _close(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), accum=ad, mask=md, act=2, alpha=0.2),
               (xr.grad + r(acc)) * torch.where(r(m) > 0, torch.ones_like(r(m)), torch.full_like(r(m), 0.2)))
        _close(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), accum=ad), xr.grad + r(acc))
        kinds = set(rec[0] for rec in H.PROFILE)
    finally:
        H.PROFILE = None
    assert kinds == {"conv_fwd_wino4", "conv_dgrad_wino4"}, kinds


def test_filter_images_follow_the_optimizer(dev, wino4):
    """A WinoFilters set makes a parameter's F(4x4) images on first request, refreshes them from the master afterwards, and drops
    the parameter from the F(2x2) refresh once its layers have only asked for the F(4x4) form."""
    H = wino4
    from oracle import ops as O
    N, Hh, W, C, K = 1, 32, 16, 64, 64
    p4 = torch.nn.Parameter(_rand((3, 3, C, K), 2, 0.1).float().to(dev))
    p2 = torch.nn.Parameter(_rand((3, 3, C, K), 3, 0.1).float().to(dev))
    wf = H.WinoFilters([p4, p2])
    x = _rand((N, Hh, W, C), 1).float().to(dev)
    x2 = _rand((N, 6, 6, C), 5).float().to(dev)                 # 6 x 6: no F(4x4) form
    y0 = H.conv2d_fwd(x, p4)
    H.conv2d_fwd(x2, p2)
    assert hasattr(p4, "_dpig_wino4") and not hasattr(p2, "_dpig_wino4")
    with torch.no_grad():
        p4.mul_(2.0)
        p2.mul_(2.0)
    wf.refresh()                                                # the set's first optimizer step: prune + refresh
    assert not hasattr(p4, "_dpig_wino") and hasattr(p2, "_dpig_wino") and wf.pruned
    y1 = H.conv2d_fwd(x, p4)
    _close(y1, 2 * y0.double().cpu(), 1e-6)
    _close(H.conv2d_fwd(x2, p2), O.conv2d_same(x2.cpu().double(), p2.detach().cpu().double(), None, 1), 2e-5)
    # an F(2x2) request for the pruned filter is still served (made on the spot)
    prev = H.set_wino4_mode(0)
    try:
        _close(H.conv2d_fwd(x, p4), y1.double().cpu(), 4e-5)
    finally:
        H.set_wino4_mode(prev)
    wf.detach()


def test_stage1_step_with_the_large_maps_on_f4_equals_the_exact_mode(dev):
    """Config(compute_dtype='f32w') with the F(4x4, 3x3) kernel wherever the layer has the form: one train step of the stage-I trainer
    (width 64) against the same step in 'f32' mode -- losses to 5e-5 (the batch-norm critic's, evaluated after the generator's update,
    to 1e-3), generator output to 1e-4 of its range; after a second step (not compared: Adam's first updates are sign-like, so the two
    arithmetic orders' 1e-6 gradient differences move individual weights by the learning rate) the F(4x4) images still follow the
    masters and filters that only use them have left the F(2x2) refresh."""
    import numpy as np
    import dpig_amd.hip_ops as H
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
    res = {}
    prev_wino, prev4 = H.set_wino_mode(2), H.set_wino4_mode(2)
    try:
        for mode in ("f32", "f32w"):
            lib.delete_all_params(); slim.reset_scopes()
            np.random.seed(0)
            B = 2
            tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, conv_hidden_num=64, z_num=16, compute_dtype=mode, g_lr=1e-3, d_lr=1e-3), dev)
            bg = synthetic.to_device(synthetic.make_batch(B, seed=21), dev)
            bd = synthetic.to_device(synthetic.make_batch(B, seed=22), dev)
            tr.init_net(bg)
            tr.step = 1
            H.PROFILE = []
            out = tr.train_step(bg, bd)
            kinds = set(r[0] for r in H.PROFILE)
            H.PROFILE = None
            res[mode] = ({k: float(v) for k, v in out.items() if hasattr(v, "numel") and v.numel() == 1}, out["G"].clone())
            tr.train_step(bg, bd)
            if mode == "f32w":
                assert "conv_fwd_wino4" in kinds and "conv_dgrad_wino4" in kinds and "conv_wgrad_wino" in kinds, kinds
                wf = tr.G_flat.wino
                assert len(wf.p4) > 5 and wf.pruned and len(wf.p2) < len(wf.params)
                for p in wf.p4[:3]:                                            # refreshed after Adam moved the filters
                    fresh = H.wino4_images(p.data.clone())
                    assert torch.equal(p._dpig_wino4[0], fresh[0]) and torch.equal(p._dpig_wino4[1], fresh[1])
                for p in wf.p2[:3]:
                    assert torch.equal(p._dpig_wino[0], H.wino_images(p.data.clone())[0])
            else:
                assert "conv_fwd_wino4" not in kinds
        for k, tol in (("g_loss", 5e-5), ("L1Loss", 5e-5), ("d_loss", 1e-3)):
            assert abs(res["f32w"][0][k] - res["f32"][0][k]) <= tol * abs(res["f32"][0][k]), (k, res["f32w"][0][k], res["f32"][0][k])
        Gd = (res["f32w"][1] - res["f32"][1]).abs().max().item()
        assert Gd <= 1e-4 * res["f32"][1].abs().max().item(), Gd
    finally:
        H.PROFILE = None
        H.set_wino4_mode(prev4)
        H.set_wino_mode(prev_wino)
        H.set_compute("f32")
        lib.delete_all_params(); slim.reset_scopes()
