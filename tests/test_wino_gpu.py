"""GPU parity of the Winograd F(2x2, 3x3) fp32 kernel (csrc/dpig_conv_wino.hip) through dpig_conv2d_fwd_wino / _dgrad_wino and the
'f32w' mode of hip_ops: same fp64 oracle (oracle.ops.conv2d_same + its autograd gradients), same 2e-5 of max|ref| bar as the direct
fp32 kernels of tests/test_conv_gpu.py -- the arithmetic type is the same, only the evaluation order differs -- on shapes that hit every
structural edge: a partial last row block, tiles that straddle images, exactly one block, several channel blocks / chunks, image
borders on every side, channel slices of wider buffers, and every fused epilogue of the family."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def _close(got, ref, tol=2e-5):
    ref = ref.double()
    err = (got.double().cpu() - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-12)
    assert err <= tol * scale, "max err %.3e vs max|ref| %.3e (%.2e relative)" % (err, scale, err / scale)
    return err / scale


@pytest.fixture
def wino():
    import dpig_amd.hip_ops as H
    H.set_compute("f32w")
    prev = H.set_wino_mode(2)           # wherever legal: the cost model would keep these small layers on the direct kernel
    prev4 = H.set_wino4_mode(0)         # (this file is about the F(2x2, 3x3) kernels: tests/test_wino4_gpu.py holds the F(4x4, 3x3) ones)
    yield H
    H.set_wino4_mode(prev4)
    H.set_wino_mode(prev)
    H.set_compute("f32")


# (N, H, W, C, K)
SHAPES = [
    (2, 8, 6, 64, 64),        # 24 tiles: one partial block, one chunk-block
    (1, 16, 16, 128, 64),     # exactly 64 tiles
    (3, 10, 14, 64, 128),     # 105 tiles: blocks straddle images, partial last block, two channel blocks
    (2, 4, 4, 192, 64),       # 2 x 2 tiles per image: every tile touches two borders; 24 chunks
    (5, 2, 2, 64, 64),        # one tile per image: all four borders
    (1, 32, 24, 64, 192),     # 192 tiles x 3 channel blocks; raw-gather blocks of 4 x 16 tiles (wino_block_kernel): 3 block columns
    (2, 16, 8, 64, 64),       # one 4 x 16 block that straddles two images: zero padding where the stack holds the other image
    (4, 8, 16, 64, 128),      # four images in every block column (tile rows 4 | 4 | 4 | 4), two block columns
    (3, 32, 16, 128, 64),     # 3 x 2 blocks, 16 chunks
]


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_and_dgrad_against_oracle(dev, wino, shape):
    H = wino
    from oracle import ops as O
    N, Hh, W, C, K = shape
    x = _rand((N, Hh, W, C), 1).requires_grad_(True)
    w = _rand((3, 3, C, K), 2, 1.5 / (9 * C) ** 0.5)
    b = _rand((K,), 3)
    ref = O.conv2d_same(x, w, b, 1)
    dy = _rand(tuple(ref.shape), 4)
    (rdx,) = torch.autograd.grad(ref, x, dy)
    xd, wd, bd, dyd = x.detach().float().to(dev), w.float().to(dev), b.float().to(dev), dy.float().to(dev)
    H.PROFILE = []
    try:
        y = H.conv2d_fwd(xd, wd, bd)
        dx = H.conv2d_dgrad(dyd, wd, (N, Hh, W, C))
        dw = torch.full((3, 3, C, K), 7.0, device=dev)
        db = torch.empty((K,), device=dev)
        H.conv2d_wgrad(xd, dyd, (3, 3, C, K), out=dw, db=db)
        dw2 = H.conv2d_wgrad(xd, dyd, (3, 3, C, K), out=dw.clone(), beta=1.0)      # accumulation into an existing gradient
        kinds = [r[0] for r in H.PROFILE]
    finally:
        H.PROFILE = None
    assert kinds == ["conv_fwd_wino", "conv_dgrad_wino", "conv_wgrad_wino", "conv_wgrad_wino"], kinds   # no silent direct fall-back
    if shape in ((2, 4, 4, 192, 64), (3, 32, 16, 128, 64)):       # few workgroups, >= 16 chunks: these run as input-channel SPLIT plans
        import ctypes
        dsc = H._desc(N, Hh, W, C, K, 3, 3, 1, C, K)
        assert H.lib().dpig_conv2d_wino_workspace_bytes(ctypes.byref(dsc), 0) >= 2 * N * Hh * W * K * 4
    xr2, wr2 = xd.cpu().double(), wd.cpu().double().requires_grad_(True)
    (rdw,) = torch.autograd.grad(O.conv2d_same(xr2, wr2, None, 1), wr2, dyd.cpu().double())
    _close(dw, rdw)
    _close(dw2, 2 * rdw)
    _close(db, dyd.cpu().double().sum(dim=(0, 1, 2)))                       # (formed inside the same two launches)
    db2 = db.clone()
    H.conv2d_wgrad(xd, dyd, (3, 3, C, K), out=dw.clone(), db=db2, db_beta=1.0)
    _close(db2, 2 * dyd.cpu().double().sum(dim=(0, 1, 2)))
    assert torch.equal(H.conv2d_wgrad(xd, dyd, (3, 3, C, K)), dw)          # repeatable (fixed summation order)
    # operands are fp32-rounded on the device: compare with the oracle on the same rounded values
    ref32 = O.conv2d_same(xd.cpu().double().requires_grad_(True), wd.cpu().double(), bd.cpu().double(), 1)
    _close(y, ref32.detach())
    xr = xd.cpu().double().requires_grad_(True)
    (rdx32,) = torch.autograd.grad(O.conv2d_same(xr, wd.cpu().double(), None, 1), xr, dyd.cpu().double())
    _close(dx, rdx32)
    # and the direct kernels agree with it to the same bar (two evaluation orders of one fp32 sum)
    H.set_compute("f32")
    try:
        yd = H.conv2d_fwd(xd, wd, bd)
    finally:
        H.set_compute("f32w")
    assert float((y - yd).abs().max()) <= 4e-5 * float(yd.abs().max())
    assert torch.equal(H.conv2d_fwd(xd, wd, bd), y)                       # repeatable


@pytest.mark.parametrize("geom", [(3, 10, 14), (2, 16, 8)], ids=["generic", "block"])
def test_fused_epilogues_and_channel_slices(dev, wino, geom):
    H = wino
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
    _close(H.conv2d_fwd(xd, wd, None), conv0.detach())
    _close(H.conv2d_fwd(xd, wd, bd, act=1), O.relu(conv))
    _close(H.conv2d_fwd(xd, wd, bd, act=2, alpha=0.2), O.leaky_relu(conv, 0.2))
    _close(H.conv2d_fwd(xd, wd, bd, act=1, residual=rd), O.relu(conv + r(res)))
    out, out_act = torch.empty((N, Hh, W, K), device=dev), torch.empty((N, Hh, W, K), device=dev)
    H.conv2d_fwd(xd, wd, bd, act=1, residual=rd, res_after_act=True, out=out, out_act=out_act)
    _close(out_act, O.relu(conv))
    _close(out, O.relu(conv) + r(res))
    _close(H.conv2d_fwd(xd, wd, bd, act=1, residual=rd, res_after_act=True), O.relu(conv) + r(res))
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
    H.PROFILE = []
    try:
        _close(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C)), xr.grad)
        _close(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), mask=md, act=1), xr.grad * (r(m) > 0))
        _close(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), accum=ad, mask=md, act=2, alpha=0.2), (xr.grad + r(acc)) * torch.where(r(m) > 0, 1.0, 0.2))
        _close(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), accum=ad), xr.grad + r(acc))
        assert all(k[0] == "conv_dgrad_wino" for k in H.PROFILE) and len(H.PROFILE) == 4
    finally:
        H.PROFILE = None


def test_layers_without_a_winograd_form_stay_on_the_direct_kernels(dev, wino):
    """stride 2, 5x5, 1x1, odd maps, channel counts that are not multiples of 64, the class-indexed residual: `conv2d_wino_eligible` says
    no and the call runs the direct kernel (same results as 'f32' mode, bit for bit)."""
    H = wino
    cases = [(2, 8, 6, 64, 64, 3, 2), (2, 8, 6, 64, 64, 5, 1), (2, 8, 6, 64, 64, 1, 1), (2, 7, 6, 64, 64, 3, 1), (2, 8, 6, 96, 64, 3, 1),
             (2, 8, 6, 64, 32, 3, 1)]
    for N, Hh, W, C, K, k, s in cases:
        x, w = _rand((N, Hh, W, C), 1).float().to(dev), _rand((k, k, C, K), 2, 0.1).float().to(dev)
        H.PROFILE = []
        try:
            y = H.conv2d_fwd(x, w, None, stride=s)
            kinds = [r[0] for r in H.PROFILE]
        finally:
            H.PROFILE = None
        assert kinds == ["conv_fwd_mfma"], (kinds, (N, Hh, W, C, K, k, s))
        H.set_compute("f32")
        try:
            assert torch.equal(H.conv2d_fwd(x, w, None, stride=s), y)
        finally:
            H.set_compute("f32w")


@pytest.mark.parametrize("order", ["0", "1"], ids=["filter-major", "activation-major"])
def test_both_workgroup_orders_against_the_oracle(dev, order):
    """The forward / dgrad kernels walk their workgroups filter-major or activation-major, picked per launch by HBM bytes
    (WParams.xmajor); DPIG_WINO_XMAJOR pins the order for a whole process.  The oracle and epilogue tests above, every shape, in a
    sub-process under each order."""
    import os, subprocess, sys
    env = dict(os.environ, DPIG_WINO_XMAJOR=order)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_forward_and_dgrad_against_oracle or test_fused_epilogues_and_channel_slices"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_one_launch_filter_refresh_equals_the_per_filter_transforms(dev, wino):
    """WinoFilters.refresh: dpig_wino_filter_transform_jobs writes, for a set of filters of different sizes (and the 5x5 / thin ones it
    must skip), exactly the images dpig_wino_filter_transform writes one filter at a time; a master whose storage moved takes the
    per-filter path and still ends up right."""
    H = wino
    shapes = [(3, 3, 64, 64), (3, 3, 128, 64), (5, 5, 64, 64), (3, 3, 64, 192), (3, 3, 256, 256), (3, 3, 64, 3), (3, 3, 192, 128)]
    params = [torch.nn.Parameter(_rand(sh, 10 + i, 0.1).float().to(dev)) for i, sh in enumerate(shapes)]
    wf = H.WinoFilters(params)
    assert len(wf.params) == 5 and wf.plan2["total"] == sum((p.shape[2] // 64) * (p.shape[3] // 8) for p in wf.params)
    for p in wf.params:
        uf, ud = H.wino_images(p.data.clone())
        assert torch.equal(p._dpig_wino[0], uf) and torch.equal(p._dpig_wino[1], ud)
    with torch.no_grad():
        for p in wf.params:
            p.data.mul_(1.5)                                  # in place: the job table still names the masters
    wf.refresh()
    for p in wf.params:
        uf, ud = H.wino_images(p.data.clone())
        assert torch.equal(p._dpig_wino[0], uf) and torch.equal(p._dpig_wino[1], ud)
    params[1].data = params[1].data.clone() * 2.0             # storage moved: per-filter launches
    wf.refresh()
    for p in wf.params:
        uf, ud = H.wino_images(p.data.clone())
        assert torch.equal(p._dpig_wino[0], uf) and torch.equal(p._dpig_wino[1], ud)
    wf.detach()


FULL = [("dec4", 16, 128, 64, 256), ("dec3", 16, 64, 32, 512), ("dec2", 16, 32, 16, 768), ("roi b1", 112, 24, 24, 256),
        # DeepFashion 256 x 256 at batch 8 (trainer_256.py sizes; bench.py's df256_f32 line runs them by Winograd too)
        ("df enc1", 8, 256, 256, 128), ("df dec3", 8, 64, 64, 768)]


@pytest.mark.parametrize("form", ["f4", "f2"])
@pytest.mark.parametrize("layer", FULL, ids=[l[0] for l in FULL])
def test_full_size_layers_against_the_sampled_oracle(dev, layer, form):
    """BASELINE configs[1] (and two DeepFashion) layer sizes under the DEFAULT selection (cost model: every one of these layers runs the
    F(4x4, 3x3) forward / dgrad kernel; "f2": with that form switched off, the F(2x2, 3x3) kernel): 4096 sampled output / input
    positions against oracle.ops.conv2d_same*_sampled in fp64, forward + bias + ReLU and dgrad."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    _, N, Hh, W, C = layer
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    w = (torch.rand((3, 3, C, C), device=dev, generator=g) * 2 - 1) * (1.5 / (9 * C) ** 0.5)
    b = torch.rand((C,), device=dev, generator=g) - 0.5
    dy = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    gc = torch.Generator().manual_seed(6)
    P = 4096
    n, oy, ox = torch.randint(0, N, (P,), generator=gc), torch.randint(0, Hh, (P,), generator=gc), torch.randint(0, W, (P,), generator=gc)
    oy[:64], ox[64:128] = 0, 0
    oy[128:192], ox[192:256] = Hh - 1, W - 1
    xc, wc, bc, dyc = x.cpu(), w.cpu(), b.cpu(), dy.cpu()
    ref_y = torch.relu(O.conv2d_same_sampled(xc, wc, bc, 1, n, oy, ox))
    ref_dx = O.conv2d_same_dgrad_sampled(dyc, wc, (N, Hh, W, C), 1, n, oy, ox)
    taps = [(r, c) for r in range(3) for c in range(3)]
    ci, co = torch.randperm(C, generator=gc)[:48], torch.randperm(C, generator=gc)[:48]
    ref_dw = O.conv2d_same_wgrad_sampled(xc, dyc, (3, 3, C, C), 1, taps, ci, co)
    H.set_compute("f32w")
    prev4 = H.set_wino4_mode(0) if form == "f2" else H.get_wino4_mode()
    H.PROFILE = []
    try:
        y = H.conv2d_fwd(x, w, b, act=1)
        dx = H.conv2d_dgrad(dy, w, (N, Hh, W, C))
        dw = H.conv2d_wgrad(x, dy, (3, 3, C, C))
        kinds = [r[0] for r in H.PROFILE]
    finally:
        H.PROFILE = None
        H.set_wino4_mode(prev4)
        H.set_compute("f32")
    sfx = "4" if form == "f4" else ""
    assert kinds == ["conv_fwd_wino" + sfx, "conv_dgrad_wino" + sfx, "conv_wgrad_wino"], kinds
    idx = (n.to(dev), oy.to(dev), ox.to(dev))
    _close(y[idx], ref_y, 1e-4)
    _close(dx[idx], ref_dx, 1e-4)
    _close(dw.reshape(9, C, C)[:, ci.to(dev)][:, :, co.to(dev)], ref_dw, 2e-4)


def test_stage1_step_in_winograd_mode_equals_the_exact_mode(dev):
    """Config(compute_dtype='f32w'): one g_optim + d_optim of the stage-I trainer (width 64, so that the 3x3 layers have Winograd forms)
    against the same step in 'f32' mode: losses to 1e-5 (the batch-norm critic's to 2e-4), generator output to 1e-4 of its range, and the Winograd images follow the
    masters through the optimizer step."""
    import numpy as np
    import dpig_amd.hip_ops as H
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
    res = {}
    prev_wino = H.set_wino_mode(2)
    prev4 = H.set_wino4_mode(0)          # (the F(2x2, 3x3) kernels' step test; the F(4x4, 3x3) twin is in tests/test_wino4_gpu.py)
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
            if mode == "f32w":
                assert len(tr.G_flat.wino.params) > 20
                p0 = tr.G_flat.wino.params[0]
                u0 = p0._dpig_wino[0].clone()
            H.PROFILE = []
            out = tr.train_step(bg, bd)
            kinds = set(r[0] for r in H.PROFILE)
            H.PROFILE = None
            if mode == "f32w":
                assert "conv_fwd_wino" in kinds and "conv_dgrad_wino" in kinds and "conv_wgrad_wino" in kinds
                assert not torch.equal(p0._dpig_wino[0], u0)                 # refreshed after Adam moved the filter
                assert torch.equal(p0._dpig_wino[0], H.wino_images(p0.data.clone())[0])
            else:
                assert "conv_fwd_wino" not in kinds
            res[mode] = ({k: float(v) for k, v in out.items() if hasattr(v, "numel") and v.numel() == 1}, out["G"].clone(),
                         tr.G_flat.grad.clone())
        # (d_loss is evaluated AFTER the generator's update, through a batch-norm critic on 2 images: it amplifies the two fp32
        # summation orders' 1e-6 differences most -- 2e-5 at full width, scripts/diag_wino_traj.py)
        for k, tol in (("g_loss", 1e-5), ("L1Loss", 1e-5), ("d_loss", 2e-4)):
            assert abs(res["f32w"][0][k] - res["f32"][0][k]) <= tol * abs(res["f32"][0][k]), (k, res["f32w"][0][k], res["f32"][0][k])
        Gd = (res["f32w"][1] - res["f32"][1]).abs().max().item()
        assert Gd <= 1e-4 * res["f32"][1].abs().max().item(), Gd
    finally:
        H.PROFILE = None
        H.set_wino4_mode(prev4)
        H.set_wino_mode(prev_wino)
        H.set_compute("f32")
        lib.delete_all_params(); slim.reset_scopes()
