"""Pins the CPU oracle (oracle/ops.py, torch-CPU) -- the reference ships no golden vectors, so:
  * every op is cross-checked against the independent numpy loop restatement oracle/naive.py;
  * analytic known-answer tests nail the TF-1.4 semantics of SURVEY.md Appendix B.
CPU only (runs in the not-gpu tier)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import naive
from oracle import ops as O


def rnd(shape, seed, scale=1.0):
    return np.random.default_rng(seed).uniform(-1, 1, size=shape) * scale


T = lambda a: torch.tensor(a, dtype=torch.float64)  # noqa: E731


# ---- SAME padding (Appendix B-1) ------------------------------------------------------------------
@pytest.mark.parametrize("inp,k,s,out,before,after", [
    (128, 3, 1, 128, 1, 1), (128, 3, 2, 64, 0, 1), (64, 5, 2, 32, 1, 2), (7, 3, 2, 4, 1, 1),
    (9, 5, 2, 5, 2, 2), (48, 3, 2, 24, 0, 1), (3, 3, 1, 3, 1, 1), (16, 1, 1, 16, 0, 0)])
def test_same_pad_kat(inp, k, s, out, before, after):
    assert O.same_pad(inp, k, s) == (out, before, after)
    assert naive.same_pad(inp, k, s) == (out, before)


# ---- conv2d ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 8, 6, 3, 5, 3, 1), (2, 8, 6, 4, 5, 3, 2), (1, 9, 7, 3, 4, 5, 2),
                                   (2, 5, 5, 2, 3, 1, 1), (1, 7, 9, 3, 2, 3, 2)])
def test_conv_matches_naive(shape):
    N, H, W, C, K, k, s = shape
    x, w, b = rnd((N, H, W, C), 1), rnd((k, k, C, K), 2), rnd((K,), 3)
    got = O.conv2d_same(T(x), T(w), T(b), s).numpy()
    np.testing.assert_allclose(got, naive.conv2d_same(x, w, b, s), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("shape", [(2, 8, 6, 3, 5, 3, 1), (2, 8, 6, 4, 5, 3, 2), (1, 9, 7, 3, 4, 5, 2)])
def test_conv_grads_match_naive(shape):
    N, H, W, C, K, k, s = shape
    x, w = T(rnd((N, H, W, C), 1)).requires_grad_(True), T(rnd((k, k, C, K), 2)).requires_grad_(True)
    y = O.conv2d_same(x, w, None, s)
    dy = rnd(tuple(y.shape), 3)
    y.backward(T(dy))
    np.testing.assert_allclose(x.grad.numpy(), naive.conv2d_same_dgrad(dy, w.detach().numpy(), (N, H, W, C), s),
                               rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(w.grad.numpy(), naive.conv2d_same_wgrad(x.detach().numpy(), dy, (k, k, C, K), s),
                               rtol=1e-11, atol=1e-12)


def test_conv_delta_image_is_flipped_kernel():
    """Cross-correlation (Appendix B-2): a unit impulse at p paints w[ky,kx] at p-(ky-1, kx-1)."""
    w = rnd((3, 3, 1, 1), 4)
    x = np.zeros((1, 7, 7, 1)); x[0, 3, 3, 0] = 1.0
    y = O.conv2d_same(T(x), T(w), None, 1).numpy()[0, :, :, 0]
    np.testing.assert_allclose(y[2:5, 2:5], w[::-1, ::-1, 0, 0], atol=1e-15)


def test_conv_constant_image_border_classes():
    """Constant image through a SAME 3x3 conv: 9 border classes = sums over the valid taps
    (the structure behind the G.stem collapse); stride 2 pads only bottom/right."""
    w = rnd((3, 3, 1, 1), 5)[:, :, 0, 0]
    y = O.conv2d_same(T(np.ones((1, 6, 6, 1))), T(w[:, :, None, None]), None, 1).numpy()[0, :, :, 0]
    assert abs(y[2, 3] - w.sum()) < 1e-14
    assert abs(y[0, 0] - w[1:, 1:].sum()) < 1e-14
    assert abs(y[0, 3] - w[1:, :].sum()) < 1e-14
    assert abs(y[5, 5] - w[:2, :2].sum()) < 1e-14
    y2 = O.conv2d_same(T(np.ones((1, 6, 6, 1))), T(w[:, :, None, None]), None, 2).numpy()[0, :, :, 0]
    assert y2.shape == (3, 3)
    assert abs(y2[0, 0] - w.sum()) < 1e-14              # no top/left pad for even input (pad = (0,1))
    assert abs(y2[2, 2] - w[:2, :2].sum()) < 1e-14


def test_deconv_is_adjoint_of_strided_conv():
    """tflib Deconv2D == conv2d_transpose == gradient of the stride-2 SAME conv: <F(u), v> = <u, F^T(v)>."""
    wt = T(rnd((5, 5, 4, 3), 6))       # (k, k, Cout, Cin) as deconv2d.py:61-67
    v = T(rnd((2, 4, 3, 3), 7))        # deconv input  [N,H,W,Cin]
    u = T(rnd((2, 8, 6, 4), 8))        # deconv output-shaped
    Fu = O.conv2d_same(u, wt, None, 2)
    Ftv = O.conv2d_transpose_same(v, wt, None, 2)
    assert tuple(Ftv.shape) == (2, 8, 6, 4)
    assert abs((Fu * v).sum().item() - (u * Ftv).sum().item()) < 1e-10


# ---- norms ----------------------------------------------------------------------------------------
def test_batchnorm_matches_naive_and_kat():
    x, sc, of = rnd((3, 4, 5, 6), 1, 2.0) + 0.5, rnd((6,), 2) + 1.5, rnd((6,), 3)
    got = O.batchnorm_train(T(x), T(sc), T(of)).numpy()
    np.testing.assert_allclose(got, naive.batchnorm_train(x, sc, of), rtol=1e-12, atol=1e-12)
    # KAT: two values {0, 2} per channel -> mean 1, biased var 1 -> (x-1)/sqrt(1+1e-5)
    xk = np.zeros((2, 1, 1, 1)); xk[1] = 2.0
    yk = O.batchnorm_train(T(xk), T(np.ones(1)), T(np.zeros(1))).numpy().reshape(-1)
    np.testing.assert_allclose(yk, np.array([-1.0, 1.0]) / math.sqrt(1 + 1e-5), rtol=1e-14)


def test_layernorm_matches_naive():
    x, sc, of = rnd((3, 4, 5, 6), 1, 2.0) - 0.2, rnd((6,), 2) + 1.5, rnd((6,), 3)
    np.testing.assert_allclose(O.layernorm(T(x), T(sc), T(of)).numpy(), naive.layernorm(x, sc, of), rtol=1e-12,
                               atol=1e-12)


# ---- resize / crop --------------------------------------------------------------------------------
def test_upsample_matches_naive():
    x = rnd((2, 3, 4, 5), 1)
    up = O.upsample2x(T(x)).numpy()
    np.testing.assert_array_equal(up, naive.upsample2x(x))
    assert up[0, 5, 7, 2] == x[0, 2, 3, 2]


def test_crop_and_resize_matches_naive_incl_grad():
    H, W = 16, 8
    img = T(rnd((2, H, W, 3), 1)).requires_grad_(True)
    px = np.array([[0, 0, 1, 1], [2, 1, 12, 6], [0, 0, 15, 7], [5, 5, 6, 6]], dtype=np.float64)
    boxes = np.concatenate([px / np.array([H, W, H, W]), [[-0.3, 0.2, 1.2, 0.7]]])
    ind = np.array([0, 1, 1, 0, 1])
    out = O.crop_and_resize(img, T(boxes), torch.tensor(ind), 6, 5)
    np.testing.assert_allclose(out.detach().numpy(), naive.crop_and_resize(img.detach().numpy(), boxes, ind, 6, 5),
                               rtol=1e-12, atol=1e-13)
    dout = rnd(tuple(out.shape), 2)
    out.backward(T(dout))
    np.testing.assert_allclose(img.grad.numpy(), naive.crop_and_resize_grad_image(dout, boxes, ind, (2, H, W, 3)),
                               rtol=1e-11, atol=1e-13)
    # KAT: the full-image box with crop == image size is the identity (y1=x1=0, y2=x2=1)
    ident = O.crop_and_resize(img.detach(), T([[0, 0, 1, 1]]), torch.tensor([1]), H, W)
    np.testing.assert_allclose(ident[0].numpy(), img.detach()[1].numpy(), atol=1e-13)
    # the reference normalises by /H, /W (models.py:410-413): box [0,0,H,W]/[H,W] hits y2 = 1 exactly
    # and everything outside [0, H-1] is extrapolated with 0
    far = O.crop_and_resize(img.detach(), T([[1.5, 1.5, 2.0, 2.0]]), torch.tensor([0]), 3, 3)
    assert float(far.abs().max()) == 0.0


# ---- losses / optimizer ---------------------------------------------------------------------------
def test_sigmoid_ce_kat():
    x = T([0.0, 50.0, -50.0, 3.0])
    np.testing.assert_allclose(O.sigmoid_cross_entropy_with_logits(x, torch.ones(4, dtype=torch.float64)).numpy(),
                               [math.log(2), 0.0, 50.0, math.log1p(math.exp(-3.0))], atol=1e-12)
    np.testing.assert_allclose(O.sigmoid_cross_entropy_with_logits(x, torch.zeros(4, dtype=torch.float64)).numpy(),
                               naive.sigmoid_cross_entropy_with_logits(x.numpy(), 0.0), atol=1e-12)


def test_tf_adam_closed_form():
    """First TF-Adam step: m=(1-b1)g, v=(1-b2)g^2, lr_t=lr*sqrt(1-b2)/(1-b1) -> p - lr*g/(|g|+eps')
    with eps' = eps/sqrt(1-b2): a sign step for |g| >> eps'."""
    p, g = T([1.0, -2.0, 0.5]), T([0.3, -4.0, 1e-3])
    lr, b1, b2, eps = 2e-5, 0.5, 0.999, 1e-8
    p1, m1, v1 = O.tf_adam_step(p, g, torch.zeros_like(p), torch.zeros_like(p), lr, b1, b2, eps, 1)
    expect = p - lr * g / (g.abs() + eps / math.sqrt(1 - b2))
    np.testing.assert_allclose(p1.numpy(), expect.numpy(), rtol=1e-12)
    # three steps against the independent numpy restatement
    grads = [rnd((5,), 10 + i) for i in range(3)]
    pp, mm, vv = T(np.ones(5)), torch.zeros(5, dtype=torch.float64), torch.zeros(5, dtype=torch.float64)
    for t, gg in enumerate(grads, 1):
        pp, mm, vv = O.tf_adam_step(pp, T(gg), mm, vv, 1e-3, 0.5, 0.999, 1e-8, t)
    np.testing.assert_allclose(pp.numpy(), naive.tf_adam(np.ones(5), grads, 1e-3, 0.5, 0.999, 1e-8), rtol=1e-12)


# ---- model-level structure (tiny widths, CPU) -----------------------------------------------------
def test_oracle_model_shapes_names_and_losses():
    from dpig_amd import synthetic
    from oracle import models as OM
    ob = OM.batch_to_torch(synthetic.make_batch(2, seed=1))
    P = OM.ParamStore(seed=2)
    gl, aux = OM.stage1_g_loss(P, ob, hidden_num=8, z_num=4)
    dl, _ = OM.stage1_d_loss(P, ob, hidden_num=8, z_num=4)
    assert tuple(aux["G"].shape) == (2, 128, 64, 3)
    # SURVEY Appendix F creation order: stem Conv, Conv_1,_2; ROI tower Conv_3.._16 + fully_connected;
    # Bg tower Conv_17.._30 + fully_connected_1; G: Conv.._29, fully_connected(_1)
    names = list(P.p.keys())
    assert names[0] == "Encoder/G_encoder/Conv/weights"
    assert "Encoder/G_encoder/Conv_30/weights" in P.p and "Encoder/G_encoder/Conv_31/weights" not in P.p
    assert "Encoder/G_encoder/fully_connected_1/weights" in P.p
    assert "ID_AE/G/Conv_29/weights" in P.p and "ID_AE/G/Conv_30/weights" not in P.p
    assert tuple(P.p["Encoder/G_encoder/fully_connected/weights"].shape) == (3 * 3 * 8 * 5, 32)
    assert tuple(P.p["ID_AE/G/Conv/weights"].shape) == (3, 3, 352 + 18, 8)
    assert tuple(P.p["Discriminator.Output.W"].shape) == (8 * 4 * 8 * 64, 1)
    assert "Discriminator.BN2.moving_mean" in P.p and not P.trainable["Discriminator.BN2.moving_mean"]
    assert torch.isfinite(gl) and torch.isfinite(dl)
    # g_loss = sce(D(G),1) + 20*L1 (trainer.py:623)
    assert abs(gl.item() - (aux["g_loss_only"].item() + 20 * aux["L1Loss"].item())) < 1e-12


@pytest.mark.parametrize("normalized", [True, False])
def test_pose_maps_scatter_and_shifts_equal_disc(normalized):
    """utils.py:237-318 restated two ways: scatter + 49 zero-padded shifts (oracle/ops.py) vs a Euclidean disc of
    radius 4 per visible keypoint (oracle/naive.py, the reference's own numpy variant utils.py:320-340)."""
    import numpy as np
    import torch
    from oracle import naive, ops
    rng = np.random.default_rng(3)
    B, K, Hh, W = 3, 18, 32, 16
    rcv = np.zeros((B, K, 3))
    if normalized:
        rcv[..., 0] = rng.uniform(-1.2, 1.2, (B, K)); rcv[..., 1] = rng.uniform(-1.2, 1.2, (B, K))
    else:
        rcv[..., 0] = rng.integers(0, Hh, (B, K)); rcv[..., 1] = rng.integers(0, W, (B, K))
    rcv[..., 2] = (rng.uniform(size=(B, K)) < 0.8).astype(np.float64)
    rcv[0, 0] = (-1.0 if normalized else 0.0, -1.0 if normalized else 0.0, 1.0)          # corner keypoint
    rcv[0, 1, 2] = 0.5                                                                    # fractional visibility
    t = torch.from_numpy(rcv.reshape(B, K * 3))
    if normalized:
        pts = ops.coord2channel_simple_rcv(t, K, True, Hh, W)
        want = naive.pose_disc_map(rcv, K, Hh, W)
    else:
        pts = ops.coord2channel_simple_rcv(t, K, False, Hh, W)
        norm = rcv.copy()                          # the disc restatement takes normalised coordinates
        norm[..., 0] = (rcv[..., 0] + 0.25) / Hh * 2 - 1
        norm[..., 1] = (rcv[..., 1] + 0.25) / W * 2 - 1
        want = naive.pose_disc_map(norm, K, Hh, W)
    got = ops.tf_poseInflate(pts, K, 4, Hh, W).numpy()
    assert np.array_equal(got, want)
    assert ((pts.numpy() == -1) | (pts.numpy() == 2 * rcv[:, None, None, :, 2] - 1)).all()


def test_ssim_skimage_restatement_equals_window_loops():
    """trainer.py:516-521 metric: the scipy uniform_filter restatement of skimage.compare_ssim vs explicit 7x7 window
    loops with unbiased (co)variances; identical images give exactly 1."""
    import numpy as np
    from oracle import naive, ops
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (20, 15, 3), dtype=np.uint8)
    b = np.clip(a.astype(int) + rng.integers(-40, 41, a.shape), 0, 255).astype(np.uint8)
    ga, gb = ops.rgb2gray_u8(a), ops.rgb2gray_u8(b)
    R = gb.max() - gb.min()
    assert abs(ops.ssim_skimage(ga, gb, R) - naive.ssim_window_loops(ga, gb, R)) < 1e-12
    assert abs(ops.ssim_skimage(ga, ga, R) - 1.0) < 1e-12
    assert 0.0 <= gb.min() and gb.max() <= 1.0


def test_oracle_reproduces_small_golden():
    """The committed width-16 golden vectors (tests/golden/stage1_market_b2_w16.npz, written by
    `make_golden.py --small`) are re-derived from seeds by today's oracle: any change of the oracle's arithmetic, of
    the synthetic inputs or of the parameter initialisation shows up here, on the CPU, in seconds."""
    import importlib.util
    import os
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = np.load(os.path.join(here, "golden", "stage1_market_b2_w16.npz"))
    got = mg.small_outputs()
    assert set(got.keys()) == set(want.keys())
    for k in want.keys():
        scale = max(float(np.abs(want[k]).max()), 1e-12)
        assert float(np.abs(np.asarray(got[k]) - want[k]).max()) <= 1e-9 * scale, k


# ---- pinned against the reference's own python: tests/golden/pose_reference.npz holds outputs of utils.py
#      py_poseInflate / _getSparsePose / _sparse2dense, executed from the reference's text by make_pose_golden.py -------
def _pose_reference_cases():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pose_reference.npz"))
    for name in ("pixel_128x64", "normalized_128x64", "normalized_256x256"):
        rcv = z[name + "/rcv"]
        norm, H, W = (int(v) for v in z[name + "/meta"])
        n = rcv.shape[0] * H * W * rcv.shape[1]
        want = np.unpackbits(z[name + "/bits"])[:n].reshape(rcv.shape[0], H, W, rcv.shape[1]).astype(bool)
        yield name, z, rcv, bool(norm), H, W, want


def test_pose_oracle_equals_reference_py_poseInflate():
    """The TF-graph pose pipeline as restated in oracle/ops.py (coord2channel_simple_rcv -> tf_poseInflate,
    utils.py:237-318) produces exactly the maps the reference's numpy rasteriser py_poseInflate (utils.py:320-347)
    produced for the same keypoints; for pixel coordinates also the dataset converter's 'Solid' channels."""
    seen = 0
    for name, z, rcv, norm, H, W, want in _pose_reference_cases():
        B, K = rcv.shape[:2]
        t = torch.from_numpy(rcv.reshape(B, K * 3)).double()
        got = O.tf_poseInflate(O.coord2channel_simple_rcv(t, K, norm, H, W), K, 4, H, W).numpy()
        assert set(np.unique(got)) <= {-1.0, 1.0}
        assert np.array_equal(got > 0, want), name
        assert want.any(axis=(1, 2)).sum() == int((rcv[..., 2] != 0).sum())       # one disc per visible keypoint
        if name == "pixel_128x64":
            solid = np.unpackbits(z[name + "/solid_bits"])[:want.size].reshape(want.shape).astype(bool)
            assert np.array_equal(solid, want)
        seen += 1
    assert seen == 3


@pytest.mark.skipif(not __import__("os").path.exists("/root/reference/utils.py"), reason="the reference tree is not mounted")
def test_pose_fixture_is_what_the_reference_code_produces_today():
    """Where the reference is mounted (the build container, not the GPU box): re-run its four pose functions from
    their source text and compare with the committed fixture, so neither the fixture nor the generator can drift."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_pose_golden", os.path.join(here, "golden", "make_pose_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    f = gen.reference_functions()
    seen = 0
    for name, z, rcv, norm, H, W, want in _pose_reference_cases():
        dense = f["py_poseInflate"](rcv.copy(), is_normalized=norm, radius=4, img_H=H, img_W=W)
        assert np.array_equal(dense > 0, want), name
        seen += 1
    assert seen == 3


def test_oracle_reproduces_mode_goldens():
    """tests/golden/modes_w32.npz (the four `_gan_loss` modes incl. the wgan-gp penalty and the critic's gradients, the
    weights after two TF-Adam iterations of the training loop, model 101 and the stage-II losses; `make_golden.py
    --modes`) re-derived from seeds by today's oracle: the fixture the GPU box compares against cannot drift from the
    restatement it was made with."""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = np.load(os.path.join(here, "golden", "modes_w32.npz"))
    got = mg.mode_outputs()
    assert set(got) == set(gold.files)
    for k in gold.files:
        ref = gold[k]
        scale = max(float(np.abs(ref).max()), 1e-12)
        assert np.abs(np.asarray(got[k], dtype=np.float64) - ref).max() <= 1e-7 * scale, k
    # sanity of what the fixture pins: the penalty is part of the wgan-gp critic loss, lsgan differs from dcgan
    assert float(gold["loss/wgan-gp/penalty"]) > 0
    assert abs(float(gold["loss/lsgan/d_loss"]) - float(gold["loss/dcgan/d_loss"])) > 1e-3


# ---- generated shapes: oracle/ops.py (torch, differentiable) against oracle/naive.py (numpy loops) ------------------------
# with no reference vectors, each TF-1.4 op is pinned by two independently written restatements agreeing
# on hypothesis-drawn shapes -- asymmetric SAME pads at both strides and all three kernel sizes, degenerate sizes
# (1-pixel images, kernels larger than the image), ragged crops and out-of-image boxes.
from hypothesis import given, settings, strategies as st   # noqa: E402

_conv_shapes = st.tuples(st.integers(1, 2), st.integers(1, 9), st.integers(1, 9), st.integers(1, 4), st.integers(1, 4),
                         st.sampled_from([1, 3, 5]), st.sampled_from([1, 2]))


@settings(max_examples=40, deadline=None)
@given(_conv_shapes, st.integers(0, 2 ** 31 - 1))
def test_generated_conv_shapes_fwd_dgrad_wgrad(shape, seed):
    N, H, W, C, K, k, s = shape
    x, w, b = rnd((N, H, W, C), seed), rnd((k, k, C, K), seed + 1), rnd((K,), seed + 2)
    xt, wt = T(x).requires_grad_(True), T(w).requires_grad_(True)
    y = O.conv2d_same(xt, wt, T(b), s)
    ref = naive.conv2d_same(x, w, b, s)
    assert y.shape == ref.shape == (N, -(-H // s), -(-W // s), K)
    assert np.abs(y.detach().numpy() - ref).max() < 1e-12
    dy = rnd(ref.shape, seed + 3)
    y.backward(T(dy))
    assert np.abs(xt.grad.numpy() - naive.conv2d_same_dgrad(dy, w, x.shape, s)).max() < 1e-12
    assert np.abs(wt.grad.numpy() - naive.conv2d_same_wgrad(x, dy, w.shape, s)).max() < 1e-12


@settings(max_examples=25, deadline=None)
@given(st.tuples(st.integers(1, 3), st.integers(1, 6), st.integers(1, 6), st.integers(1, 5)), st.integers(0, 2 ** 31 - 1))
def test_generated_norm_and_upsample_shapes(shape, seed):
    x, sc, of = rnd(shape, seed), rnd((shape[3],), seed + 1) + 1.5, rnd((shape[3],), seed + 2)
    assert np.abs(O.batchnorm_train(T(x), T(sc), T(of)).numpy() - naive.batchnorm_train(x, sc, of)).max() < 1e-10
    assert np.abs(O.layernorm(T(x), T(sc), T(of)).numpy() - naive.layernorm(x, sc, of)).max() < 1e-10
    assert np.array_equal(O.upsample2x(T(x)).numpy(), naive.upsample2x(x))


@settings(max_examples=25, deadline=None)
@given(st.tuples(st.integers(1, 2), st.integers(2, 9), st.integers(2, 9), st.integers(1, 3)), st.integers(1, 5), st.integers(1, 6),
       st.integers(1, 6), st.integers(0, 2 ** 31 - 1))
def test_generated_crop_and_resize_shapes_incl_grad(shape, nbox, ch, cw, seed):
    N, H, W, C = shape
    rng = np.random.default_rng(seed)
    img = rnd(shape, seed)
    y1, x1 = rng.uniform(-0.3, 0.9, nbox), rng.uniform(-0.3, 0.9, nbox)
    boxes = np.stack([y1, x1, y1 + rng.uniform(0.0, 0.8, nbox), x1 + rng.uniform(0.0, 0.8, nbox)], axis=1)   # some reach outside [0,1]
    ind = rng.integers(0, N, nbox)
    it = T(img).requires_grad_(True)
    out = O.crop_and_resize(it, T(boxes), torch.tensor(ind), ch, cw)
    ref = naive.crop_and_resize(img, boxes, ind, ch, cw)
    assert np.abs(out.detach().numpy() - ref).max() < 1e-12
    dout = rnd(ref.shape, seed + 1)
    out.backward(T(dout))
    assert np.abs(it.grad.numpy() - naive.crop_and_resize_grad_image(dout, boxes, ind, img.shape)).max() < 1e-12


def test_oracle_reproduces_stage2_df256_golden():
    """tests/golden/stage2_df256_w16.npz (`make_golden.py --stage2-256`: the DeepFashion stage-II models 102 / 103 / 104 of
    trainer_256.py:266-700) is re-derived from seeds by today's oracle."""
    import importlib.util
    import os
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = np.load(os.path.join(here, "golden", "stage2_df256_w16.npz"))
    got = mg.stage2_256_outputs()
    assert set(got.keys()) == set(want.keys())
    for k in want.keys():
        scale = max(float(np.abs(want[k]).max()), 1e-12)
        assert float(np.abs(np.asarray(got[k], dtype=np.float64) - want[k]).max()) <= 1e-9 * scale, k


def test_batchnorm_side_branches_against_numpy_loops():
    """The oracle's restatements of the reference Batchnorm's non-training branches (batchnorm.py:31-37, 57-68, 74-87) against
    explicit numpy loops over channels / items."""
    from oracle import ops as O
    rng = np.random.RandomState(3)
    x = rng.randn(3, 5, 4, 6)
    sc, of, mm, mv = rng.rand(6) + 0.5, rng.randn(6), rng.randn(6), rng.rand(6) + 0.2
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))
    got = O.batchnorm_inference_blend(t(x), t(sc), t(of), t(mm), t(mv)).numpy()
    ref = np.empty_like(x)
    for n in range(3):
        for c in range(6):
            m = x[n, :, :, c].mean() / 3. + (2. / 3.) * mm[c]
            v = x[n, :, :, c].var() / 3. + (2. / 3.) * mv[c]
            ref[n, :, :, c] = (x[n, :, :, c] - m) / np.sqrt(v + 1e-5) * sc[c] + of[c]
    assert np.abs(got - ref).max() < 1e-12
    nm, nv = O.batchnorm_moving_update(t(x), t(mm), t(mv), 4)
    for c in range(6):
        col = x[..., c].ravel()
        assert abs(float(nm[c]) - (0.8 * mm[c] + 0.2 * col.mean())) < 1e-12
        assert abs(float(nv[c]) - (0.8 * mv[c] + 0.2 * col.var(ddof=1))) < 1e-12
    y = rng.randn(4, 6, 5)                                  # [N, C, L], moments over the batch only: per-(c, l) parameters
    s2, o2 = rng.rand(1, 6, 5) + 0.5, rng.randn(1, 6, 5)
    got = O.batchnorm_unfused(t(y), [0], t(s2), t(o2)).numpy()
    ref = (y - y.mean(0, keepdims=True)) / np.sqrt(y.var(0, keepdims=True) + 1e-5) * s2 + o2
    assert np.abs(got - ref).max() < 1e-12


def test_tflib_batchnorm_inference_and_unfused_branches_match_the_oracle():
    """tflib.ops.batchnorm.Batchnorm off the hot path (every reference call site passes is_training=None): the inference blend
    (is_training=False) and the unfused branch are tensor plumbing on the caller's device -- checked here on the host against the
    oracle, incl. the parameter shapes the reference creates."""
    import dpig_amd.tflib as lib
    import dpig_amd.tflib.ops  # noqa
    from oracle import ops as O
    lib.delete_all_params()
    lib.set_device("cpu")
    try:
        g = torch.Generator().manual_seed(7)
        x = torch.randn(3, 6, 5, 4, generator=g)                             # NCHW
        lib.param('B.moving_mean', torch.randn(6, generator=g).numpy(), trainable=False)
        lib.param('B.moving_variance', (torch.rand(6, generator=g) + 0.2).numpy(), trainable=False)
        lib.param('B.scale', (torch.rand(6, generator=g) + 0.5).numpy())
        lib.param('B.offset', torch.randn(6, generator=g).numpy())
        y = lib.ops.batchnorm.Batchnorm('B', [0, 2, 3], x, is_training=False, stats_iter=0)
        d = lambda n: lib.param(n).detach().double()
        ref = O.batchnorm_inference_blend(x.double().permute(0, 2, 3, 1), d('B.scale'), d('B.offset'), d('B.moving_mean'), d('B.moving_variance'))
        assert tuple(y.shape) == (3, 6, 5, 4)
        assert (y.permute(0, 2, 3, 1).double() - ref).abs().max() < 1e-5
        assert torch.equal(lib.param('B.moving_mean'), d('B.moving_mean').float())          # the inference branch never updates
        x3 = torch.randn(4, 6, 5, generator=g)
        y3 = lib.ops.batchnorm.Batchnorm('B3', [0, 2], x3, is_training=False, stats_iter=0)  # [N, C, L] through the same branch
        ref3 = O.batchnorm_inference_blend(x3.double().permute(0, 2, 1).unsqueeze(2), d('B3.scale'), d('B3.offset'), d('B3.moving_mean'),
                                           d('B3.moving_variance'))
        assert tuple(y3.shape) == (4, 6, 5) and (y3.permute(0, 2, 1).unsqueeze(2).double() - ref3).abs().max() < 1e-5
        z = lib.ops.batchnorm.Batchnorm('U', [0], x3, fused=False)                           # unfused: moments over the batch only
        assert tuple(lib.param('U.scale').shape) == (1, 6, 5) and tuple(lib.param('U.offset').shape) == (1, 6, 5)
        refz = O.batchnorm_unfused(x3.double(), [0], d('U.scale'), d('U.offset'))
        assert (z.double() - refz).abs().max() < 1e-5
        z2 = lib.ops.batchnorm.Batchnorm('U2', [1, 2], x3, fused=False)                      # no batch axis: shared parameters (the warning)
        assert tuple(lib.param('U2.scale').shape) == (1, 1, 1)
        assert (z2.double() - O.batchnorm_unfused(x3.double(), [1, 2], d('U2.scale'), d('U2.offset'))).abs().max() < 1e-5
    finally:
        lib.delete_all_params()
        lib.set_device(None)



def test_gp_sweeps_equal_double_backward():
    """oracle/gp_sweeps.py (the gradient penalty as three explicit sweeps, SURVEY Appendix E -- the form csrc/dpig_gp.hip evaluates) against
    torch's double backward of `oracle.models.gradient_penalty` (trainer.py:222-236 over wgan_gp.py:407-440): value and every parameter
    gradient to fp64 round-off, at Market 128x64 and at 256x256 (8 logit rows per image).  With bf16 stores the same sweeps give
    the emulation the GPU's 'bf16' storage mode is held against; how far bf16 storage alone moves the result is printed."""
    import torch
    from oracle import gp_sweeps as GS
    from oracle import models as OM
    for shape, dim in (((2, 128, 64, 3), 8), ((1, 256, 256, 3), 4)):
        g = torch.Generator().manual_seed(21)
        B = shape[0]
        P = OM.ParamStore(seed=23)
        real = torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1
        fake = torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1
        alpha = torch.rand(B, generator=g, dtype=torch.float64)
        D_o = lambda t: OM.dcgan_discriminator(P, t, "wgan-gp", dim=dim)                  # noqa: E731
        D_o(real[:1])
        names = OM.d_var_names(P)
        with torch.no_grad():
            for n in names:
                if not n.endswith(("Filters", "Output.W")):
                    P.p[n].add_(0.2 * (torch.rand(P.p[n].shape, generator=g, dtype=torch.float64) - 0.5))
        gp_ref = OM.gradient_penalty(D_o, real, fake, alpha, 10.0)
        refs = dict(zip(names, torch.autograd.grad(gp_ref, [P.p[n] for n in names], allow_unused=True)))
        pen, grads = GS.gp_sweeps(P.p, real, fake, alpha, 10.0, dim=dim)
        assert abs(pen.item() - gp_ref.item()) < 1e-11 * abs(gp_ref.item())
        for n in names:
            if n.endswith("Output.b"):
                continue
            if refs[n] is None:
                assert float(grads[n].abs().max()) == 0.0
                continue
            assert (grads[n] - refs[n]).abs().max().item() < 1e-10 * max(refs[n].abs().max().item(), 1e-30), n
        pen16, g16 = GS.gp_sweeps(P.p, real, fake, alpha, 10.0, dim=dim, store=GS.bf16_round)
        worst = max((g16[n] - refs[n]).norm().item() / refs[n].norm().item() for n in names if refs[n] is not None and not n.endswith("Output.b"))
        print("bf16 storage of the sweeps' tensors alone moves the penalty by %.1e (rel) and a parameter gradient by up to %.1e (rel L2)"
              % (abs(pen16.item() - gp_ref.item()) / abs(gp_ref.item()), worst))
        assert worst < 0.2


# ---- the sampled forms used at BASELINE full sizes (tests/test_fullsize_gpu.py::test_sampled_oracle_*) ---------------------
@pytest.mark.parametrize("shape", [(2, 8, 6, 3, 5, 3, 1, False), (2, 8, 6, 4, 5, 3, 2, False), (1, 9, 7, 3, 4, 5, 2, False),
                                   (2, 5, 5, 2, 3, 1, 1, False), (1, 7, 9, 3, 2, 3, 2, False), (2, 4, 3, 5, 4, 1, 1, True),
                                   (3, 2, 2, 4, 4, 3, 1, False), (2, 12, 10, 3, 4, 5, 1, False)])
def test_sampled_conv_equals_dense(shape):
    """conv2d_same_sampled / _dgrad_sampled / _wgrad_sampled are conv2d_same and its autograd gradients restricted to the requested
    elements -- EVERY element here (all positions, all taps, all channels), so a wrong pad / stride / border rule cannot hide."""
    N, H, W, C, K, k, s, up = shape
    x = T(rnd((N, H, W, C), 1)).requires_grad_(True)
    w = T(rnd((k, k, C, K), 2)).requires_grad_(True)
    b = T(rnd((K,), 3))
    y = O.conv2d_same(O.upsample2x(x) if up else x, w, b, s)
    dy = T(rnd(tuple(y.shape), 4))
    dx, dw = torch.autograd.grad(y, [x, w], dy)
    Ho, Wo = y.shape[1], y.shape[2]
    n, oy, ox = [t.reshape(-1) for t in torch.meshgrid(torch.arange(N), torch.arange(Ho), torch.arange(Wo), indexing="ij")]
    got = O.conv2d_same_sampled(x.detach(), w.detach(), b, s, n, oy, ox, upsample2x=up)
    np.testing.assert_allclose(got.numpy(), y.detach().reshape(-1, K).numpy(), rtol=1e-12, atol=1e-13)
    n, iy, ix = [t.reshape(-1) for t in torch.meshgrid(torch.arange(N), torch.arange(H), torch.arange(W), indexing="ij")]
    got = O.conv2d_same_dgrad_sampled(dy, w.detach(), (N, H, W, C), s, n, iy, ix, upsample2x=up)
    np.testing.assert_allclose(got.numpy(), dx.reshape(-1, C).numpy(), rtol=1e-12, atol=1e-13)
    taps = [(r, c) for r in range(k) for c in range(k)]
    got = O.conv2d_same_wgrad_sampled(x.detach(), dy, (k, k, C, K), s, taps, torch.arange(C), torch.arange(K), upsample2x=up)
    np.testing.assert_allclose(got.reshape(k, k, C, K).numpy(), dw.numpy(), rtol=1e-12, atol=1e-13)
    # a strict subset in scrambled order picks the same numbers
    sel = torch.tensor([len(n) - 1, 0, len(n) // 2])
    sub = O.conv2d_same_dgrad_sampled(dy, w.detach(), (N, H, W, C), s, n[sel], iy[sel], ix[sel], upsample2x=up)
    np.testing.assert_allclose(sub.numpy(), dx.reshape(-1, C)[sel].numpy(), rtol=1e-12, atol=1e-13)
    ci, co = torch.tensor([C - 1, 0]), torch.tensor([0, K - 1, 1])
    sub = O.conv2d_same_wgrad_sampled(x.detach(), dy, (k, k, C, K), s, taps[-1:], ci, co, upsample2x=up)
    np.testing.assert_allclose(sub[0].numpy(), dw[taps[-1][0], taps[-1][1]][ci][:, co].numpy(), rtol=1e-12, atol=1e-13)
