"""Parity of the remaining model variants on the hot path against the CPU oracle (small widths):
stage-II embedding GAN pieces (GaussianFCRes mapper + FC critic, wgan mode, RMSProp + clipping;
trainer.py:715-868), the DeepFashion 256x256 stage-I graph (trainer_256.py:31-88: deeper ROI encoder,
D on the concatenated pair with joint BatchNorm statistics, 8 logit rows per image), and the
LayerNorm discriminator that MODE='wgan-gp' selects (wgan_gp.py:34-40)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(got, ref):
    ref = ref.detach().double()
    return (got.detach().double().cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


def _load(P, dev):
    import dpig_amd.tflib as lib
    from dpig_amd import slim
    lib.delete_all_params()
    slim.reset_scopes()
    lib.set_device(dev)
    for n, v in P.state_numpy().items():
        lib.param(n, v, trainable=P.trainable[n])


def test_stage2_mapper_critic_wgan(dev):
    import dpig_amd.tflib as lib
    from dpig_amd import synthetic
    from dpig_amd.trainer import Config
    from dpig_amd.trainer_stage2 import DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI
    from oracle import models as OM
    B, HID = 4, 8
    g = torch.Generator().manual_seed(0)
    P = OM.ParamStore(seed=5)
    batch_np = synthetic.make_batch(B, seed=9)
    ob = OM.batch_to_torch(batch_np)
    with torch.no_grad():
        embs = OM.encoder_fgbg(P, ob["x"], ob["mask_r6"], ob["part_bbox"], ob["part_vis"], 7, 32, 5, HID)
    real = {"Fg": embs[:, :224], "Bg": embs[:, 224:]}
    z = {"Fg": torch.randn(B, 224, generator=g, dtype=torch.float64) * 0.2,
         "Bg": torch.randn(B, 128, generator=g, dtype=torch.float64) * 0.2}
    ref = {}
    for side, hid in (("Fg", 512), ("Bg", 256)):
        ref[side] = OM.stage2_losses(P, real[side], z[side], side, hid)
    _load(P, dev)
    tr = DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI(Config(batch_size=B, conv_hidden_num=HID, g_lr=1e-3, d_lr=1e-3), dev)
    gb = synthetic.to_device(batch_np, dev)
    tr.init_net(gb)
    assert set(lib._params.keys()) == set(P.p.keys())
    fg, bg, _ = tr.encode(gb)
    assert _rel(torch.cat([fg, bg], 1), embs) < 1e-4
    for side, key in (("fg", "Fg"), ("bg", "Bg")):
        g_ref, d_ref, fake_ref = ref[key]
        gnames = [n for n in P.p if n.startswith("Gaussian_FC_%s/" % key)]
        dnames = [n for n in P.p if ("%s_FCDis_" % key) in n]
        gg = dict(zip(gnames, torch.autograd.grad(g_ref, [P.p[n] for n in gnames], retain_graph=True)))
        dg = dict(zip(dnames, torch.autograd.grad(d_ref, [P.p[n] for n in dnames])))
        zz = z[key].float().to(dev)
        d_loss = tr.d_optim_embs(side, gb, z=zz)               # critic first: mapper weights still the oracle's
        assert abs(d_loss.item() - d_ref.item()) < 1e-4 * max(abs(d_ref.item()), 1e-3)
        for n in dnames:
            assert _rel(lib._params[n]._dpig_grad, dg[n]) < 2e-3, n
        # RMSProp (rms slot = ones) + clip to +-0.01 (trainer.py:119-128)
        for n in dnames:
            p_new, _, _ = OM.tf_rmsprop_step(P.p[n].detach(), dg[n], torch.ones_like(dg[n]), torch.zeros_like(dg[n]), 1e-3)
            assert (lib._params[n].detach().double().cpu() - p_new.clamp(-0.01, 0.01)).abs().max().item() < 2e-5, n
        with torch.no_grad():                                   # restore the critic for the mapper check
            for n in dnames:
                lib._params[n].copy_(P.p[n].to(torch.float32))
        g_loss = tr.g_optim_embs(side, z=zz)
        assert abs(g_loss.item() - g_ref.item()) < 1e-4 * max(abs(g_ref.item()), 1e-3)
        for n in gnames:
            assert _rel(lib._params[n]._dpig_grad, gg[n]) < 2e-3, n
    out = tr.train_step(gb)                # step 0: no mapper update, 5 clipped critic updates per side
    assert "g_loss_embs_fg" not in out and "d_loss_embs_bg" in out
    out = tr.train_step(gb)
    assert "g_loss_embs_fg" in out and "g_loss_embs_bg" in out
    assert tr.opts["fg"][1].t == 1 + 5 + 5 and tr.opts["fg"][0].t == 1 + 1
    for _, df in tr.flats.values():
        assert float(df.flat.abs().max()) <= 0.01 + 1e-9
    lib.delete_all_params()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_stage2_hipgraph_replay_matches_eager(dev, dtype):
    """Model 3 (trainer.py:715-868) with its optimizer ops replayed as hipGraphs (`enable_graphs`: one graph per (side, op), the loop order
    of trainer.py:821-845) against the same trainer launched eagerly: same weights, same device-generator state, three steps (the first
    without mapper updates, a sequence of distinct batches for the critic updates) -- losses and every mapper / critic weight identical
    (the captured samplers draw from the generator at the offsets eager execution uses); enabling graphs does not move the weights."""
    import dpig_amd.hip_ops as H
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic
    from dpig_amd.trainer import Config
    from dpig_amd.trainer_stage2 import DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI
    B, HID = 4, 16
    batches = [synthetic.to_device(synthetic.make_batch(B, seed=40 + i), dev) for i in range(3)]
    res = {}
    try:
        for mode in ("eager", "graph"):
            lib.delete_all_params(); slim.reset_scopes(); lib.set_device(dev)
            import numpy as np
            np.random.seed(3)
            tr = DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI(Config(batch_size=B, conv_hidden_num=HID, g_lr=1e-3, d_lr=1e-3,
                                                                     compute_dtype=dtype), dev)
            tr.init_net(batches[0])
            w0 = [f.flat.clone() for pair in tr.flats.values() for f in pair]
            if mode == "graph":
                tr.enable_graphs(batches[0])
                assert tr._graphs is not None
                for a, f in zip(w0, [f for pair in tr.flats.values() for f in pair]):
                    assert torch.equal(a, f.flat), "enable_graphs moved the weights"
            torch.cuda.manual_seed(1234)
            outs = []
            for step in range(3):
                o = tr.train_step(batches)
                outs.append({k: float(v) for k, v in o.items()})
            res[mode] = (outs, [f.flat.clone() for pair in tr.flats.values() for f in pair],
                         [(o.t) for pair in tr.opts.values() for o in pair])
        assert res["eager"][2] == res["graph"][2]
        for oe, og in zip(res["eager"][0], res["graph"][0]):
            assert oe.keys() == og.keys()
            for k in oe:
                assert oe[k] == og[k], (k, oe[k], og[k])
        for a, b in zip(res["eager"][1], res["graph"][1]):
            assert torch.equal(a, b)
        assert float((res["graph"][1][0] - w0[0]).abs().max()) > 0
    finally:
        H.set_compute("f32")
        lib.delete_all_params(); slim.reset_scopes()


def test_stage1_256_graph(dev):
    """Model 101 at 256x256, width 8: activations, the 8-rows-per-image logits (F8), losses, one step."""
    import dpig_amd.tflib as lib
    from dpig_amd import synthetic
    from dpig_amd.trainer import Config
    from dpig_amd.trainer_256 import DPIG_Encoder_GAN_BodyROI_256
    from oracle import models as OM
    B, HID, ZN = 2, 8, 8
    batch_np = synthetic.make_batch(B, img_H=256, img_W=256, seed=4)
    ob = OM.batch_to_torch(batch_np)
    P = OM.ParamStore(seed=6)
    with torch.no_grad():
        ref = OM.stage1_256_forward(P, ob, HID, ZN, 6)
    _load(P, dev)
    cfg = Config(batch_size=B, img_H=256, img_W=256, conv_hidden_num=HID, z_num=ZN)
    assert cfg.repeat_num == 6
    tr = DPIG_Encoder_GAN_BodyROI_256(cfg, dev)
    gb = synthetic.to_device(batch_np, dev)
    tr.init_net(gb)
    assert set(lib._params.keys()) == set(P.p.keys())
    with torch.no_grad():
        embs, _ = tr.encode(gb)
        G, _ = tr.generate(embs, gb["pose"])
        D_pos, D_neg = tr.disc_pair(gb["x"], G)
    assert tuple(embs.shape) == (B, 224) and tuple(D_pos.shape) == (8 * B,) and tuple(D_neg.shape) == (8 * B,)
    assert _rel(embs, ref["embs"]) < 1e-4 and _rel(G, ref["G"]) < 1e-4
    assert _rel(torch.cat([D_pos, D_neg]), ref["D_z"]) < 1e-3
    o = tr.train_step(gb, gb)              # step 0: d_optim only
    assert abs(o["d_loss"].item() - ref["d_loss"].item()) < 1e-3 * abs(ref["d_loss"].item())
    o = tr.train_step(gb, gb)
    assert torch.isfinite(o["g_loss"]) and torch.isfinite(o["d_loss"])
    lib.delete_all_params()


def test_layernorm_discriminator_wgan_gp_mode(dev):
    """MODE='wgan-gp' swaps the discriminator's BatchNorm for LayerNorm (wgan_gp.py:34-40)."""
    import dpig_amd.tflib as lib
    from dpig_amd.wgan_gp import WGAN_GP
    from oracle import models as OM
    g = torch.Generator().manual_seed(1)
    x = torch.rand(3, 128, 64, 3, generator=g, dtype=torch.float64) * 2 - 1
    P = OM.ParamStore(seed=8)
    xr = x.clone().requires_grad_(True)
    ref = OM.dcgan_discriminator(P, xr, "wgan-gp")
    names = OM.d_var_names(P)
    grads = torch.autograd.grad(ref.sum(), [xr] + [P.p[n] for n in names])
    _load(P, dev)
    wg = WGAN_GP(MODE='wgan-gp', BATCH_SIZE=3)
    xg = x.float().to(dev).requires_grad_(True)
    out = wg.DCGANDiscriminator(xg.permute(0, 3, 1, 2), input_dim=3)
    assert 'Discriminator.BN2.moving_mean' not in lib._params          # LayerNorm creates no moving stats
    assert _rel(out, ref) < 1e-4
    out.sum().backward()
    assert _rel(xg.grad, grads[0]) < 2e-3
    for n, gr in zip(names, grads[1:]):
        assert _rel(lib._params[n].grad, gr) < 2e-3, n
    lib.delete_all_params()


def test_second_order_layernorm_kernel(dev):
    """dpig_ln_bwd2 against torch double backward of the oracle LayerNorm."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    g = torch.Generator().manual_seed(2)
    shape = (3, 6, 5, 12)
    x = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 0.4).requires_grad_(True)
    sc = (torch.rand(12, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    of = torch.zeros(12, dtype=torch.float64)
    dy = (torch.rand(shape, generator=g, dtype=torch.float64) - 0.5).requires_grad_(True)
    u = torch.rand(shape, generator=g, dtype=torch.float64) - 0.5
    y = O.leaky_relu(O.layernorm(x, sc, of), 0.2)
    (dx,) = torch.autograd.grad(y, x, dy, create_graph=True)
    d_dy, d_x, d_sc = torch.autograd.grad((dx * u).sum(), [dy, x, sc])
    xg = x.detach().float().to(dev)
    yg, mean, rstd = H.ln_fwd(xg, sc.detach().float().to(dev), of.float().to(dev), 1e-5, 2, 0.2)
    g_dy, g_x, g_sc = H.ln_bwd2(u.float().to(dev), dy.detach().float().to(dev), xg, yg, sc.detach().float().to(dev),
                                mean, rstd, 2, 0.2)
    assert _rel(g_dy, d_dy) < 5e-5 and _rel(g_x, d_x) < 5e-5 and _rel(g_sc, d_sc) < 5e-5


@pytest.mark.parametrize("kind", ["image_ln_critic", "fc_critic"])
def test_wgan_gp_gradient_penalty_double_backward(dev, kind):
    """d(LAMBDA*GP)/d(theta_D) through conv5x5s2 / LayerNorm / LeakyReLU / Linear (image critic) and through
    the FC critic, against torch's double backward of the oracle (same alpha)."""
    import dpig_amd.tflib as lib
    from dpig_amd.trainer import gradient_penalty
    from dpig_amd.wgan_gp import WGAN_GP
    from oracle import models as OM
    g = torch.Generator().manual_seed(3)
    P = OM.ParamStore(seed=9)
    if kind == "image_ln_critic":
        B = 2
        real = torch.rand(B, 128, 64, 3, generator=g, dtype=torch.float64) * 2 - 1
        fake = torch.rand(B, 128, 64, 3, generator=g, dtype=torch.float64) * 2 - 1
        alpha = torch.rand(B, 1, 1, 1, generator=g, dtype=torch.float64)
        D_o = lambda t: OM.dcgan_discriminator(P, t, "wgan-gp")                     # noqa: E731
    else:
        B = 6
        real = torch.rand(B, 224, generator=g, dtype=torch.float64) - 0.5
        fake = torch.rand(B, 224, generator=g, dtype=torch.float64) - 0.5
        alpha = torch.rand(B, 1, generator=g, dtype=torch.float64)
        D_o = lambda t: OM.fc_discriminator(P, t, 224, name="Fg_FCDis_")            # noqa: E731
    xh = (real + alpha * (fake - real)).requires_grad_(True)
    (gr,) = torch.autograd.grad(D_o(xh).sum(), xh, create_graph=True)
    gp_ref = 10.0 * ((gr.reshape(B, -1).pow(2).sum(1).sqrt() - 1) ** 2).mean()
    names = OM.d_var_names(P)
    refs = dict(zip(names, torch.autograd.grad(gp_ref, [P.p[n] for n in names], allow_unused=True)))
    _load(P, dev)
    wg = WGAN_GP(MODE='wgan-gp', BATCH_SIZE=B)
    if kind == "image_ln_critic":
        D_h = lambda t: wg.DCGANDiscriminator(t.permute(0, 3, 1, 2), input_dim=3)   # noqa: E731
    else:
        D_h = lambda t: wg.FCDiscriminator(t, input_dim=224, name="Fg_FCDis_")      # noqa: E731
    gp = gradient_penalty(D_h, real.float().to(dev), fake.float().to(dev), 10.0, alpha.float().to(dev))
    assert abs(gp.item() - gp_ref.item()) < 1e-4 * abs(gp_ref.item())
    gp.backward()
    for n in names:
        if refs[n] is None:
            continue
        got = lib._params[n].grad
        assert got is not None, n
        scale = max(refs[n].abs().max().item(), 1e-7)
        assert (got.double().cpu() - refs[n]).abs().max().item() < 5e-3 * scale, n
    lib.delete_all_params()


@pytest.mark.parametrize("shape,dim", [((2, 128, 64, 3), 64), ((1, 256, 256, 3), 16)])
def test_fused_gp_double_backward_entry_point(dev, shape, dim):
    """`dpig_gp_double_backward` (one C-ABI call: forward + input gradient + penalty + the analytic second-order sweep,
    trainer.py:222-236 over wgan_gp.py:407-440) against `oracle.models.gradient_penalty` differentiated by torch in fp64:
    penalty value and every parameter gradient within 2e-5 of the tensor's max |ref|; value-only mode; beta accumulation;
    256x256 inputs produce the reference's 8 logit rows per image."""
    from dpig_amd import hip_ops as H
    from oracle import models as OM
    g = torch.Generator().manual_seed(5)
    B = shape[0]
    P = OM.ParamStore(seed=13)
    real = torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1
    fake = torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1
    alpha = torch.rand(B, generator=g, dtype=torch.float64)
    D_o = lambda t: OM.dcgan_discriminator(P, t, "wgan-gp", dim=dim)                  # noqa: E731
    D_o(real[:1])                                    # creates the variables; then move biases / scales / offsets off their
    names = [n for n in OM.d_var_names(P)]           # zero / one initial values so every term of the sweep is exercised
    with torch.no_grad():
        for n in names:
            if not n.endswith(("Filters", "Output.W")):
                P.p[n].add_(0.2 * (torch.rand(P.p[n].shape, generator=g, dtype=torch.float64) - 0.5))
    gp_ref = OM.gradient_penalty(D_o, real, fake, alpha, 10.0)
    refs = dict(zip(names, torch.autograd.grad(gp_ref, [P.p[n] for n in names], allow_unused=True)))
    params = {n: P.p[n].detach().float().to(dev).contiguous() for n in names}
    pen, slopes, grads = H.gp_double_backward(params, real.float().to(dev), fake.float().to(dev), alpha.float().to(dev), 10.0,
                                              dim=dim, grads=True)
    assert abs(pen.item() - gp_ref.item()) < 2e-5 * abs(gp_ref.item())
    worst = 0.0
    for n in names:
        if n.endswith("Output.b"):
            assert refs[n] is None or float(refs[n].abs().max()) == 0.0
            continue
        ref = refs[n]
        if ref is None:                               # LayerNorm 4's offset only moves LeakyReLU masks: no gradient
            assert n.endswith("BN4.offset") and float(grads[n].abs().max()) == 0.0
            continue
        scale = max(ref.abs().max().item(), 1e-12)
        err = (grads[n].double().cpu() - ref).abs().max().item() / scale
        worst = max(worst, err)
        assert err < 2e-5, (n, err)
    print("fused gp double backward %s: worst per-tensor error %.2e of max|ref|" % (shape, worst))
    # value only
    pen2, slopes2, none = H.gp_double_backward(params, real.float().to(dev), fake.float().to(dev), alpha.float().to(dev), 10.0,
                                               dim=dim)
    assert none is None and torch.equal(pen2, pen) and torch.equal(slopes2, slopes)
    # grads = beta * grads + ...
    acc = {n: torch.ones_like(t) for n, t in params.items()}
    H.gp_double_backward(params, real.float().to(dev), fake.float().to(dev), alpha.float().to(dev), 10.0, dim=dim, grads=acc,
                         beta=0.5)
    for n in names:
        if n.endswith("Output.b"):
            continue
        want = grads[n] + 0.5
        assert (acc[n] - want).abs().max().item() <= 1e-6 * max(want.abs().max().item(), 1.0), n


@pytest.mark.parametrize("shape,dim", [((2, 128, 64, 3), 64), ((2, 256, 256, 3), 64), ((1, 256, 256, 3), 32)])
def test_fused_gp_double_backward_bf16_storage(dev, shape, dim):
    """`dpig_gp_double_backward` with DPIG_COMPUTE_BF16_STORE (BASELINE configs[4]: "bf16 ... fused GP double-backward"): the critic's
    activations and their adjoints are bf16 tensors inside the call, convs 2-4 read bf16 filter shadows, everything accumulates in fp32.

    (1) LINK BY LINK at the per-kernel bounds.  The call writes no workspace slot twice, so afterwards every tensor of the three sweeps
    can be read back (dpig_gp_double_backward_slot); `oracle.gp_sweeps.gp_chain_links` recomputes each one in fp64 from the tensors the
    library itself stored as that link's inputs (trainer.py:222-236 over wgan_gp.py:407-440).  A bf16 result must
    equal the fp64 value to one rounding (|err| <= 2^-8 |ref| + 2e-5 max|ref| per element: `_close_bf16` of tests/test_conv_bf16_q_gpu.py);
    an fp32 result -- the input gradient, the penalty, its seed and EVERY critic parameter gradient -- holds the exact path's 2e-5 of
    max|ref| (their operands are bf16 values, whose products are exact in fp32).
    (2) END TO END against the fp64 oracle on the operands the kernels see (filters 2-4 bf16-exact): the penalty value within 2^-8; the
    parameter gradients are reported, and bounded loosely: a single rounding flip of a forward activation next to a LeakyReLU's zero
    crossing flips that unit's mask in all three sweeps, so the end-to-end distance measures bf16 STORAGE (oracle/gp_sweeps.py with bf16
    stores moves the same gradients by the same 1-5 %: tests/test_oracle.py::test_gp_sweeps_equal_double_backward), not the kernels --
    which is why (1) is the parity test.
    (3) The fused call agrees with the taped second-level-autograd path of 'bf16' mode (same kernels, other orchestration), value-only
    mode, beta accumulation."""
    import dpig_amd.tflib as lib
    from dpig_amd import hip_ops as H
    from dpig_amd.trainer import gradient_penalty
    from dpig_amd.wgan_gp import WGAN_GP
    from oracle import gp_sweeps as GS
    from oracle import models as OM
    g = torch.Generator().manual_seed(7)
    B = shape[0]
    P = OM.ParamStore(seed=17)
    real = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1).float().double()
    fake = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1).float().double()
    alpha = torch.rand(B, generator=g, dtype=torch.float64).float().double()
    D_o = lambda t: OM.dcgan_discriminator(P, t, "wgan-gp", dim=dim)                  # noqa: E731
    D_o(real[:1])
    names = [n for n in OM.d_var_names(P)]
    with torch.no_grad():
        for n in names:
            if not n.endswith(("Filters", "Output.W")):
                P.p[n].add_(0.2 * (torch.rand(P.p[n].shape, generator=g, dtype=torch.float64) - 0.5))
            P.p[n].copy_(P.p[n].float().double())                       # fp32 masters ...
            if n.endswith("Filters") and not n.endswith("Discriminator.1.Filters"):
                P.p[n].copy_(P.p[n].float().to(torch.bfloat16).double())  # ... whose bf16 shadows are exact (what convs 2-4 multiply)
    params = {n: P.p[n].detach().float().to(dev).contiguous() for n in names}
    rd, fd, ad = real.float().to(dev), fake.float().to(dev), alpha.float().to(dev)
    pen, slopes, grads = H.gp_double_backward(params, rd, fd, ad, 10.0, dim=dim, grads=True, compute=H.COMPUTE_BF16_STORE)
    stored = {k: v.double().cpu() for k, v in H.gp_double_backward_tensors(shape, dim, H.COMPUTE_BF16_STORE, dev).items()}
    assert stored["A1"].abs().max() > 0
    # ---- (1) every link from the library's own stored inputs ---------------------------------------------------------------------------
    worst = {"store": 0.0, "f32": 0.0}
    nlinks = 0
    for name, ref, kind in GS.gp_chain_links(P.p, stored, real, fake, alpha, 10.0, dim=dim):
        got = pen.double().cpu().reshape(()) if name == "penalty" else (stored[name] if name in stored else grads[name].double().cpu())
        ref = ref.detach()
        scale = max(ref.abs().max().item(), 1e-30)
        err = (got - ref).abs()
        if kind == "store":
            excess = (err - (ref.abs() * 2.0 ** -8 + 2e-5 * scale)).max().item()
            assert excess <= 0, "link %s: a stored element is more than one bf16 rounding away (by %.3e, scale %.3e)" % (name, excess, scale)
            worst["store"] = max(worst["store"], (err / (ref.abs() + 2e-5 * scale / 2.0 ** -8)).max().item())
        else:
            assert err.max().item() <= 2e-5 * scale, "link %s: %.3e of max|ref|" % (name, err.max().item() / scale)
            worst["f32"] = max(worst["f32"], err.max().item() / scale)
        nlinks += 1
    assert nlinks == 50
    print("fused gp double backward, bf16 storage %s dim %d: %d links; worst bf16 link %.2f ulp-units of 2^-8, worst fp32 link %.2e of max|ref|"
          % (shape, dim, nlinks, worst["store"] / 2.0 ** -8, worst["f32"]))
    # ---- (2) end to end ------------------------------------------------------------------------------------------------------------------------
    gp_ref = OM.gradient_penalty(D_o, real, fake, alpha, 10.0)
    refs = dict(zip(names, torch.autograd.grad(gp_ref, [P.p[n] for n in names], allow_unused=True)))
    vrel = abs(pen.item() - gp_ref.item()) / abs(gp_ref.item())
    e2e = {n: ((grads[n].double().cpu() - refs[n]).norm() / refs[n].norm()).item() for n in names if refs[n] is not None and not n.endswith("Output.b")}
    print("   end to end vs the fp64 oracle: penalty rel %.2e; parameter gradients rel-L2 %.1e .. %.1e" % (vrel, min(e2e.values()), max(e2e.values())))
    assert vrel < 2.0 ** -8 and max(e2e.values()) < 0.15
    assert float(grads["Discriminator.BN4.offset"].abs().max()) == 0.0
    # ---- (3) value only / accumulation / the taped path ------------------------------------------------------------------------------------------
    pen2, slopes2, none = H.gp_double_backward(params, rd, fd, ad, 10.0, dim=dim, compute=H.COMPUTE_BF16_STORE)
    assert none is None and torch.equal(pen2, pen) and torch.equal(slopes2, slopes)
    acc = {n: torch.ones_like(t) for n, t in params.items()}
    H.gp_double_backward(params, rd, fd, ad, 10.0, dim=dim, grads=acc, beta=0.5, compute=H.COMPUTE_BF16_STORE)
    for n in names:
        if not n.endswith("Output.b"):
            want = grads[n] + 0.5
            assert (acc[n] - want).abs().max().item() <= 1e-6 * max(want.abs().max().item(), 1.0), n
    if dim == 64:
        _load(P, dev)
        try:
            H.set_compute("bf16")
            wg = WGAN_GP(MODE="wgan-gp", BATCH_SIZE=B)
            D_h = lambda t: wg.DCGANDiscriminator(t.permute(0, 3, 1, 2), input_dim=3)   # noqa: E731
            gp = gradient_penalty(D_h, rd, fd, 10.0, ad)
            gp.backward()
            assert abs(gp.item() - pen.item()) < 1e-3 * abs(pen.item())
            for n in names:
                if refs[n] is None or n.endswith("Output.b"):
                    continue
                t = lib._params[n].grad
                assert t is not None, "the taped bf16 path delivers no penalty gradient for %s" % n
                assert ((t - grads[n]).norm() / grads[n].norm()).item() < 5e-3, n
        finally:
            H.set_compute("f32")
            lib.delete_all_params()


def test_stage1_step_in_wgan_gp_mode(dev):
    """The dormant MODE='wgan-gp' branch end to end (trainer.py:222-236, 131-135): LayerNorm critic,
    5 critic iterations per step with the gradient penalty, Adam(beta1=.5, beta2=.9); d_loss matches the
    oracle for the same alpha."""
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg, gan_loss
    from oracle import models as OM
    B, HID, ZN = 2, 8, 8
    batch_np = synthetic.make_batch(B, seed=12)
    ob = OM.batch_to_torch(batch_np)
    P = OM.ParamStore(seed=13)
    with torch.no_grad():
        _, G_o = OM.stage1_forward(P, ob, hidden_num=HID, z_num=ZN)
    g = torch.Generator().manual_seed(14)
    alpha = torch.rand(B, 1, 1, 1, generator=g, dtype=torch.float64)
    d_real = OM.dcgan_discriminator(P, ob["x"], "wgan-gp")
    d_fake = OM.dcgan_discriminator(P, G_o, "wgan-gp")
    xh = (ob["x"] + alpha * (G_o - ob["x"])).requires_grad_(True)
    (gr,) = torch.autograd.grad(OM.dcgan_discriminator(P, xh, "wgan-gp").sum(), xh, create_graph=True)
    d_ref = d_fake.mean() - d_real.mean() + 10.0 * ((gr.reshape(B, -1).pow(2).sum(1).sqrt() - 1) ** 2).mean()
    _load(P, dev)
    tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, conv_hidden_num=HID, z_num=ZN, gan_mode='wgan-gp'), dev)
    gb = synthetic.to_device(batch_np, dev)
    tr.init_net(gb)
    assert set(lib._params.keys()) == set(P.p.keys())
    assert (tr.d_opt.b1, tr.d_opt.b2) == (0.5, 0.9)
    with torch.no_grad():
        embs, _ = tr.encode(gb)
        G, _ = tr.generate(embs, gb["pose"])
    D_pos, D_neg = tr.disc_pair(gb["x"], G)
    _, d_loss = gan_loss(tr.wgan_gp, D_pos, D_neg, Discriminator=tr.discriminate, real_data=gb["x"], fake_data=G,
                         alpha=alpha.float().to(dev))
    assert abs(d_loss.item() - d_ref.item()) < 1e-3 * abs(d_ref.item())
    # the critic step through the one-call penalty (`dpig_gp_double_backward`, the default) and through the taped double
    # backward: same loss, same flat critic gradient
    tr.gp_alpha = alpha.float().to(dev)
    assert tr._fused_gp() is not None
    o1 = tr._d_optim_eager(gb, update=False)
    g1 = tr.D_flat.grad.clone()
    tr.config.fused_gp = False
    assert tr._fused_gp() is None
    o2 = tr._d_optim_eager(gb, update=False)
    g2 = tr.D_flat.grad.clone()
    tr.config.fused_gp = True
    del tr.gp_alpha
    assert abs(o1["d_loss"].item() - d_ref.item()) < 1e-3 * abs(d_ref.item())
    assert abs(o1["d_loss"].item() - o2["d_loss"].item()) < 1e-5 * abs(d_ref.item())
    assert (g1 - g2).abs().max().item() < 1e-4 * g2.abs().max().item()
    o = tr.train_step(gb, gb)
    assert tr.d_opt.t == 5 and torch.isfinite(o["d_loss"])
    o = tr.train_step(gb, gb)
    assert tr.g_opt.t == 1 and tr.d_opt.t == 10 and torch.isfinite(o["g_loss"])
    lib.delete_all_params()


def test_pose_fc_autoencoder(dev):
    """PoseEncoderFCRes / PoseDecoderFCRes (models.py:488-515, as called at trainer.py:647-654): forward values,
    gradients of the reconstruction read-out through the coordinate head, straight-through visibility head."""
    import dpig_amd.tflib as lib
    from dpig_amd import models, slim
    from oracle import models as OM
    g = torch.Generator().manual_seed(4)
    B = 6
    rcv = torch.rand(B, 54, generator=g, dtype=torch.float64) * 2 - 1
    P = OM.ParamStore(seed=15)
    z_o = OM.pose_encoder_fc_res(P, rcv)
    coord_o, vis_o, prob_o = OM.pose_decoder_fc_res(P, z_o)
    names = list(P.p.keys())
    r = torch.rand(B, 36, generator=g, dtype=torch.float64) - 0.5
    grads = dict(zip(names, torch.autograd.grad((coord_o * r).sum(), [P.p[n] for n in names], allow_unused=True)))
    _load(P, dev)
    with slim.variable_scope("PoseAE"):
        z, enc_var = models.PoseEncoderFCRes(rcv.float().to(dev), z_num=32, repeat_num=4, hidden_num=512,
                                             data_format='NHWC', activation_fn=slim.leaky_relu)
        coord, vis, dec_var = models.PoseDecoderFCRes(z, 18, repeat_num=4, hidden_num=512, data_format='NHWC',
                                                      activation_fn=slim.leaky_relu)
    assert set(lib._params.keys()) == set(P.p.keys()) and len(enc_var) + len(dec_var) == len(names)
    assert _rel(z, z_o) < 1e-4 and _rel(coord, coord_o) < 1e-4
    # binaryRound: identical wherever the probability is not within round-off of 0.5
    safe = (prob_o - 0.5).abs() > 1e-4
    assert torch.equal(vis.detach().cpu().double()[safe], vis_o[safe])
    (coord * r.float().to(dev)).sum().backward()
    for n in names:
        if grads[n] is not None:
            assert _rel(lib._params[n].grad, grads[n]) < 2e-3, n
    lib.delete_all_params()


def test_stage1_bf16_compute_mode(dev):
    """Config(compute_dtype='bf16') (BASELINE configs 3-5): conv GEMMs on the bf16 matrix pipe, fp32 tensors /
    accumulation / master weights.  Embedding, generator output and losses stay within bf16 operand rounding of the
    fp64 oracle (8 mantissa bits per operand, averaged over the reduction), a training step runs, and fp32 mode is
    untouched afterwards."""
    import dpig_amd.hip_ops as H
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
    from oracle import models as OM
    lib.delete_all_params(); slim.reset_scopes()
    B, HID, ZN = 2, 64, 16
    np.random.seed(0)
    batch_np = synthetic.make_batch(B, seed=31)
    ob = OM.batch_to_torch(batch_np)
    P = OM.ParamStore(seed=12)
    with torch.no_grad():
        OM.stage1_g_loss(P, ob, hidden_num=HID, z_num=ZN)
        OM.stage1_d_loss(P, ob, hidden_num=HID, z_num=ZN)
    lib.set_device(dev)
    for n, v in P.state_numpy().items():
        lib.param(n, v, trainable=P.trainable[n])
    tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, conv_hidden_num=HID, z_num=ZN, compute_dtype='bf16c'), dev)
    batch = synthetic.to_device(batch_np, dev)
    try:
        tr.init_net(batch)
        assert H.get_compute() == "bf16c"
        with torch.no_grad():
            embs_o, G_o = OM.stage1_forward(P, ob, hidden_num=HID, z_num=ZN)
            embs, _ = tr.encode(batch)
            G, _ = tr.generate(embs, batch["pose"])
        rel = lambda a, b: (a.double().cpu() - b.double()).abs().max().item() / max(b.abs().max().item(), 1e-12)
        assert 1e-6 < rel(embs, embs_o) < 3e-2          # (> 1e-6: the bf16 pipe really was used)
        assert rel(G, G_o) < 5e-2
        gl_o, _ = OM.stage1_g_loss(P, ob, hidden_num=HID, z_num=ZN)
        tr.step = 1
        w0 = tr.G_flat.flat.detach().clone()
        out = tr.train_step(batch, batch)
        assert abs(float(out["g_loss"]) - float(gl_o)) < 5e-2 * abs(float(gl_o))
        assert all(np.isfinite(float(v)) for v in out.values() if hasattr(v, "numel") and v.numel() == 1)
        assert float((tr.G_flat.flat - w0).abs().max()) > 0
    finally:
        H.set_compute("f32")


def test_stage1_split_bf16_mode_holds_the_fp32_model_bar(dev):
    """Config(compute_dtype='bf16x3'): fp32 tensors, conv products as three bf16 MFMAs on two-term splits.  At a width where
    the split pipe serves most layers (64): embedding / generator output stay within 1e-4 max|ref| of the fp64 oracle (the
    exact path's own bar; north-star model bar 1e-3) and the critic logits within 1e-3.
    Gradients: every kernel is held to 2e-5 in test_conv_gpu.py; through ~40 ReLU layers the gradient is piecewise constant
    in the weights, so ANY 4e-6 perturbation (the split's operand truncation is <= 3.8e-6) moves it by flipping units that
    sit on a kink (module docstring of test_model_gpu.py).  The bound is therefore calibrated in-test: the split run's
    gradient may differ from the exact fp32 run's by no more than 3x what the exact kernels themselves show when the weights
    are perturbed by 4e-6 relative (measured: 5.8e-3 vs 4.9e-3 in L2), and by less than 2e-2 outright."""
    import dpig_amd.hip_ops as H
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
    from oracle import models as OM
    lib.delete_all_params(); slim.reset_scopes()
    B, HID, ZN = 2, 64, 16
    np.random.seed(0)
    batch_np = synthetic.make_batch(B, seed=33)
    ob = OM.batch_to_torch(batch_np)
    P = OM.ParamStore(seed=14)
    with torch.no_grad():
        OM.stage1_g_loss(P, ob, hidden_num=HID, z_num=ZN)
        OM.stage1_d_loss(P, ob, hidden_num=HID, z_num=ZN)
    lib.set_device(dev)
    for n, v in P.state_numpy().items():
        lib.param(n, v, trainable=P.trainable[n])
    tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, conv_hidden_num=HID, z_num=ZN, compute_dtype='bf16x3'), dev)
    batch = synthetic.to_device(batch_np, dev)
    rel = lambda a, b: (a.detach().double().cpu() - b.detach().double()).abs().max().item() / max(b.abs().max().item(), 1e-12)
    r = torch.randn(tuple(batch["x"].shape), device=dev, generator=torch.Generator(device=dev).manual_seed(5))

    def trunk_grad():
        tr.G_flat.zero_grad()
        embs, _ = tr.encode(batch)
        G, _ = tr.generate(embs, batch["pose"])
        G.backward(r)
        tr.G_flat.finalize()
        return embs.detach().clone(), tr.G_flat.grad.detach().double().clone()
    l2 = lambda a, b: float((a - b).norm() / b.norm())
    try:
        tr.init_net(batch)
        assert H.get_compute() == "bf16x3"
        with torch.no_grad():
            embs_o, G_o = OM.stage1_forward(P, ob, hidden_num=HID, z_num=ZN)
            d_real_o = OM.dcgan_discriminator(P, ob["x"])
            d_real = tr.discriminate(batch["x"])
            embs, _ = tr.encode(batch)
            G, _ = tr.generate(embs, batch["pose"])
        assert rel(embs, embs_o) < 1e-4 and rel(G, G_o) < 1e-4 and rel(d_real, d_real_o) < 1e-3
        e_x3, g_x3 = trunk_grad()
        H.set_compute("f32")
        e_32, g_32 = trunk_grad()
        assert not torch.equal(e_32, e_x3)                       # the split pipe really served the first run
        with torch.no_grad():
            w0 = tr.G_flat.flat.detach().clone()
            tr.G_flat.flat.mul_(1.0 + 4e-6 * torch.randn(w0.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(6)))
        _, g_pert = trunk_grad()
        with torch.no_grad():
            tr.G_flat.flat.copy_(w0)
        d_x3, d_pert = l2(g_x3, g_32), l2(g_pert, g_32)
        assert d_x3 < 3.0 * d_pert and d_x3 < 2e-2, (d_x3, d_pert)
    finally:
        H.set_compute("f32")


def _pose_rcv(B, seed):
    g = torch.Generator().manual_seed(seed)
    r = torch.rand(B, 18, 1, generator=g, dtype=torch.float64) * 127
    c = torch.rand(B, 18, 1, generator=g, dtype=torch.float64) * 63
    v = (torch.rand(B, 18, 1, generator=g, dtype=torch.float64) < 0.85).double()
    return torch.cat([r, c, v], -1).reshape(B, 54)


def test_stage2_pose_autoencoder_trainer(dev):
    """Model 2 (trainer.py:626-713): reconstruction loss of the pose auto-encoder, its gradients (x20, straight-through
    visibility head) and one TF-Adam(beta1=0.5) step against the oracle."""
    import dpig_amd.tflib as lib
    from dpig_amd import slim
    from dpig_amd.trainer import Config
    from dpig_amd.trainer_stage2 import DPIG_PoseRCV_AE_BodyROI
    from oracle import models as OM
    from oracle import ops as O
    lib.delete_all_params(); slim.reset_scopes()
    B, LR = 6, 1e-3
    rcv = _pose_rcv(B, 3)
    P = OM.ParamStore(seed=16)
    loss_o, z_o, G_o = OM.pose_ae_loss(P, rcv)
    names = list(P.p.keys())
    grads = dict(zip(names, torch.autograd.grad(loss_o * 20, [P.p[n] for n in names], allow_unused=True)))
    _load(P, dev)
    tr = DPIG_PoseRCV_AE_BodyROI(Config(batch_size=B, g_lr=LR), dev)
    batch = {"pose_rcv": rcv.float().to(dev)}
    tr.init_net(batch)
    assert set(lib._params.keys()) == set(P.p.keys()) and len(tr.G_var_pose) == len(names)
    assert (tr.g_opt.b1, tr.g_opt.b2) == (0.5, 0.999)
    o0 = tr.train_step(batch)                       # step 0: no update (trainer.py:678-680)
    assert tr.g_opt.t == 0 and abs(float(o0["reconstruct_loss"]) - loss_o.item()) < 1e-5 * loss_o.item()
    o1 = tr.train_step(batch)
    assert tr.g_opt.t == 1 and abs(float(o1["reconstruct_loss"]) - loss_o.item()) < 1e-5 * loss_o.item()
    assert _rel(o1["G_pose_rcv"], G_o.detach()) < 1e-4
    for n in names:
        if grads[n] is None:
            continue
        assert _rel(lib._params[n]._dpig_grad, grads[n]) < 2e-3, n
        p_new, _, _ = O.tf_adam_step(P.p[n].detach(), grads[n], torch.zeros_like(grads[n]), torch.zeros_like(grads[n]), LR, 0.5,
                                     0.999, 1e-8, 1)
        big = grads[n].abs() > 1e-3 * grads[n].abs().max()      # (sign-like first step: skip gradients at the rounding floor)
        assert ((lib._params[n].detach().double().cpu() - p_new).abs()[big]).max().item() < 0.02 * LR, n
    maps = tr.G_pose(o1["G_pose_rcv"])
    assert tuple(maps.shape) == (B, 128, 64, 18)
    lib.delete_all_params(); slim.reset_scopes()


def test_stage2_pose_embedding_gan_trainer(dev):
    """Model 4 (trainer.py:868-1040): frozen pose encoder -> real embeddings, PoseGaussian mapper, critic on the pair
    [real; fake], wgan losses, RMSProp + clip, loop order; sampled poses decode to keypoints."""
    import dpig_amd.tflib as lib
    from dpig_amd import slim
    from dpig_amd.trainer import Config
    from dpig_amd.trainer_stage2 import DPIG_subnetSamplePoseRCV_GAN_BodyROI
    from oracle import models as OM
    lib.delete_all_params(); slim.reset_scopes()
    B, LR = 6, 1e-3
    rcv = _pose_rcv(B, 4)
    g = torch.Generator().manual_seed(9)
    z = torch.randn(B, 32, generator=g, dtype=torch.float64) * 0.2
    P = OM.ParamStore(seed=17)
    g_ref, d_ref, fake_o, real_o = OM.pose_gan_losses(P, rcv, z)
    gnames = [n for n in P.p if n.startswith("PoseGaussian/")]
    dnames = [n for n in P.p if "Pose_emb_Discriminator." in n]
    gg = dict(zip(gnames, torch.autograd.grad(g_ref, [P.p[n] for n in gnames], retain_graph=True)))
    dg = dict(zip(dnames, torch.autograd.grad(d_ref, [P.p[n] for n in dnames])))
    _load(P, dev)
    tr = DPIG_subnetSamplePoseRCV_GAN_BodyROI(Config(batch_size=B, g_lr=LR, d_lr=LR), dev)
    batch = {"pose_rcv": rcv.float().to(dev)}
    tr.init_net(batch)
    assert set(lib._params.keys()) == set(P.p.keys())
    assert sorted(p.dpig_name for p in tr.G_flat.params) == sorted(gnames) and sorted(p.dpig_name for p in tr.D_flat.params) == sorted(dnames)
    zz = z.float().to(dev)
    real, _ = tr.encode_pose(batch["pose_rcv"])
    assert _rel(real, real_o) < 1e-4
    d_loss = tr.d_optim_embs(batch, z=zz)
    assert abs(d_loss.item() - d_ref.item()) < 1e-4 * max(abs(d_ref.item()), 1e-3)
    for n in dnames:
        assert _rel(lib._params[n]._dpig_grad, dg[n]) < 2e-3, n
        p_new, _, _ = OM.tf_rmsprop_step(P.p[n].detach(), dg[n], torch.ones_like(dg[n]), torch.zeros_like(dg[n]), LR)
        assert (lib._params[n].detach().double().cpu() - p_new.clamp(-0.01, 0.01)).abs().max().item() < 2e-5, n
    with torch.no_grad():
        for n in dnames:
            lib._params[n].copy_(P.p[n].to(torch.float32))
    g_loss = tr.g_optim_embs(batch, z=zz)
    assert abs(g_loss.item() - g_ref.item()) < 1e-4 * max(abs(g_ref.item()), 1e-3)
    for n in gnames:
        assert _rel(lib._params[n]._dpig_grad, gg[n]) < 2e-3, n
    t_g, t_d = tr.g_opt.t, tr.d_opt.t
    out = tr.train_step(batch)                      # step 0: 5 clipped critic updates, no mapper update
    assert "g_loss_embs" not in out and tr.d_opt.t == t_d + 5 and tr.g_opt.t == t_g
    out = tr.train_step(batch)
    assert "g_loss_embs" in out and tr.g_opt.t == t_g + 1 and tr.d_opt.t == t_d + 10
    assert float(tr.D_flat.flat.abs().max()) <= 0.01 + 1e-9
    rcv_s = tr.sample(zz)
    assert tuple(rcv_s.shape) == (B, 18, 3) and set(rcv_s[..., 2].unique().tolist()) <= {0.0, 1.0}
    lib.delete_all_params(); slim.reset_scopes()


@pytest.mark.parametrize("model", ["market", "df256"])
def test_stage1_bf16_storage_mode(dev, model):
    """Config(compute_dtype='bf16') (BASELINE configs 3-5): activations, their gradients and the filter shadows stored
    as bf16, bf16 matrix pipe, fp32 accumulation / master weights / gradients / optimizer.  Against the fp64 oracle on the
    UNROUNDED weights: embedding, generator output and losses within bf16 storage rounding (8 mantissa bits per stored
    activation, over a ~40-layer graph); a training step moves the fp32 masters and refreshes the shadows; the fp32 mode is
    untouched afterwards.  model 'df256' is the trainer_256.py graph (BASELINE configs[3])."""
    import dpig_amd.hip_ops as H
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
    from dpig_amd.trainer_256 import DPIG_Encoder_GAN_BodyROI_256
    from oracle import models as OM
    lib.delete_all_params(); slim.reset_scopes()
    np.random.seed(0)
    if model == "market":
        B, HID, ZN, Hh, Ww = 2, 64, 16, 128, 64
        cls, kw = DPIG_Encoder_GAN_BodyROI_FgBg, {}
    else:
        B, HID, ZN, Hh, Ww = 2, 32, 16, 256, 256
        cls, kw = DPIG_Encoder_GAN_BodyROI_256, {"img_H": 256, "img_W": 256}
    batch_np = synthetic.make_batch(B, img_H=Hh, img_W=Ww, seed=31)
    ob = OM.batch_to_torch(batch_np)
    P = OM.ParamStore(seed=12)
    with torch.no_grad():
        if model == "market":
            embs_o, G_o = OM.stage1_forward(P, ob, hidden_num=HID, z_num=ZN)
            gl_o = OM.stage1_g_loss(P, ob, hidden_num=HID, z_num=ZN)[0]
            dl_o = OM.stage1_d_loss(P, ob, hidden_num=HID, z_num=ZN)[0]
        else:
            ref = OM.stage1_256_forward(P, ob, HID, ZN, 6)
            embs_o, G_o, gl_o, dl_o = ref["embs"], ref["G"], ref["g_loss"], ref["d_loss"]
    lib.set_device(dev)
    for n, v in P.state_numpy().items():
        lib.param(n, v, trainable=P.trainable[n])
    tr = cls(Config(batch_size=B, conv_hidden_num=HID, z_num=ZN, compute_dtype='bf16', **kw), dev)
    batch = synthetic.to_device(batch_np, dev)
    try:
        tr.init_net(batch)
        assert H.get_compute() == "bf16"
        assert set(lib._params.keys()) == set(P.p.keys())
        assert len(tr.G_flat.shadows.params) > 20 and len(tr.D_flat.shadows.params) == 3
        with torch.no_grad():
            embs, _ = tr.encode(batch)
            G, _ = tr.generate(embs, batch["pose"])
        assert G.dtype == torch.float32                         # the 3-channel image stays fp32
        rel = lambda a, b: (a.double().cpu() - b.double()).abs().max().item() / max(b.abs().max().item(), 1e-12)
        # The bound comes from the storage FORMAT, not from this library: the oracle graph is re-run with every
        # tensor the library keeps in bf16 rounded where the library rounds it (oracle.models.STORE), once exactly and once with the
        # stored values perturbed by 1e-6 relative before rounding -- the size of an fp32-vs-fp64 accumulation difference, which flips
        # roundings that sit on a bf16 boundary.  What those flips move (the format's own noise floor, ~6e-3 of max|ref| through the
        # ~50 stored tensors of the graph) is the unit: the library must stay within 3 x of it from the emulated-storage oracle.
        bf = lambda t: t.bfloat16().to(t.dtype)
        gn = torch.Generator().manual_seed(5)
        jit = lambda t: bf(t * (1 + 2e-6 * (torch.rand(t.shape, generator=gn, dtype=t.dtype) - 0.5)))

        def oracle_eg(store):
            OM.STORE = store
            try:
                with torch.no_grad():
                    if model == "market":
                        return OM.stage1_forward(P, ob, hidden_num=HID, z_num=ZN)
                    r = OM.stage1_256_forward(P, ob, HID, ZN, 6)
                    return r["embs"], r["G"]
            finally:
                OM.STORE = None
        embs_s, G_s = oracle_eg(bf)
        embs_j, G_j = oracle_eg(jit)
        floor_e, floor_G = rel(embs_j, embs_s), rel(G_j, G_s)
        e_embs, e_G = rel(embs, embs_s), rel(G, G_s)
        msg = ("bf16 storage (%s): vs emulated-storage oracle: embedding %.3e (format floor %.3e), G %.3e (floor %.3e); vs exact oracle %.3e / %.3e"
               % (model, e_embs, floor_e, e_G, floor_G, rel(embs, embs_o), rel(G, G_o)))
        print(msg)
        assert 1e-5 < e_embs <= 3 * max(floor_e, 2.0 ** -8), msg
        assert e_G <= 3 * max(floor_G, 2.0 ** -8), msg
        tr.step = 1
        w0 = tr.G_flat.flat.detach().clone()
        sh0 = tr.G_flat.shadows.buf.detach().clone()
        d0 = tr._d_optim_eager(batch, update=False)         # (before the generator moves: the oracle's weights)

        # the critic alone, on the LIBRARY's own generator output (the generator's deviation is bounded above): emulated-storage
        # oracle critic, same 3 x its own flip floor (batch norm over 2-4 images makes it the touchiest part of the graph)
        def oracle_dloss(store):
            OM.STORE = store
            try:
                with torch.no_grad():
                    Gc = G.double().cpu()
                    if model == "market":
                        dr, df_ = OM.dcgan_discriminator(P, ob["x"], "dcgan"), OM.dcgan_discriminator(P, Gc, "dcgan")
                    else:
                        dz = OM.dcgan_discriminator(P, torch.cat([ob["x"], Gc], dim=0), "dcgan")
                        dr, df_ = torch.split(dz, dz.shape[0] // 2)
                    return float(OM.gan_loss("dcgan", dr, df_)[1])
            finally:
                OM.STORE = None
        dl_s, dl_j = oracle_dloss(bf), oracle_dloss(jit)
        floor_d = abs(dl_j - dl_s) / abs(dl_s)
        e_d = abs(float(d0["d_loss"]) - dl_s) / abs(dl_s)
        msg_d = "bf16 storage (%s): d_loss vs emulated-storage critic %.3e (floor %.3e); vs exact oracle %.3e" % (
            model, e_d, floor_d, abs(float(d0["d_loss"]) - float(dl_o)) / abs(float(dl_o)))
        print(msg_d)
        try:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
            with open(os.path.join(root, "gpurun_out", "bf16_model_bars.txt"), "a") as fh:
                fh.write(msg + "\n" + msg_d + "\n")
        except Exception:
            pass
        assert e_d <= 3 * max(floor_d, 2.0 ** -7), msg_d
        out = tr.train_step(batch, batch)
        assert abs(float(out["g_loss"]) - float(gl_o)) < 5e-2 * abs(float(gl_o))
        assert all(np.isfinite(float(v)) for v in out.values() if hasattr(v, "numel") and v.numel() == 1)
        assert float((tr.G_flat.flat - w0).abs().max()) > 0
        assert tr.G_flat.grad.dtype == torch.float32 and bool(torch.isfinite(tr.G_flat.grad).all())
        # the shadows follow the masters: plain shadow == bf16(master) for every conv filter
        assert not torch.equal(tr.G_flat.shadows.buf, sh0)
        for p in tr.G_flat.shadows.params[:4]:
            assert torch.equal(p._dpig_shadow[0], p.data.to(torch.bfloat16))
            assert torch.equal(p._dpig_shadow[1], p.data.permute(0, 1, 3, 2).contiguous().to(torch.bfloat16))
    finally:
        H.set_compute("f32")
        lib.delete_all_params(); slim.reset_scopes()


def test_inference_harness(dev):
    """SURVEY 8f-3 (tester.py:256-417): with sampling off the harness reproduces the trainer's generator path
    (same variables, pose maps rasterised from the keypoints) and the critic score; with sampling on it draws
    appearance / pose from the FC mappers; outputs are denormalised images with an SSIM against the input."""
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic
    from dpig_amd.tester import DPIG_FourNetsFgBg_testOnly
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
    from oracle import ops as O
    lib.delete_all_params(); slim.reset_scopes()
    B = 2
    np.random.seed(1)
    cfg = Config(batch_size=B, conv_hidden_num=16, z_num=8)
    tr = DPIG_Encoder_GAN_BodyROI_FgBg(cfg, dev)
    batch = synthetic.to_device(synthetic.make_batch(B, seed=41), dev)
    tr.init_net(batch)
    rng = np.random.default_rng(7)
    rcv = np.zeros((B, 18, 3), np.float32)
    rcv[..., 0] = rng.integers(4, 124, (B, 18)); rcv[..., 1] = rng.integers(4, 60, (B, 18)); rcv[..., 2] = 1.0
    rcv_t = torch.from_numpy(rcv.reshape(B, -1)).to(dev)
    # the trainer's path on the pose map of exactly these keypoints
    pose_map = O.tf_poseInflate(O.coord2channel_simple_rcv(torch.from_numpy(rcv.reshape(B, -1)).double(), 18, False, 128, 64),
                                18, 4, 128, 64).float().to(dev)
    with torch.no_grad():
        embs, _ = tr.encode(batch)
        G_ref, _ = tr.generate(embs, pose_map)
        score_ref = tr.discriminate(G_ref)
    te = DPIG_FourNetsFgBg_testOnly(cfg, dev)
    te.built = False
    # the Encoder / ID_AE variables exist already (created by the trainer): reuse them, create the samplers
    slim.reset_scopes()
    out = te.run(batch, rcv_t)
    assert torch.equal(out["pose_map"], pose_map)
    assert torch.allclose(out["G"], torch.clamp((G_ref + 1) * 127.5, 0, 255), atol=1e-3)
    assert torch.allclose(out["G_dis_score"], score_ref.reshape(B, -1).mean(1), atol=1e-4)
    assert float(out["reconstruct_loss"]) == 0.0 and tuple(out["ssim_G_x"].shape) == (B,)
    # sampling on: fixed noise makes it repeatable; one foreground for the whole batch
    ts = DPIG_FourNetsFgBg_testOnly(cfg, dev, sample_app=True, sample_pose=True, one_app_per_batch=True)
    ts.built = True                                   # every variable exists now
    z_fg, z_bg = torch.randn(B, 224, device=dev) * 0.2, torch.randn(B, 128, device=dev) * 0.2
    o1 = ts.run(batch, rcv_t, z_fg=z_fg, z_bg=z_bg)
    o2 = ts.run(batch, rcv_t, z_fg=z_fg, z_bg=z_bg)
    assert torch.equal(o1["G"], o2["G"]) and torch.isfinite(o1["G"]).all()
    assert torch.equal(o1["embs"][0, :224], o1["embs"][1, :224]) and not torch.equal(o1["embs"][0, 224:], o1["embs"][1, 224:])
    assert float(o1["G"].min()) >= 0.0 and float(o1["G"].max()) <= 255.0
    assert set(np.unique(o1["G_pose_rcv"][..., 2].cpu().numpy())) <= {0.0, 1.0}       # binaryRound visibilities


def test_inference_harness_against_oracle(dev):
    """tester.py:256-417 with every sampler on (appearance from the two Gaussian mappers, pose from the PoseGaussian
    mapper through the pose decoder), fixed noise: the harness's embedding, decoded keypoints, rasterised pose maps,
    generated image and critic score against the ORACLE's graph of the same chain (not against the trainer's own path)."""
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic
    from dpig_amd.tester import DPIG_FourNetsFgBg_testOnly
    from dpig_amd.trainer import Config
    from oracle import models as OM
    from oracle import ops as O
    lib.delete_all_params(); slim.reset_scopes()
    B, HID, ZN = 2, 16, 8
    batch_np = synthetic.make_batch(B, seed=43)
    ob = OM.batch_to_torch(batch_np)
    rcv = _pose_rcv(B, 6)
    g = torch.Generator().manual_seed(10)
    z_fg = torch.randn(B, 224, generator=g, dtype=torch.float64) * 0.2
    z_bg = torch.randn(B, 128, generator=g, dtype=torch.float64) * 0.2
    z_pose = torch.randn(B, 32, generator=g, dtype=torch.float64) * 0.2
    P = OM.ParamStore(seed=18)
    with torch.no_grad():
        norm = OM.normalise_pose_rcv(rcv)
        OM.pose_encoder_fc_res(P, norm)                                          # (exists in the graph; unused when sampling)
        pose_e = OM.gaussian_fc_res(P, z_pose, 32, 4, 512, scope="PoseGaussian/G_FC")
        coord, vis, _ = OM.pose_decoder_fc_res(P, pose_e)
        G_pose_rcv = torch.cat([coord.reshape(B, 18, 2), vis.unsqueeze(-1)], -1)
        pose_map = O.tf_poseInflate(O.coord2channel_simple_rcv(G_pose_rcv.reshape(B, -1), 18, True, 128, 64), 18, 4, 128, 64)
        OM.encoder_fgbg(P, ob["x"], ob["mask_r6"], ob["part_bbox"], ob["part_vis"], 7, 32, 5, HID)
        fg = OM.gaussian_fc_res(P, z_fg, 224, 4, 512, scope="Gaussian_FC_Fg/G_FC")
        bg = OM.gaussian_fc_res(P, z_bg, 128, 4, 256, scope="Gaussian_FC_Bg/G_FC")
        embs = torch.cat([fg, bg], -1)
        embs_rep = embs.reshape(B, 1, 1, -1).expand(B, 128, 64, embs.shape[1])
        G, _ = OM.generator_uae(P, embs_rep, pose_map, 3, ZN, 5, HID)
        score = OM.dcgan_discriminator(P, G, "dcgan")
    _load(P, dev)
    te = DPIG_FourNetsFgBg_testOnly(Config(batch_size=B, conv_hidden_num=HID, z_num=ZN), dev, sample_app=True, sample_pose=True,
                                    sample_pose_embedding=True)
    out = te.run(synthetic.to_device(batch_np, dev), rcv.float().to(dev), z_fg=z_fg.float().to(dev), z_bg=z_bg.float().to(dev),
                 z_pose=z_pose.float().to(dev))
    assert set(lib._params.keys()) == set(P.p.keys())
    assert _rel(out["embs"], embs) < 1e-4
    assert _rel(out["G_pose_rcv"][..., :2], G_pose_rcv[..., :2]) < 1e-4 and torch.equal(out["G_pose_rcv"][..., 2].cpu().double(), G_pose_rcv[..., 2])
    assert torch.equal(out["pose_map"].cpu().double(), pose_map)
    assert (out["G"].double().cpu() - torch.clamp((G + 1) * 127.5, 0, 255)).abs().max().item() < 1e-3 * 255
    assert _rel(out["G_dis_score"], score.reshape(B, -1).mean(1)) < 1e-3
    lib.delete_all_params(); slim.reset_scopes()


@pytest.mark.parametrize("kind,flags", [
    ("four_nets", dict(sample_app=True, sample_pose=True, one_app_per_batch=True)),
    ("four_nets", dict()),
    ("sample_factor", dict(sample_fg=True, sample_bg=False, sample_pose=True)),
    ("sample_factor", dict(sample_fg=False, sample_bg=True, sample_pose=False)),
    ("condition", dict()),
    ("condition_256", dict()),
    ("sample_factor_256", dict(sample_app=True, sample_pose=True)),
    ("sample_factor_256", dict()),
])
def test_remaining_tester_pipelines_against_oracle(dev, kind, flags):
    """The other five pipelines of tester.py (:4-253, 419-613, 616-772, 775-914, 917-1138) against the oracle's layer-for-layer
    graph of the same chain: embedding fed to the generator, decoded / held keypoints, rasterised pose map (bit-exact), the
    generated image and the critic score; fixed noise for the Gaussian mappers."""
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic, tester
    from dpig_amd.trainer import Config
    from oracle import models as OM
    lib.delete_all_params(); slim.reset_scopes()
    is256 = kind.endswith("256")
    Hh, W, rep = (256, 256, 6) if is256 else (128, 64, 5)
    B, HID, ZN = 2, 8 if is256 else 16, 8
    batch_np = synthetic.make_batch(B, img_H=Hh, img_W=W, seed=47)
    ob = OM.batch_to_torch(batch_np)
    g = torch.Generator().manual_seed(12)
    rcv = torch.stack([torch.rand(B, 18, generator=g, dtype=torch.float64) * (Hh - 1), torch.rand(B, 18, generator=g, dtype=torch.float64) * (W - 1),
                       (torch.rand(B, 18, generator=g, dtype=torch.float64) < 0.8).double()], -1).reshape(B, 54)
    rcv = rcv.float().double()
    n_app = 224 if kind in ("four_nets", "sample_factor_256") else 224
    z = dict(z_app=torch.randn(B, n_app, generator=g, dtype=torch.float64) * 0.2, z_fg=torch.randn(B, 224, generator=g, dtype=torch.float64) * 0.2,
             z_bg=torch.randn(B, 128, generator=g, dtype=torch.float64) * 0.2,
             z_pose=torch.randn(B, 100 if kind == "four_nets" else 32, generator=g, dtype=torch.float64) * 0.2)
    P = OM.ParamStore(seed=19)
    with torch.no_grad():
        ref = OM.tester_pipeline(P, kind, ob, rcv=rcv, pose_target=ob["pose"], hidden_num=HID, z_num=ZN, repeat_num=rep, img_H=Hh,
                                 img_W=W, **z, **flags)
    _load(P, dev)
    cls = {"four_nets": tester.DPIG_FourNets_testOnly, "sample_factor": tester.DPIG_FourNetsFgBg_testOnlySampleFactor,
           "condition": tester.DPIG_FourNetsFgBg_testOnlyCondition, "condition_256": tester.DPIG_ThreeNetsApp_testOnlyCondition_256,
           "sample_factor_256": tester.DPIG_ThreeNetsApp_testOnlySampleFactor_256}[kind]
    te = cls(Config(batch_size=B, conv_hidden_num=HID, z_num=ZN, img_H=Hh, img_W=W), dev, **flags)
    gb = synthetic.to_device(batch_np, dev)
    out = te.run(gb, pose_rcv=rcv.float().to(dev), pose_target=gb["pose"], **{k: v.float().to(dev) for k, v in z.items()})
    assert set(lib._params.keys()) == set(P.p.keys()), (sorted(set(lib._params) ^ set(P.p))[:6])
    assert _rel(out["embs"], ref["embs"]) < 1e-4
    if "G_pose_rcv" in ref:
        assert _rel(out["G_pose_rcv"][..., :2], ref["G_pose_rcv"][..., :2]) < 1e-4
        assert torch.equal(out["G_pose_rcv"][..., 2].cpu().double(), ref["G_pose_rcv"][..., 2])
        assert abs(float(out["reconstruct_loss"]) - float(ref["reconstruct_loss"])) <= 1e-5 * max(1.0, float(ref["reconstruct_loss"]))
    assert torch.equal(out["pose_map"].cpu().double(), ref["pose_map"])
    assert (out["G"].double().cpu() - torch.clamp((ref["G"] + 1) * 127.5, 0, 255)).abs().max().item() < 1e-3 * 255
    if "score" in ref:
        assert _rel(out["G_dis_score"], ref["score"]) < 1e-3
    else:
        assert "G_dis_score" not in out
    out2 = te.run(gb, pose_rcv=rcv.float().to(dev), pose_target=gb["pose"], **{k: v.float().to(dev) for k, v in z.items()})
    assert torch.equal(out2["G"], out["G"])               # second call reuses the variables
    lib.delete_all_params(); slim.reset_scopes()


# ---- DeepFashion 256 x 256 stage II (run_DF_train.sh:39-77; trainer_256.py:266-700) and BASELINE configs[4] --------------
def _s2gold():
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = np.load(os.path.join(here, "golden", "stage2_df256_w16.npz"))
    return mg, gold


def _near(a, b, tol):
    return abs(float(a) - float(b)) <= tol * max(abs(float(b)), 1e-3)


def test_df256_stage2_appearance_gan_trainer(dev):
    """Model 102 (trainer_256.py:266-400): frozen `GeneratorCNN_ID_Encoder_BodyROI` (repeat_num + 1 levels, 48 x 48 crops, no
    visibility flags) -> real embeddings, `Gaussian_FC` mapper, `FCDis_` critic on the pair, wgan losses, RMSProp + clip, the
    loop order -- against the committed golden fixture AND the oracle's parameter gradients."""
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic
    from dpig_amd.trainer import Config
    from dpig_amd.trainer_256 import DPIG_Encoder_subSampleAppNet_GAN_BodyROI_256
    from oracle import models as OM
    lib.delete_all_params(); slim.reset_scopes()
    mg, gold = _s2gold()
    bseed, pseed, zseed, _, B, W = [int(v) for v in gold["meta"]]
    LR = 1e-3
    batch_np = synthetic.make_batch(B, img_H=256, img_W=256, seed=bseed)
    ob = OM.batch_to_torch(batch_np)
    P = OM.ParamStore(seed=pseed)
    with torch.no_grad():
        embs_o = OM.encoder_body_roi(P, ob["x"], ob["part_bbox"], 7, 32, 7, W, roi_size=48)
    gz = torch.Generator().manual_seed(zseed)
    z = torch.randn(B, embs_o.shape[1], generator=gz, dtype=torch.float64) * 0.2
    g_ref, d_ref, fake_o = OM.stage2_256_losses(P, embs_o, z)
    gnames = [n for n in P.p if n.startswith("Gaussian_FC/")]
    dnames = [n for n in P.p if "FCDis_Discriminator." in n]
    gg = dict(zip(gnames, torch.autograd.grad(g_ref, [P.p[n] for n in gnames], retain_graph=True)))
    dg = dict(zip(dnames, torch.autograd.grad(d_ref, [P.p[n] for n in dnames])))
    _load(P, dev)
    cfg = Config(batch_size=B, img_H=256, img_W=256, conv_hidden_num=W, g_lr=LR, d_lr=LR)
    tr = DPIG_Encoder_subSampleAppNet_GAN_BodyROI_256(cfg, dev)
    gb = synthetic.to_device(batch_np, dev)
    tr.init_net(gb)
    assert set(lib._params.keys()) == set(P.p.keys())
    assert sorted(p.dpig_name for p in tr.G_flat.params) == sorted(gnames) and sorted(p.dpig_name for p in tr.D_flat.params) == sorted(dnames)
    real, _ = tr.encode(gb)
    assert tuple(real.shape) == (B, 224)
    assert _rel(real, torch.from_numpy(gold["m102/embs"])) < 1e-3 and _rel(real, embs_o) < 1e-4
    zz = z.float().to(dev)
    with torch.no_grad():
        fake, _ = tr.mapper(tr.dim, zz)
    assert _rel(fake, torch.from_numpy(gold["m102/fake"])) < 1e-3
    d_loss = tr.d_optim_embs(gb, z=zz)                        # critic first: the mapper still has the oracle's weights
    assert _near(d_loss.item(), gold["m102/d_loss"], 1e-3) and _near(d_loss.item(), d_ref.item(), 1e-4)
    for n in dnames:
        assert _rel(lib._params[n]._dpig_grad, dg[n]) < 2e-3, n
        p_new, _, _ = OM.tf_rmsprop_step(P.p[n].detach(), dg[n], torch.ones_like(dg[n]), torch.zeros_like(dg[n]), LR)
        assert (lib._params[n].detach().double().cpu() - p_new.clamp(-0.01, 0.01)).abs().max().item() < 2e-5, n
    with torch.no_grad():
        for n in dnames:
            lib._params[n].copy_(P.p[n].to(torch.float32))
    g_loss = tr.g_optim_embs(gb, z=zz)
    assert _near(g_loss.item(), gold["m102/g_loss"], 1e-3) and _near(g_loss.item(), g_ref.item(), 1e-4)
    for n in gnames:
        assert _rel(lib._params[n]._dpig_grad, gg[n]) < 2e-3, n
    t_g, t_d = tr.g_opt.t, tr.d_opt.t
    out = tr.train_step(gb)                 # step 0: five clipped critic updates, no mapper update (trainer_256.py:362-373)
    assert "g_loss_embs" not in out and tr.d_opt.t == t_d + 5 and tr.g_opt.t == t_g
    out = tr.train_step(gb)
    assert "g_loss_embs" in out and tr.g_opt.t == t_g + 1 and tr.d_opt.t == t_d + 10
    assert float(tr.D_flat.flat.abs().max()) <= 0.01 + 1e-9
    lib.delete_all_params(); slim.reset_scopes()


def test_df256_stage2_pose_trainers(dev):
    """Models 103 / 104 (trainer_256.py:404-509, 511-700): the pose auto-encoder and the pose-embedding GAN with the 256-pixel
    keypoint normalisation, against the golden fixture."""
    import dpig_amd.tflib as lib
    from dpig_amd import slim
    from dpig_amd.trainer import Config
    from dpig_amd.trainer_256 import DPIG_PoseRCV_AE_BodyROI_256, DPIG_subnetSamplePoseRCV_GAN_BodyROI_256
    from oracle import models as OM
    mg, gold = _s2gold()
    _, pseed, zseed, poseseed, B, _ = [int(v) for v in gold["meta"]]
    rcv = mg.pose_rcv_256(6, poseseed)
    cfg = Config(batch_size=6, img_H=256, img_W=256, g_lr=1e-3, d_lr=1e-3)
    # model 103
    lib.delete_all_params(); slim.reset_scopes()
    P2 = OM.ParamStore(seed=pseed + 1)
    OM.pose_ae_loss(P2, rcv, img_H=256, img_W=256)
    _load(P2, dev)
    tr = DPIG_PoseRCV_AE_BodyROI_256(cfg, dev)
    batch = {"pose_rcv": rcv.float().to(dev)}
    tr.init_net(batch)
    assert set(lib._params.keys()) == set(P2.p.keys())
    o0 = tr.train_step(batch)
    assert _near(o0["reconstruct_loss"], gold["m103/reconstruct_loss"], 1e-4)
    o1 = tr.train_step(batch)
    assert tr.g_opt.t == 1 and _rel(o1["G_pose_rcv"], torch.from_numpy(gold["m103/G_pose_rcv"])) < 1e-3
    assert tuple(tr.G_pose(o1["G_pose_rcv"]).shape) == (6, 256, 256, 18)
    # model 104
    lib.delete_all_params(); slim.reset_scopes()
    P3 = OM.ParamStore(seed=pseed + 2)
    gz = torch.Generator().manual_seed(zseed)
    torch.randn(B, 224, generator=gz, dtype=torch.float64)            # (the generator drew model 102's z first)
    zz = torch.randn(6, 32, generator=gz, dtype=torch.float64) * 0.2
    OM.pose_gan_losses(P3, rcv, zz, img_H=256, img_W=256)
    _load(P3, dev)
    tr = DPIG_subnetSamplePoseRCV_GAN_BodyROI_256(cfg, dev)
    tr.init_net(batch)
    assert set(lib._params.keys()) == set(P3.p.keys())
    real, _ = tr.encode_pose(batch["pose_rcv"])
    assert _rel(real, torch.from_numpy(gold["m104/real"])) < 1e-3
    snap = [p.detach().clone() for p in tr.D_flat.params]
    d_loss = tr.d_optim_embs(batch, z=zz.float().to(dev))
    assert _near(d_loss.item(), gold["m104/d_loss"], 1e-3)
    with torch.no_grad():
        for p, s0 in zip(tr.D_flat.params, snap):
            p.copy_(s0)
    g_loss = tr.g_optim_embs(batch, z=zz.float().to(dev))
    assert _near(g_loss.item(), gold["m104/g_loss"], 1e-3)
    lib.delete_all_params(); slim.reset_scopes()


def test_df256_wgan_gp_bf16_step(dev):
    """BASELINE configs[4]'s per-GPU workload in small: model 101 (trainer_256.py:31-88) with MODE='wgan-gp' (LayerNorm
    critic on the pair [x; G], 8 logit rows per image, gradient penalty through the double-backward sweep) in bf16 storage
    mode.  d_loss for a pinned alpha against the fp64 oracle within bf16 storage rounding; a full step (5 critic iterations +
    g_optim) stays finite and moves both parameter sets."""
    import dpig_amd.hip_ops as H
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic
    from dpig_amd.trainer import Config
    from dpig_amd.trainer_256 import DPIG_Encoder_GAN_BodyROI_256
    from oracle import models as OM
    lib.delete_all_params(); slim.reset_scopes()
    B, HID, ZN = 2, 32, 16
    batch_np = synthetic.make_batch(B, img_H=256, img_W=256, seed=33)
    ob = OM.batch_to_torch(batch_np)
    P = OM.ParamStore(seed=14)
    with torch.no_grad():
        embs_o = OM.encoder_roi(P, ob["x"], ob["part_bbox"], ob["part_vis"], 7, 32, 7, HID, roi_size=64)
        G_o, _ = OM.generator_uae(P, embs_o.reshape(B, 1, 1, -1).expand(B, 256, 256, embs_o.shape[1]), ob["pose"], 3, ZN, 5, HID)
    g = torch.Generator().manual_seed(15)
    alpha = torch.rand(B, generator=g, dtype=torch.float64)
    D = lambda t: OM.dcgan_discriminator(P, t, "wgan-gp")          # noqa: E731
    _, d_ref = OM.gan_losses("wgan-gp", D, ob["x"], G_o, alpha)
    lib.set_device(dev)
    for n, v in P.state_numpy().items():
        lib.param(n, v, trainable=P.trainable[n])
    tr = DPIG_Encoder_GAN_BodyROI_256(Config(batch_size=B, img_H=256, img_W=256, conv_hidden_num=HID, z_num=ZN, gan_mode='wgan-gp',
                                             compute_dtype='bf16'), dev)
    batch = synthetic.to_device(batch_np, dev)
    try:
        tr.init_net(batch)
        assert H.get_compute() == "bf16" and set(lib._params.keys()) == set(P.p.keys())
        assert (tr.d_opt.b1, tr.d_opt.b2) == (0.5, 0.9)
        tr.gp_alpha = alpha.float().to(dev)
        assert tr._fused_gp() is not None, "bf16 mode must take the one-call penalty (dpig_gp_double_backward, DPIG_COMPUTE_BF16_STORE)"
        d0 = tr._d_optim_eager(batch, update=False)
        g_fused = tr.D_flat.grad.detach().clone()
        print("df256 wgan-gp bf16: d_loss %.5f (oracle %.5f)" % (float(d0["d_loss"]), float(d_ref)))
        # measured 7.6e-4 (the generated image carries bf16 storage noise of the whole E + G forward; the penalty term's own parity is
        # test_fused_gp_double_backward_bf16_storage: every link at the per-kernel bound)
        assert abs(float(d0["d_loss"]) - float(d_ref)) < 1e-2 * max(abs(float(d_ref)), 1.0)
        # the same critic update through the taped second-level-autograd path: every critic gradient agrees
        tr.config.fused_gp = False
        d1 = tr._d_optim_eager(batch, update=False)
        tr.config.fused_gp = True
        g_taped = tr.D_flat.grad.detach().clone()
        assert abs(float(d1["d_loss"]) - float(d0["d_loss"])) < 1e-3 * abs(float(d0["d_loss"]))
        for i, (prm, o) in enumerate(zip(tr.D_flat.params, tr.D_flat.offsets)):
            a, q = g_fused[o:o + prm.numel()], g_taped[o:o + prm.numel()]
            if float(q.norm()) > 0:
                assert float((a - q).norm() / q.norm()) < 1e-2, "critic parameter %d %s" % (i, tuple(prm.shape))
        tr.gp_alpha = None
        w_g, w_d = tr.G_flat.flat.detach().clone(), tr.D_flat.flat.detach().clone()
        o = tr.train_step(batch, batch)                     # step 0: five critic iterations
        assert tr.d_opt.t == 5 and torch.isfinite(o["d_loss"])
        o = tr.train_step(batch, batch)
        assert tr.g_opt.t == 1 and tr.d_opt.t == 10 and torch.isfinite(o["g_loss"]) and torch.isfinite(o["d_loss"])
        assert float((tr.G_flat.flat - w_g).abs().max()) > 0 and float((tr.D_flat.flat - w_d).abs().max()) > 0
        assert bool(torch.isfinite(tr.G_flat.grad).all()) and bool(torch.isfinite(tr.D_flat.grad).all())
    finally:
        H.set_compute("f32")
        lib.delete_all_params(); slim.reset_scopes()


def test_stage2_graph_warmup_leaves_the_learning_rates_alone(dev):
    """`enable_graphs` warms up with real eager steps at step 1; with lr_update_step = 2 those steps halve g_lr / d_lr in
    place -- the warm-up must put them back like the weights and optimizer slots; `_feed` refuses a batch the graphs were not captured for."""
    import dpig_amd.hip_ops as H
    import dpig_amd.tflib as lib
    import numpy as np
    from dpig_amd import slim, synthetic
    from dpig_amd.trainer import Config
    from dpig_amd.trainer_stage2 import DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI
    try:
        lib.delete_all_params(); slim.reset_scopes(); lib.set_device(dev)
        np.random.seed(3)
        batch = synthetic.to_device(synthetic.make_batch(2, seed=40), dev)
        tr = DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI(Config(batch_size=2, conv_hidden_num=16, g_lr=1e-3, d_lr=2e-3, lr_update_step=2), dev)
        tr.init_net(batch)
        tr.enable_graphs(batch)
        assert float(tr.g_lr) == pytest.approx(1e-3) and float(tr.d_lr) == pytest.approx(2e-3)
        bad = dict(batch)
        bad["x"] = batch["x"][:1]
        with pytest.raises(RuntimeError):
            tr.train_step(bad)
        with pytest.raises(RuntimeError):
            tr.train_step(dict(batch, extra=batch["x"]))
    finally:
        H.set_compute("f32")
        lib.delete_all_params(); slim.reset_scopes()


def test_join_side_streams_accepts_an_unindexed_device(dev):
    """the side streams are keyed by the tensors' concrete device index; a trainer built with torch.device('cuda') (index None)
    must still have its current stream wait for them."""
    from dpig_amd import autograd as A
    x = torch.ones(1 << 20, device=dev)
    with A.side_branch(x, key="t", enabled=True) as sb:
        y = x * 2
        for _ in range(50):
            y = y + 1
    assert any(idx == dev.index for (idx, _) in A._SIDE_STREAMS)
    A.join_side_streams(torch.device("cuda"))            # index None
    A.join_side_streams("cuda")
    assert float(y[0]) == 52.0
