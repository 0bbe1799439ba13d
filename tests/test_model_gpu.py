"""GPU parity of the model graphs (E, G, D, losses, one optimizer step) against the CPU oracle.

Live comparison at a reduced width (conv_hidden_num=16) that the fp64 oracle finishes in seconds
on the GPU box's host cores; the full-width model is pinned by tests/test_golden_gpu.py against
committed fixtures.

Tolerances.  Forward activations: 1e-3 relative to max|ref| (north_star bar; observed ~1e-6).
Gradients of a network with ReLU / LeakyReLU kinks are piecewise constant in the input: a
pre-activation that sits within fp32 round-off (~1e-6 relative) of a kink takes a different slope
in an fp32 and an fp64 evaluation.  Measured here (scripts/diag_dg.py, DESIGN.md "kinks"): feeding
the *fp64 oracle itself* the HIP generator output (which differs from the oracle's by 1.1e-6)
moves dD/dG by 2.1 % through ONE flipped LeakyReLU unit, while a random perturbation of the same
size moves it by 3e-6.  So:
  * gradients through the smooth-ish E+G trunk (ReLU flips there each carry ~1/sqrt(width) of a
    layer's gradient) are checked tightly with a linear read-out loss        -> 2e-3
  * gradients of the full GAN loss (through D's batch-of-2 BatchNorm + LeakyReLU) are checked
    against the coarse bound a single flip can produce                        -> 3e-2
  * exact (2e-5) gradient parity of every kernel is in test_conv_gpu.py / test_ops_gpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HID, ZNUM = 16, 8


def _rel(got, ref):
    ref = ref.detach().double()
    got = got.detach().double().cpu()
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


def _setup(dev, B=2, seed=3, from_keypoints=False):
    import dpig_amd.tflib as lib
    from dpig_amd import slim, synthetic
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
    from oracle import models as OM
    lib.delete_all_params()
    slim.reset_scopes()
    np.random.seed(0)
    batch_np = synthetic.make_batch(B, seed=seed)
    if from_keypoints:                                 # geometry derived from keypoints by the converter's rules (dataprep)
        batch_np = {k: v for k, v in synthetic.make_batch_from_keypoints(B, seed=seed).items() if k != "keypoints"}
    ob = OM.batch_to_torch(batch_np)
    P = OM.ParamStore(seed=11)
    # run the oracle once to create every variable, then load the same values into the HIP model
    OM.stage1_g_loss(P, ob, hidden_num=HID, z_num=ZNUM)
    OM.stage1_d_loss(P, ob, hidden_num=HID, z_num=ZNUM)
    lib.set_device(dev)
    for n, v in P.state_numpy().items():
        lib.param(n, v, trainable=P.trainable[n])
    cfg = Config(batch_size=B, conv_hidden_num=HID, z_num=ZNUM, g_lr=2e-3, d_lr=2e-3)
    tr = DPIG_Encoder_GAN_BodyROI_FgBg(cfg, dev)
    gb = synthetic.to_device(batch_np, dev)
    tr.init_net(gb)
    return tr, gb, P, ob, OM


def test_param_names_match_oracle(dev):
    import dpig_amd.tflib as lib
    tr, gb, P, ob, OM = _setup(dev)
    assert set(lib._params.keys()) == set(P.p.keys())
    assert len(tr.G_flat.params) == len(OM.g_var_names(P))
    assert len(tr.D_flat.params) == len(OM.d_var_names(P))
    assert tr.G_flat.numel >= sum(P.p[n].numel() for n in OM.g_var_names(P))


def test_forward_activations(dev):
    tr, gb, P, ob, OM = _setup(dev)
    with torch.no_grad():
        embs_o, G_o = OM.stage1_forward(P, ob, hidden_num=HID, z_num=ZNUM)
        d_real_o = OM.dcgan_discriminator(P, ob["x"])
        embs, _ = tr.encode(gb)
        G, _ = tr.generate(embs, gb["pose"])
        d_real = tr.discriminate(gb["x"])
    assert _rel(embs, embs_o) < 1e-4
    assert _rel(G, G_o) < 1e-4
    assert _rel(d_real, d_real_o) < 1e-3


def test_forward_and_trunk_gradients_on_converter_style_inputs(dev):
    """The same checks on a batch whose mask / boxes / visibility come from keypoints through the reference converter's rules
    (`dataprep`, pinned by tests/golden/prep_reference.npz): boxes that hug the limbs and touch the image border, sentinel boxes
    for the leg-less figure (index 3), a connected body mask -- the geometry the records really hold."""
    import dpig_amd.tflib as lib
    tr, gb, P, ob, OM = _setup(dev, B=4, seed=8, from_keypoints=True)
    assert float(gb["part_vis"][3, 2]) == 0 and gb["part_bbox"][3, 2].tolist() == [0, 0, 1, 1]
    gnames = OM.g_var_names(P)
    g = torch.Generator().manual_seed(6)
    r = torch.randn(tuple(ob["x"].shape), generator=g, dtype=torch.float64)
    embs_o, G_o = OM.stage1_forward(P, ob, hidden_num=HID, z_num=ZNUM)
    ggrads = dict(zip(gnames, torch.autograd.grad((G_o * r).sum(), [P.p[n] for n in gnames])))
    tr.G_flat.zero_grad()
    embs, _ = tr.encode(gb)
    G, _ = tr.generate(embs, gb["pose"])
    assert _rel(embs, embs_o) < 1e-4 and _rel(G, G_o) < 1e-4
    G.backward(r.float().to(dev))
    tr.G_flat.finalize()
    # ReLU kinks (module docstring): on this batch ONE unit of the ROI tower sits within fp32 round-off of zero and an fp32
    # evaluation of the ORACLE ITSELF moves Conv_6's gradient by 1.1 % (scripts/diag_kp.py).  So the bar per tensor is the
    # single-flip bound, or -- where the oracle's own fp32 evaluation is further than that from its fp64 one -- 1.5x that distance.
    P32 = OM.ParamStore(seed=11)
    P32.p = {n: v.detach().float().requires_grad_(v.requires_grad) for n, v in P.p.items()}
    P32.trainable = dict(P.trainable)
    ob32 = {k: (v.float() if v.is_floating_point() else v) for k, v in ob.items()}
    _, G32 = OM.stage1_forward(P32, ob32, hidden_num=HID, z_num=ZNUM)
    g32 = dict(zip(gnames, torch.autograd.grad((G32 * r.float()).sum(), [P32.p[n] for n in gnames])))
    # Which unit flips is evaluation-specific (the HIP sums run in a different order than the CPU's), so a tensor the fp32
    # oracle gets exactly may carry a flip here: per tensor 5e-3 (B=4: twice the units of the B=2 tests), the bulk far below.
    errs = []
    for n in gnames:
        own = _rel(g32[n], ggrads[n])
        errs.append(_rel(lib._params[n]._dpig_grad, ggrads[n]))
        assert errs[-1] < max(5e-3, 1.5 * own), (n, errs[-1], own)
    assert sorted(errs)[len(errs) // 2] < 2e-4, sorted(errs)[len(errs) // 2]


def test_step_from_keypoints_equals_step_from_the_pose_map(dev):
    """A batch that carries the keypoints `pose_rcv` instead of the target map (what the records hold; the reference rasterises
    inside the graph, trainer.py:556-560) takes the same g_optim / d_optim: losses and the flat gradients equal those of the
    map-fed step, the generated image equals the oracle's."""
    import dpig_amd.tflib as lib
    from dpig_amd import synthetic
    tr, gb, P, ob, OM = _setup(dev)
    kb = synthetic.keypoints_only(gb)
    assert "pose" not in kb and kb["pose_rcv"].shape == (2, 54)
    with torch.no_grad():
        _, G_o = OM.stage1_forward(P, ob, hidden_num=HID, z_num=ZNUM)
        embs, _ = tr.encode(kb)
        from dpig_amd.trainer import pose_input
        G, _ = tr.generate(embs, pose_input(kb, 128, 64))
    assert _rel(G, G_o) < 1e-4
    o_map = tr._g_optim_eager(gb, update=False)
    g_map = tr.G_flat.grad.clone()
    o_kp = tr._g_optim_eager(kb, update=False)
    g_kp = tr.G_flat.grad.clone()
    assert abs(o_map["g_loss"].item() - o_kp["g_loss"].item()) < 1e-5 * abs(o_map["g_loss"].item())
    assert (g_map - g_kp).abs().max().item() < 2e-3 * g_map.abs().max().item()           # (kinks: module docstring)
    assert ((g_map - g_kp).abs() <= 2e-4 * g_map.abs().max()).float().mean().item() > 0.995
    d_map = tr._d_optim_eager(gb, update=False)
    d_kp = tr._d_optim_eager(kb, update=False)
    assert abs(d_map["d_loss"].item() - d_kp["d_loss"].item()) < 1e-5 * abs(d_map["d_loss"].item())
    lib.delete_all_params()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_input_mask_fusion_is_bit_identical_and_active(dev, dtype):
    """A ReLU conv feeding a residual block: the block's input-gradient dgrad applies the conv's ReLU mask in its epilogue and the
    conv's backward skips its activation-gradient pass (autograd.FUSE_INPUT_MASK).  Masking in the epilogue or in a separate pass
    multiplies the same stored values by 0 / 1: the flat gradients must be IDENTICAL with the fusion on and off, and the fused run
    must launch fewer activation-gradient kernels."""
    import dpig_amd.tflib as lib
    from dpig_amd import autograd as A, hip_ops as H
    tr, gb, P, ob, OM = _setup(dev)
    tr.config.compute_dtype = dtype
    calls = [0]
    orig = H.act_bwd

    def spy(*a, **k):
        calls[0] += 1
        return orig(*a, **k)
    H.act_bwd = spy
    try:
        res = {}
        for fuse in (False, True):
            A.FUSE_INPUT_MASK[0] = fuse
            calls[0] = 0
            o = tr._g_optim_eager(gb, update=False)
            res[fuse] = (tr.G_flat.grad.clone(), o["g_loss"].item(), calls[0])
    finally:
        H.act_bwd = orig
        A.FUSE_INPUT_MASK[0] = True
        H.set_compute("f32")
    assert res[True][1] == res[False][1]
    assert torch.equal(res[True][0], res[False][0])
    assert res[True][2] <= res[False][2] - 10, (res[True][2], res[False][2])        # 3 towers x (stem + 4 stride-2 convs) at least
    lib.delete_all_params()


@pytest.mark.parametrize("dtype,graphs", [("f32", False), ("bf16", False), ("f32", True)])
def test_two_stream_towers_are_bit_identical(dev, dtype, graphs):
    """The ROI tower on a side stream beside the background branch (autograd.side_branch, models.py two-branch encoder): same
    kernels, same operands, a workspace per stream -- losses and every gradient / weight must be IDENTICAL to the one-stream run,
    eagerly (g_optim gradients, then a d_optim update) and through hipGraph replay (weights after two full steps)."""
    import dpig_amd.tflib as lib
    from dpig_amd import autograd as A, hip_ops as H
    res = {}
    defaults = (A.TWO_STREAM[0], A.WGRAD_STREAM[0], A.D_OVERLAP[0])
    try:
        for two in (False, True):
            A.TWO_STREAM[0] = two
            A.WGRAD_STREAM[0] = two                  # ... and the filter gradients on their own stream (autograd.wgrad_overlap)
            A.D_OVERLAP[0] = two                     # ... and the critic's real-image pass beside the generator forward (d_optim)
            tr, gb, P, ob, OM = _setup(dev)
            tr.config.compute_dtype = dtype
            if graphs:
                tr.enable_graphs(gb, gb)
                tr.step = 1
                outs = [tr.train_step(gb, gb) for _ in range(2)]
                torch.cuda.synchronize()
                res[two] = (tr.G_flat.flat.clone(), tr.D_flat.flat.clone(), outs[-1]["g_loss"].item(), outs[-1]["d_loss"].item())
            else:
                o = tr._g_optim_eager(gb, update=False)
                g = tr.G_flat.grad.clone()
                od = tr._d_optim_eager(gb, update=True)
                torch.cuda.synchronize()
                res[two] = (g, tr.D_flat.flat.clone(), o["g_loss"].item(), od["d_loss"].item())
            lib.delete_all_params()
    finally:
        A.TWO_STREAM[0], A.WGRAD_STREAM[0], A.D_OVERLAP[0] = defaults
        H.set_compute("f32")
    assert res[True][2] == res[False][2] and res[True][3] == res[False][3]
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])


def test_trunk_gradients_linear_readout(dev):
    """d/dtheta of <G, r> for a fixed random r: every E+G kernel's backward, no D, no |.| kink."""
    import dpig_amd.tflib as lib
    tr, gb, P, ob, OM = _setup(dev)
    gnames = OM.g_var_names(P)
    g = torch.Generator().manual_seed(5)
    r = torch.randn(tuple(ob["x"].shape), generator=g, dtype=torch.float64)
    _, G_o = OM.stage1_forward(P, ob, hidden_num=HID, z_num=ZNUM)
    ggrads = dict(zip(gnames, torch.autograd.grad((G_o * r).sum(), [P.p[n] for n in gnames])))
    tr.G_flat.zero_grad()
    embs, _ = tr.encode(gb)
    G, _ = tr.generate(embs, gb["pose"])
    G.backward(r.float().to(dev))
    tr.G_flat.finalize()
    worst = max(_rel(lib._params[n]._dpig_grad, ggrads[n]) for n in gnames)
    for n in gnames:
        assert lib._params[n].grad is None, "gradient of %s bypassed the flat sink" % n
    assert worst < 2e-3, worst


def test_g_and_d_gradients_and_adam_step(dev):
    import dpig_amd.tflib as lib
    tr, gb, P, ob, OM = _setup(dev)
    gnames, dnames = OM.g_var_names(P), OM.d_var_names(P)
    og = OM.OracleAdam(P, gnames, 2e-3)
    gl, aux = OM.stage1_g_loss(P, ob, hidden_num=HID, z_num=ZNUM)
    ggrads = dict(zip(gnames, torch.autograd.grad(gl, [P.p[n] for n in gnames], allow_unused=True)))
    out = tr.g_optim(gb)
    assert abs(out["g_loss"].item() - gl.item()) < 1e-4 * abs(gl.item())
    assert abs(out["L1Loss"].item() - aux["L1Loss"].item()) < 1e-5
    for n in gnames:
        if ggrads[n] is not None:
            assert _rel(lib._params[n]._dpig_grad, ggrads[n]) < 3e-2, n
    og.step(ggrads)
    for n in gnames:
        # TF-Adam's first step moves every element by ~lr*sign(g) (a sign step: discontinuous at
        # g = 0), so compare where the gradient is clearly non-zero, against the step size
        if ggrads[n] is None:
            continue
        big = ggrads[n].abs() > 0.1 * ggrads[n].abs().max()
        diff = (lib._params[n].detach().double().cpu() - P.p[n].detach()).abs()[big]
        assert diff.max().item() < 0.1 * 2e-3, (n, diff.max().item())
    # d_optim: make the two implementations start from the same G-side weights again
    with torch.no_grad():
        for n in gnames:
            lib._params[n].copy_(P.p[n].to(torch.float32))
    dl, _ = OM.stage1_d_loss(P, ob, hidden_num=HID, z_num=ZNUM)
    dgrads = dict(zip(dnames, torch.autograd.grad(dl, [P.p[n] for n in dnames], allow_unused=True)))
    out = tr.d_optim(gb)
    assert abs(out["d_loss"].item() - dl.item()) < 1e-4 * abs(dl.item())
    errs = {}
    for n in dnames:
        # the conv biases feeding a BatchNorm have an exactly-zero true gradient (BN removes the
        # mean): both sides hold round-off there, compare on an absolute floor instead
        got, ref = lib._params[n]._dpig_grad.double().cpu(), dgrads[n].double()
        floor = 1e-3 if n in ("Discriminator.2.Biases", "Discriminator.3.Biases", "Discriminator.4.Biases") else 1e-6
        errs[n] = (got - ref).abs().max().item() / max(ref.abs().max().item(), floor)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    assert worst[0][1] < 3e-2, worst


def test_train_step_order(dev):
    """trainer.py:336-347: g_optim is skipped at step 0; d_optim runs once per step in dcgan mode."""
    tr, gb, P, ob, OM = _setup(dev)
    o0 = tr.train_step(gb, gb)
    assert "g_loss" not in o0 and "d_loss" in o0
    o1 = tr.train_step(gb, gb)
    assert "g_loss" in o1 and "d_loss" in o1
    assert tr.g_opt.t == 1 and tr.d_opt.t == 2


def test_hipgraph_replay_matches_eager(dev):
    """The captured g_optim / d_optim graphs replay the same arithmetic as eager launches
    (device-side Adam step counter included): 3 training steps, same weights within round-off
    (the only non-deterministic kernel is the crop_and_resize scatter-add)."""
    import dpig_amd.tflib as lib
    res = []
    for use_graph in (False, True):
        tr, gb, P, ob, OM = _setup(dev)
        if use_graph:
            # the capture warm-up runs real optimizer steps and must put everything back: same weights, empty Adam
            # slots, step counters at zero (a run -- or a restored checkpoint -- is not moved by enabling graphs)
            snap = [t.clone() for t in (tr.G_flat.flat, tr.D_flat.flat)]
            tr.enable_graphs(gb, gb, warmup=1)
            for fl, s0 in zip((tr.G_flat, tr.D_flat), snap):
                assert torch.equal(fl.flat, s0) and float(fl.m.abs().sum()) == 0.0 and float(fl.v.abs().sum()) == 0.0
            for opt in (tr.g_opt, tr.d_opt):
                assert int(opt.state[0]) == 0 and opt.t == 0
        for _ in range(3):
            out = tr.train_step(gb, gb)
        torch.cuda.synchronize()
        res.append((tr.G_flat.flat.clone(), tr.D_flat.flat.clone(), out["g_loss"].item(), out["d_loss"].item(),
                    tr.g_opt.t, tr.d_opt.t, tr.g_opt.state[0].item()))
        lib.delete_all_params()
    (g0, d0, gl0, dl0, tg0, td0, s0), (g1, d1, gl1, dl1, tg1, td1, s1) = res
    assert (tg0, td0) == (tg1, td1) == (2, 3) and s0 == s1 == 2
    assert abs(gl0 - gl1) < 2e-3 * abs(gl0) and abs(dl0 - dl1) < 2e-3 * abs(dl0), (gl0, gl1, dl0, dl1)
    # TF-Adam's first steps are sign-like (|step| ~ lr whatever |g|): round-off on near-zero gradients --
    # here from the one non-deterministic kernel, the crop_and_resize scatter-add -- flips single elements by
    # up to 2*lr per step.  Compare in units of lr: no element further than 3 steps x 2 lr, mean far below lr.
    lr = 2e-3
    assert (g0 - g1).abs().max().item() <= 6 * lr and (g0 - g1).abs().mean().item() < 0.05 * lr
    assert (d0 - d1).abs().max().item() <= 6 * lr and (d0 - d1).abs().mean().item() < 0.05 * lr


def test_split_backward_equals_full_backward(dev):
    """The data-parallel schedule of g_optim (critic + decoder backward, [all-reduce of the decoder slice], encoder
    backward) produces bit-identical gradients to one backward pass, eagerly and as two replayed hipGraphs."""
    import dpig_amd.tflib as lib
    tr, gb, P, ob, OM = _setup(dev)
    assert 0 < tr._n_dec < len(tr.G_flat.params) and 0 < tr._enc_off < tr.G_flat.numel
    # stages in completion order: generator, background tower, ROI tower, stem -- together exactly the parameter list, each a
    # contiguous slice of the flat gradient buffer (creation order of the encoder: stem, ROI tower, background tower)
    assert [s[0] for s in tr._stages] == ["generator", "bg", "roi", "stem"]
    assert sorted(i for _, lo, hi in tr._stages for i in range(lo, hi)) == list(range(len(tr.G_flat.params)))
    assert tr._stages[3][1] == tr._n_dec and tr._stages[1][2] == len(tr.G_flat.params) and tr._stages[3][2] - tr._stages[3][1] == 6
    assert sum(tr._stage_slice(s).numel() for s in tr._stages) == tr.G_flat.numel
    grads = lambda: [p._dpig_grad.clone() for p in tr.G_flat.params]       # (the flat buffer also has alignment padding)
    same = lambda a, b: all(torch.equal(x, y) for x, y in zip(a, b))
    tr.config.split_backward = False
    tr._g_optim_eager(gb, update=False)
    ref = grads()
    assert sum(float(g.abs().sum()) for g in ref[:tr._n_dec]) > 0 and sum(float(g.abs().sum()) for g in ref[tr._n_dec:]) > 0
    tr.config.split_backward = True
    tr.G_flat.grad.fill_(7.0)                       # stale garbage must be overwritten
    tr._g_optim_eager(gb, update=False)
    bad = [(i, tr.G_flat.params[i].dpig_name, float((a - b).abs().max())) for i, (a, b) in enumerate(zip(grads(), ref)) if not torch.equal(a, b)]
    assert not bad, ([(st[0], sum(1 for b in bad if st[1] <= b[0] < st[2]), st[2] - st[1]) for st in tr._stages], bad[:3], bad[-3:])
    # graphs: forced split -> two graphs for g_optim, optimizer launches stay eager
    w0, d0 = tr.G_flat.flat.clone(), tr.D_flat.flat.clone()
    rng0 = torch.cuda.get_rng_state(dev)
    buffers0 = {n: t.clone() for n, t in lib._params.items() if not t.requires_grad}
    tr.enable_graphs(gb, gb, warmup=1)
    assert tr._gg2 is not None and len(tr._gg2) == 3 and not tr._graph_update       # one replayed graph per encoder stage
    # the warm-up steps leave no trace: weights, the device RNG stream and the non-trainable tensors are where they were
    assert torch.equal(tr.G_flat.flat, w0) and torch.equal(tr.D_flat.flat, d0)
    assert torch.equal(torch.cuda.get_rng_state(dev), rng0)
    assert buffers0 and all(torch.equal(lib._params[n], v) for n, v in buffers0.items())
    for fl, s0 in zip((tr.G_flat, tr.D_flat), (w0, d0)):
        fl.flat.copy_(s0); fl.m.zero_(); fl.v.zero_()
    for opt in (tr.g_opt, tr.d_opt):
        opt.state.zero_(); opt.t = 0
    tr.G_flat.grad.fill_(7.0)
    tr.g_optim(gb)
    torch.cuda.synchronize()
    assert same(grads(), ref)
    assert float((tr.G_flat.flat - w0).abs().max()) > 0
    lib.delete_all_params()


def test_checkpoint_save_restore_round_trip(dev, tmp_path):
    """trainer.py:366 / :180-212 through the TF-free V2 checkpoint code (tfckpt.py): a saved model restores in place
    into a trained trainer (flat buffers keep their addresses) and into a freshly built one via Config.ckpt_path."""
    import dpig_amd.tflib as lib
    from dpig_amd import slim, tfckpt
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
    tr, gb, P, ob, OM = _setup(dev)

    def outputs(t):
        with torch.no_grad():
            embs, _ = t.encode(gb)
            G, _ = t.generate(embs, gb["pose"])
            return G.clone(), t.discriminate(G).clone()

    tr.step = 1
    tr.train_step(gb, gb)
    G0, D0 = outputs(tr)
    prefix = tr.save_checkpoint(str(tmp_path))
    assert prefix.endswith("model.ckpt-2") and tfckpt.latest_checkpoint(str(tmp_path)) == prefix
    listed = {n: s for n, s, _ in tfckpt.list_variables(prefix)}
    assert int(tfckpt.load_checkpoint(prefix, ["step"])["step"]) == 2
    assert all(tuple(p.shape) == listed[lib.tf_variable_name(n)] for n, p in lib._params.items())
    assert "Discriminator.1/Discriminator.1.Filters" in listed and "g_lr" in listed and "d_lr" in listed
    flat_ptr = tr.G_flat.flat.data_ptr()
    tr.G_flat.flat.add_(0.05)
    tr.D_flat.flat.mul_(0.5)
    G1, _ = outputs(tr)
    assert not torch.equal(G1, G0)
    assert len(tfckpt.restore(prefix)) == len(lib._params)
    G2, D2 = outputs(tr)
    assert torch.equal(G2, G0) and torch.equal(D2, D0) and tr.G_flat.flat.data_ptr() == flat_ptr
    # a fresh process-like start: new registry, random init, restore through the config
    lib.delete_all_params()
    slim.reset_scopes()
    np.random.seed(123)
    cfg = Config(batch_size=2, conv_hidden_num=HID, z_num=ZNUM, ckpt_path=prefix)
    tr2 = DPIG_Encoder_GAN_BodyROI_FgBg(cfg, dev)
    tr2.init_net(gb)
    G3, D3 = outputs(tr2)
    assert torch.equal(G3, G0) and torch.equal(D3, D0)
    lib.delete_all_params()


def test_checkpoint_resume_with_optimizer_slots(dev, tmp_path):
    """A run saved with its Adam slots and resumed through Config(ckpt_path, restore_optimizer) takes the same next step
    as the run that never stopped (weights, moments and the device step counter all restored)."""
    import dpig_amd.tflib as lib
    from dpig_amd import slim, tfckpt
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
    tr, gb, P, ob, OM = _setup(dev)
    tr.step = 1
    tr.train_step(gb, gb)
    tr.train_step(gb, gb)
    prefix = tr.save_checkpoint(str(tmp_path), include_optimizer=True)
    names = {n for n, _, _ in tfckpt.list_variables(prefix)}
    assert {"beta1_power", "beta2_power", "beta1_power_1", "beta2_power_1", "Discriminator.1/Discriminator.1.Filters/Adam",
            "ID_AE/G/Conv/weights/Adam_1"} <= names
    # the saved key SET is the reference model's (tests/golden/ckpt_keys_model1.json, an independent reading of the reference's
    # graph code; names do not depend on the width)
    import json
    import os
    fix = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_keys_model1.json")))
    ours = {"dpig_amd/adam_step", "dpig_amd/adam_step_1"}      # this repo's extension (exact integer step counts); a TF Saver ignores them
    assert ours <= names
    assert names - ours == set(fix["keys"]), (sorted(names - ours - set(fix["keys"]))[:5], sorted(set(fix["keys"]) - names)[:5])
    tr.train_step(gb, gb)
    want_g, want_d = tr.G_flat.flat.clone(), tr.D_flat.flat.clone()
    lib.delete_all_params()
    slim.reset_scopes()
    np.random.seed(99)
    cfg = Config(batch_size=2, conv_hidden_num=HID, z_num=ZNUM, g_lr=2e-3, d_lr=2e-3, ckpt_path=prefix, restore_optimizer=True)
    tr2 = DPIG_Encoder_GAN_BodyROI_FgBg(cfg, dev)
    tr2.init_net(gb)
    assert tr2.g_opt.t == 2 and int(tr2.g_opt.state[0]) == 2 and int(tr2.d_opt.state[0]) == 2
    assert tr2.step == 3                      # the `step` variable of the checkpoint (trainer.py:47) came back too
    tr2.train_step(gb, gb)
    assert torch.equal(tr2.G_flat.flat, want_g) and torch.equal(tr2.D_flat.flat, want_d)
    lib.delete_all_params()
