"""Host-side logic that needs no GPU: the tflib parameter registry, TF-slim variable naming,
synthetic batch generator, flat-parameter gradient sinks."""
import numpy as np
import pytest
import torch


def test_tflib_param_registry_semantics():
    import dpig_amd.tflib as lib
    lib.delete_all_params()
    lib.set_device("cpu")
    a = lib.param("Discriminator.1.Filters", np.ones((2, 2), dtype="float32"))
    b = lib.param("Discriminator.1.Filters", np.zeros((2, 2), dtype="float32"))       # same name -> same tensor
    assert a is b and float(a.sum()) == 4.0 and a.requires_grad
    mm = lib.param("Discriminator.BN2.moving_mean", np.zeros(3, dtype="float32"), trainable=False)
    assert not mm.requires_grad
    lib.param("Generator.x", np.zeros(1, dtype="float32"))
    # substring selection incl. the non-trainable BN stats (trainer.py:603, SURVEY C-7)
    sel = lib.params_with_name("Discriminator.")
    assert len(sel) == 2 and any(p is mm for p in sel)
    c = lib.param("alias.target", np.full(1, 7.0, dtype="float32"))
    lib.alias_params({a: c})
    assert lib.param("Discriminator.1.Filters") is c
    lib.delete_param_aliases()
    assert lib.param("Discriminator.1.Filters") is a
    with pytest.raises(Exception):
        lib.param("never.created")
    lib.delete_all_params()
    assert lib.params_with_name("") == []
    lib.set_device(None)


def test_slim_variable_naming_and_reuse():
    import dpig_amd.tflib as lib
    from dpig_amd import slim
    lib.delete_all_params()
    slim.reset_scopes()
    with slim.variable_scope("Encoder"):
        with slim.variable_scope("G_encoder") as vs:
            assert vs == "Encoder/G_encoder"
            names = [slim._unique("Conv") for _ in range(3)] + [slim._unique("fully_connected") for _ in range(2)]
    assert names == ["Encoder/G_encoder/Conv", "Encoder/G_encoder/Conv_1", "Encoder/G_encoder/Conv_2",
                     "Encoder/G_encoder/fully_connected", "Encoder/G_encoder/fully_connected_1"]
    with slim.variable_scope("Encoder"):
        with slim.variable_scope("G_encoder", reuse=True):
            assert slim._unique("Conv") == "Encoder/G_encoder/Conv"          # counters reset on re-entry
    with slim.variable_scope("ID_AE"):
        with slim.variable_scope("G"):
            assert slim._unique("Conv") == "ID_AE/G/Conv"
    lib.set_device("cpu")
    np.random.seed(0)
    w, b = slim._conv_vars("S/Conv", 3, 4, 8)
    lim = np.sqrt(6.0 / (9 * 4 + 9 * 8))
    assert tuple(w.shape) == (3, 3, 4, 8) and float(w.abs().max()) <= lim and float(b.abs().sum()) == 0.0
    assert slim.get_variables("S") == [w, b]
    lib.delete_all_params()
    lib.set_device(None)


def test_synthetic_batch_is_deterministic_and_well_formed():
    from dpig_amd import synthetic
    a, b = synthetic.make_batch(3, seed=5), synthetic.make_batch(3, seed=5)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    assert a["x"].shape == (3, 128, 64, 3) and a["x"].dtype == np.float32 and abs(a["x"]).max() <= 1.0
    assert a["pose"].shape == (3, 128, 64, 18) and set(np.unique(a["pose"])) <= {-1.0, 1.0}
    assert set(np.unique(a["mask_r6"])) <= {0.0, 1.0} and 0.1 < a["mask_r6"].mean() < 0.7
    bb, vis = a["part_bbox"], a["part_vis"]
    assert bb.shape == (3, 7, 4) and vis.shape == (3, 7)
    for i in range(3):
        for p in range(7):
            y1, x1, y2, x2 = bb[i, p]
            if vis[i, p] == 0:
                assert list(bb[i, p]) == [0, 0, 1, 1]                      # invisible-part sentinel
            else:
                assert 0 <= y1 < y2 <= 127 and 0 <= x1 < x2 <= 63 and y2 - y1 >= 8 and x2 - x1 >= 8
    assert len(synthetic._STENCIL) == 49 and len(set(synthetic._STENCIL)) == 49


def test_flat_params_and_gradient_sink_on_cpu():
    from dpig_amd import autograd as A
    from dpig_amd.trainer import FlatParams
    p1 = torch.nn.Parameter(torch.arange(6, dtype=torch.float32).reshape(2, 3))
    p2 = torch.nn.Parameter(torch.ones(5))
    buf = torch.zeros(3)                                   # non-trainable entry is skipped
    fp = FlatParams([p1, p2, buf, p1])
    assert fp.numel == 8 + 8 and len(fp.params) == 2
    assert fp.offsets == [0, 8]                            # every tensor starts on a 16-byte boundary
    assert torch.equal(fp.flat[:6], torch.arange(6, dtype=torch.float32))
    p1.data.mul_(2)                                        # parameter storage IS the flat buffer
    assert float(fp.flat[5]) == 10.0
    calls = []

    def fake_kernel(out, beta):
        calls.append(beta)
        if beta == 0.0:
            out.copy_(torch.ones_like(out))
        else:
            out.add_(torch.ones_like(out))
        return out

    fp.zero_grad()
    assert A._sink(p1, fake_kernel) is None and A._sink(p1, fake_kernel) is None
    assert calls == [0.0, 1.0] and float(fp.grad[:6].sum()) == 12.0
    fp.finalize()                                          # p2 untouched -> zeros
    assert float(fp.grad[8:13].abs().sum()) == 0.0
    assert A._sink_small(p2, torch.full((5,), 3.0)) is None and float(fp.grad[8:13].sum()) == 15.0
    fp.zero_grad()                                         # first touch after zero_grad overwrites
    A._sink(p1, fake_kernel)
    assert calls[-1] == 0.0
    q = torch.nn.Parameter(torch.zeros(2))                 # no sink registered -> plain autograd return
    assert A._sink(q, lambda out, beta: torch.ones(2)) is not None


def test_config_matches_reference_defaults():
    from dpig_amd.trainer import Config
    c = Config()
    assert (c.batch_size, c.img_H, c.img_W, c.conv_hidden_num, c.z_num, c.repeat_num) == (16, 128, 64, 128, 64, 5)
    assert Config(img_H=256, img_W=256).repeat_num == 6


def test_packed_batch_layout_round_trip():
    """prefetch.PackedLayout: fields 256-byte aligned inside one flat buffer, views carry dtype and shape, a batch of a
    different structure is not mistaken for the same layout."""
    import torch
    from dpig_amd.prefetch import PackedLayout
    b = {"x": torch.randn(4, 33, 17, 3), "ids": torch.randint(0, 1 << 30, (4, 7), dtype=torch.int32),
         "vis": torch.rand(4, 7), "empty": torch.zeros(0, 3)}
    L = PackedLayout(b)
    assert all(off % 256 == 0 for _, _, _, off, _ in L.fields) and L.nbytes % 256 == 0
    flat = torch.full((L.nbytes,), 0xAB, dtype=torch.uint8)
    L.pack(b, flat)
    out = L.views(flat.clone())
    assert all(out[k].dtype == b[k].dtype and torch.equal(out[k], b[k]) for k in b)
    assert L.matches(b) and not L.matches({k: v for k, v in b.items() if k != "vis"})
    assert not L.matches(dict(b, x=b["x"].double()))


def test_side_branch_and_wgrad_overlap_are_plain_blocks_without_a_gpu():
    """autograd.side_branch / wgrad_overlap (independent sub-graphs on side HIP streams) must be inert for host tensors: same
    objects out, no stream API touched, switches restored."""
    import torch
    from dpig_amd import autograd as A
    x = torch.randn(3, 4)
    with A.side_branch(x) as sb:
        assert not sb.active
        y = x * 2
    assert sb.join(y) is y
    a, b = sb.join(y, x)
    assert a is y and b is x
    with A.side_branch(x, key="critic", enabled=True) as sb2:
        assert not sb2.active                              # enabled, but not a GPU tensor
    on = A._WG["on"]
    with A.wgrad_overlap():
        assert A._WG["on"] == bool(A.WGRAD_STREAM[0])
        with A.wgrad_overlap():                            # nested: the outer block owns the join
            pass
    assert A._WG["on"] == on and not A._WG["used"]


def test_workspace_is_per_device_and_stream_key():
    """_lib._Workspace: grow-only, one buffer per key; a pinned (graph-captured) buffer that must grow is retired, not freed."""
    import torch
    from dpig_amd._lib import _Workspace
    ws = _Workspace()
    dev = torch.device("cpu")
    assert ws.get(0, dev) == (None, 0)
    b1, n1 = ws.get(100, dev)
    assert n1 >= 100 and ws.get(50, dev)[0] is b1                     # re-used while it is large enough
    ws.pin()
    b2, n2 = ws.get(n1 + 1, dev)
    assert b2 is not b1 and n2 > n1 and any(r is b1 for r in ws.retired)


def test_halo_staged_kloop_ordering_model():
    """The k-loops of bhq_kernel, bq_kernel<2,4> / <4,2> (csrc/dpig_conv_bf16_q.hip) and bwq_kernel (dpig_conv_bf16_wq.hip) as executable ordering models (scripts/ubench/simulate_kloop_hazards.py):
    the eight waves' fragment reads, LDS-DMA issues, counted vmcnt waits and barriers under random schedules with early and late DMA
    landings -- every read must see its own k-tile's / chunk's data with no DMA in flight to that LDS region; a wait relaxed by one piece
    must be caught (the model has teeth).  A regression guard for edits of the schedule; the kernel itself is tested on the GPU."""
    import importlib.util
    import os
    import pytest
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "ubench", "simulate_kloop_hazards.py")
    spec = importlib.util.spec_from_file_location("simulate_kloop_hazards", path)
    sim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sim)
    for nch in (1, 2, 3):
        for seed in range(16):
            lazy = (0.98, 0.5, 0.1, 0.02)[seed % 4]
            sim.run(nch, seed, lazy=lazy, make=sim.program_bhq)
            sim.run(nch, seed, lazy=lazy, make=lambda w, c: sim.program_bhq(w, c, slack=True))
    with pytest.raises(AssertionError):
        for seed in range(200):
            sim.run(3, seed, lazy=0.02, make=lambda w, c: sim.program_bhq(w, c, pb_wait=5))
    for (WM, WN) in ((2, 4), (4, 2)):                                        # bq_kernel<2,4> / <4,2>: operands gathered per tap
        for nkt in (2, 3, 9):
            for seed in range(8):
                sim.run(nkt, seed, lazy=(0.98, 0.5, 0.1, 0.02)[seed % 4], make=lambda w, k: sim.program_bq(w, k, WM, WN))
    with pytest.raises(AssertionError):
        for seed in range(200):
            sim.run(9, seed, lazy=0.02, make=lambda w, k: sim.program_bq(w, k, 2, 4, relax=1))
    for (WM, WN) in ((2, 4), (4, 2)):                                        # bwq_kernel<2,4> / <4,2> (wgrad: units cut along k)
        for nkt in (2, 3, 9):
            for seed in range(8):
                sim.run(nkt, seed, lazy=(0.98, 0.5, 0.1, 0.02)[seed % 4], make=lambda w, k: sim.program_bwq(w, k, WM, WN))
    with pytest.raises(AssertionError):
        for seed in range(200):
            sim.run(9, seed, lazy=0.02, make=lambda w, k: sim.program_bwq(w, k, 2, 4, relax=1))


def test_encoder_stage_of_files_only_variables_it_can_parse():
    """`models.encoder_stage_of`: TF-slim leaves `Conv[_k]` / `fully_connected[_k]` of the marked scope are filed by the scope
    counters; anything else -- another scope, an unknown layer kind, a non-numeric suffix -- is None (the trainer then keeps ONE
    'encoder' stage) instead of raising or being compared against the wrong counter; a snapshot of the marks is honoured."""
    from dpig_amd import models
    marks = {"E.towers_in": ("Encoder/G_encoder", 3, 0), "E.bg_begin": ("Encoder/G_encoder", 17, 1)}
    f = lambda n: models.encoder_stage_of(n, marks)            # noqa: E731
    assert f("Encoder/G_encoder/Conv/weights") == "stem" and f("Encoder/G_encoder/Conv_2/biases") == "stem"
    assert f("Encoder/G_encoder/Conv_3/weights") == "roi" and f("Encoder/G_encoder/Conv_16/weights") == "roi"
    assert f("Encoder/G_encoder/fully_connected/weights") == "roi"
    assert f("Encoder/G_encoder/Conv_17/weights") == "bg" and f("Encoder/G_encoder/fully_connected_1/biases") == "bg"
    assert f("Encoder/G_encoder/Conv_x/weights") is None         # non-numeric suffix: used to raise ValueError
    assert f("Encoder/G_encoder/LayerNorm/gamma") is None        # unknown layer kind: used to be filed by the FC counter
    assert f("Encoder/G_encoder/fully_connectedX/weights") is None
    assert f("ID_AE/G/Conv_3/weights") is None
    assert models.encoder_stage_of("Encoder/G_encoder/Conv/weights", {}) is None
