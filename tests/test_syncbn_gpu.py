"""Synchronised batch norm across data-parallel ranks: two ranks with B/2 samples each must
reproduce the single-process batch-B result -- the op alone, and one full d_optim step of the stage-I trainer.
Both ranks share the one GPU of the test box and talk over gloo (the production backend is RCCL, same code)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    if os.environ.get("DPIG_GUARD") in ("hi", "lo"):      # guard-page run of the suite (tests/conftest.py): the ranks' allocations too
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import conftest  # noqa: F401  (installs the allocator at import)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _op_worker(rank, world, port, q):
    _setup(rank, world, port)
    try:
        from dpig_amd import autograd as A
        dev = torch.device("cuda:0")
        g = torch.Generator().manual_seed(11)
        N, Hh, W, C = 4, 6, 5, 40
        x_full = torch.randn(N, Hh, W, C, generator=g) * 2 + 0.5
        dy_full = torch.randn(N, Hh, W, C, generator=g)
        scale0 = torch.rand(C, generator=g) + 0.5
        offset0 = torch.randn(C, generator=g)
        half = N // world
        sl = slice(rank * half, (rank + 1) * half)
        A.set_sync_batchnorm(True)
        x = x_full[sl].to(dev).requires_grad_(True)
        sc, of = scale0.to(dev).requires_grad_(True), offset0.to(dev).requires_grad_(True)
        y = A.batchnorm(x, sc, of, 1e-5, 2, 0.2)
        y.backward(dy_full[sl].to(dev))
        out = dict(y=y.detach().cpu(), dx=x.grad.cpu(), ds=sc.grad.cpu(), do=of.grad.cpu())
        if rank == 0:                              # single-process reference on the whole batch (fused op)
            A.set_sync_batchnorm(False)
            xf = x_full.to(dev).requires_grad_(True)
            scf, off = scale0.to(dev).requires_grad_(True), offset0.to(dev).requires_grad_(True)
            yf = A.batchnorm(xf, scf, off, 1e-5, 2, 0.2)
            yf.backward(dy_full.to(dev))
            out.update(y_ref=yf.detach().cpu(), dx_ref=xf.grad.cpu(), ds_ref=scf.grad.cpu(), do_ref=off.grad.cpu())
        q.put((rank, {k: v.numpy() for k, v in out.items()}))      # by value: the producer may exit first
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sync_batchnorm_op_matches_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_op_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: {k: torch.from_numpy(v) for k, v in o.items()} for r, o in (q.get(timeout=240) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = res[0]
    y = torch.cat([res[0]["y"], res[1]["y"]])
    dx = torch.cat([res[0]["dx"], res[1]["dx"]])
    assert torch.allclose(y, ref["y_ref"], rtol=1e-5, atol=1e-5)
    assert torch.allclose(dx, ref["dx_ref"], rtol=1e-4, atol=1e-5)
    for r in range(world):                         # every rank contributes global/world; the DDP average restores it
        assert torch.allclose(res[r]["ds"] * world, ref["ds_ref"], rtol=1e-4, atol=1e-4)
        assert torch.allclose(res[r]["do"] * world, ref["do_ref"], rtol=1e-4, atol=1e-4)


def _train_worker(rank, world, port, q):
    _setup(rank, world, port)
    try:
        import faulthandler
        faulthandler.enable()
        from dpig_amd import synthetic
        from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
        dev = torch.device("cuda:0")
        B = 4
        full_g = synthetic.make_batch(B, seed=21)
        full_d = synthetic.make_batch(B, seed=22)
        half = B // world
        pick = lambda b: {k: v[rank * half:(rank + 1) * half] for k, v in b.items()}
        np.random.seed(0)
        tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=half, conv_hidden_num=16, z_num=8, sync_bn=True), dev)
        bg, bd = synthetic.to_device(pick(full_g), dev), synthetic.to_device(pick(full_d), dev)
        tr.init_net(bg)
        tr.step = 1
        out = tr.train_step(bg, bd)
        torch.cuda.synchronize()
        res = dict(d_loss=float(out["d_loss"]), g_loss=float(out["g_loss"]),
                   D=tr.D_flat.flat.detach().cpu().numpy().copy(), G=tr.G_flat.flat.detach().cpu().numpy().copy())
        q.put((rank, res))
        dist.barrier()
    except Exception:
        import traceback
        traceback.print_exc()
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_step_with_sync_bn_matches_single_process(dev):
    """One g_optim + d_optim of the stage-I trainer: 2 ranks x B=2 with SyncBN == 1 process x B=4 (same samples)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        res[r]["D"], res[r]["G"] = torch.from_numpy(res[r]["D"]), torch.from_numpy(res[r]["G"])
    assert torch.equal(res[0]["D"], res[1]["D"]) and torch.equal(res[0]["G"], res[1]["G"])   # replicas stay in sync

    from dpig_amd import slim, synthetic
    import dpig_amd.tflib as lib
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
    lib.delete_all_params()
    slim.reset_scopes()
    B = 4
    np.random.seed(0)
    tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, conv_hidden_num=16, z_num=8), dev)
    bg = synthetic.to_device(synthetic.make_batch(B, seed=21), dev)
    bd = synthetic.to_device(synthetic.make_batch(B, seed=22), dev)
    tr.init_net(bg)
    tr.step = 1
    D0 = tr.D_flat.flat.detach().cpu().clone()
    G0 = tr.G_flat.flat.detach().cpu().clone()
    out = tr.train_step(bg, bd)
    torch.cuda.synchronize()
    # losses: each rank reports its local mean; their average is the batch-4 mean
    d_loss = 0.5 * (res[0]["d_loss"] + res[1]["d_loss"])
    g_loss = 0.5 * (res[0]["g_loss"] + res[1]["g_loss"])
    assert abs(d_loss - float(out["d_loss"])) < 2e-4 * max(1.0, abs(float(out["d_loss"])))
    assert abs(g_loss - float(out["g_loss"])) < 2e-4 * max(1.0, abs(float(out["g_loss"])))
    # parameter updates: Adam's first step moves every weight by ~lr*sign(g); compare the updates where the
    # gradient is not at the rounding floor (same criterion as tests/test_model_gpu.py)
    for name, new, old, ref in (("D", res[0]["D"], D0, tr.D_flat.flat.detach().cpu()),
                                ("G", res[0]["G"], G0, tr.G_flat.flat.detach().cpu())):
        upd, upd_ref = new - old, ref - old
        lr = 2e-5
        close = (upd - upd_ref).abs() <= 0.05 * lr
        assert close.float().mean() > 0.97, (name, close.float().mean())


def _segmented_worker(q):
    """One process, a world-size-1 gloo group and the cross-rank batch-norm op FORCED (its three all-reduces per layer and pass are
    then identity collectives): the trainer's optimizer ops captured as chains of hipGraphs with the collectives between them."""
    import faulthandler
    faulthandler.enable()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from dpig_amd import autograd as A
        from dpig_amd import synthetic
        from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
        dev = torch.device("cuda:0")
        A.batchnorm = lambda x, scale, offset, eps=1e-5, act=0, alpha=0.2, stats=None: A._SyncBatchNormFn.apply(x, scale, offset, eps, act, alpha, None)
        out = {}
        for split in (False, True):
            import dpig_amd.tflib as lib
            from dpig_amd import slim
            lib.delete_all_params(); slim.reset_scopes()
            np.random.seed(0)
            B = 2
            tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, conv_hidden_num=16, z_num=8, sync_bn=True, split_backward=split), dev)
            bg = synthetic.to_device(synthetic.make_batch(B, seed=21), dev)
            bd = synthetic.to_device(synthetic.make_batch(B, seed=22), dev)
            tr.init_net(bg)
            tr.step = 1
            tr._sync_bn_active = lambda: True
            snap = [(f.flat.clone(), f.m.clone(), f.v.clone()) for f in (tr.G_flat, tr.D_flat)]
            e = tr.train_step(bg, bd)
            torch.cuda.synchronize()
            eager = (float(e["g_loss"]), float(e["d_loss"]), tr.G_flat.flat.clone(), tr.D_flat.flat.clone())
            with torch.no_grad():
                for f, (w, m, v) in zip((tr.G_flat, tr.D_flat), snap):
                    f.flat.copy_(w); f.m.copy_(m); f.v.copy_(v)
                for o in (tr.g_opt, tr.d_opt):
                    o.state.zero_(); o.t = 0
            tr.step = 1
            tr.enable_graphs(bg, bd, warmup=1)
            gg, gd = tr._graphs[0], tr._graphs[2]
            r = tr.train_step(bg, bd)
            torch.cuda.synchronize()
            out[split] = dict(kind=type(gg).__name__, segments=(gg.segments, gd.segments), stages=0 if tr._gg2 is None else len(tr._gg2),
                              same=bool(float(r["g_loss"]) == eager[0] and float(r["d_loss"]) == eager[1] and
                                        torch.equal(tr.G_flat.flat, eager[2]) and torch.equal(tr.D_flat.flat, eager[3])))
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_optimizer_ops_with_collectives_inside_replay_as_chains_of_graphs():
    """autograd.SegmentedCapture: with cross-rank batch-norm statistics an optimizer op cannot be ONE hipGraph (the statistics' all-reduces
    sit in the middle of the critic's forward and backward passes).  It is captured as a chain of graphs with the collectives between
    them -- also where the collective is reached from the autograd engine's thread inside a backward pass.  g_optim: critic forward
    (3 BN layers x 2 statistics) + backward (3 x 1) = 9 all-reduces -> 10 graphs; d_optim: two critic passes -> 19; with the staged
    data-parallel backward the encoder's three stages follow as graphs of their own.  A replayed step equals the eager step bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_segmented_worker, args=(q,))
    p.start()
    out = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0
    for split in (False, True):
        o = out[split]
        assert o["kind"] == "SegmentedCapture" and o["segments"] == (10, 19), o
        assert o["stages"] == (3 if split else 0) and o["same"], o
