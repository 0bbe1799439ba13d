"""Data parallelism end to end on the GPU box: two ranks x B=2 must take the step one process takes on the
concatenated batch of 4.  MODE='wgan-gp' is used because its critic normalises per sample (LayerNorm, wgan_gp.py:34-38): no
cross-sample coupling, so the equality holds without SyncBN (with the dcgan critic's BatchNorm the per-rank statistics differ
from the batch-4 ones: tests/test_syncbn_gpu.py covers that case).  The generator-side backward runs in its two stages with
the decoder slice's all-reduce in flight under the encoder's backward (trainer._g_backward_decoder / _encoder); the second
case moves the gradients as bf16 ('bf16' mode's exchange).  Both ranks share the one GPU and talk over gloo (production:
RCCL, same code)."""
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
B, HID, ZN, LR = 4, 16, 8, 2e-4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _alpha():
    return torch.rand(B, generator=torch.Generator().manual_seed(5))


def _worker(rank, world, port, q, exchange, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    if os.environ.get("DPIG_GUARD") in ("hi", "lo"):      # guard-page run of the suite (tests/conftest.py): the ranks' allocations too
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import conftest  # noqa: F401  (installs the allocator at import)
    ordinal = rank if backend == "nccl" else 0           # RCCL needs one device per rank; gloo ranks share cuda:0
    torch.cuda.set_device(ordinal)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from dpig_amd import synthetic
        from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
        dev = torch.device("cuda", ordinal)
        half = B // world
        pick = lambda b: {k: v[rank * half:(rank + 1) * half] for k, v in b.items()}
        np.random.seed(0)
        tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=half, conv_hidden_num=HID, z_num=ZN, gan_mode='wgan-gp', g_lr=LR,
                                                  d_lr=LR, grad_exchange=exchange), dev)
        bg = synthetic.to_device(pick(synthetic.make_batch(B, seed=21)), dev)
        bd = synthetic.to_device(pick(synthetic.make_batch(B, seed=22)), dev)
        tr.init_net(bg)
        assert tr.allreduce.enabled and tr._split() and tr.allreduce.compress == (None if exchange == 'f32' else 'bf16')
        tr.gp_alpha = _alpha()[rank * half:(rank + 1) * half].to(dev)
        og = tr.g_optim(bg)
        od = tr.d_optim(bd)
        torch.cuda.synchronize()
        q.put((rank, dict(d_loss=float(od["d_loss"]), g_loss=float(og["g_loss"]),
                          D=tr.D_flat.flat.detach().cpu().numpy().copy(), G=tr.G_flat.flat.detach().cpu().numpy().copy())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend,exchange", [("gloo", "f32"), ("gloo", "bf16"), ("nccl", "f32"), ("nccl", "bf16")])
def test_two_rank_step_equals_single_process_on_the_concatenated_batch(dev, backend, exchange):
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("the RCCL variant needs two devices (the driver's 8-GPU node); the gloo variant covers the same code on one")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, exchange, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
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
    np.random.seed(0)
    tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, conv_hidden_num=HID, z_num=ZN, gan_mode='wgan-gp', g_lr=LR, d_lr=LR), dev)
    bg = synthetic.to_device(synthetic.make_batch(B, seed=21), dev)
    bd = synthetic.to_device(synthetic.make_batch(B, seed=22), dev)
    tr.init_net(bg)
    tr.gp_alpha = _alpha().to(dev)
    D0, G0 = tr.D_flat.flat.detach().cpu().clone(), tr.G_flat.flat.detach().cpu().clone()
    og = tr.g_optim(bg)
    od = tr.d_optim(bd)
    torch.cuda.synchronize()
    # each rank reports the mean over its half: their average is the batch-4 mean
    d_loss = 0.5 * (res[0]["d_loss"] + res[1]["d_loss"])
    g_loss = 0.5 * (res[0]["g_loss"] + res[1]["g_loss"])
    # (the critic loss is taken AFTER the generator's update: with the bf16 exchange the weights whose summed gradient sits at the
    # rounding noise floor move by +-lr the other way, and the penalty-dominated loss of a random-init critic (~340) sees that)
    tol = 2e-4 if exchange == "f32" else 1e-3
    assert abs(d_loss - float(od["d_loss"])) < tol * max(1.0, abs(float(od["d_loss"])))
    assert abs(g_loss - float(og["g_loss"])) < tol * max(1.0, abs(float(og["g_loss"])))
    # Adam's first step moves every weight by lr * g / (|g| + eps) ~ lr * sign(g): the two-rank update must equal the
    # single-process one within 0.05 * lr on >= 99.5 % of the weights whose gradient is above the fp32 summation noise floor
    # (|g| > 1e-4 of the side's largest gradient; below it the SIGN of g -- hence the whole +-lr update -- is decided by the
    # order the partial sums are added in, on one GPU as much as on two).  The bf16 exchange rounds the summed gradient to 8
    # significand bits, which cannot flip a sign: the same bar holds.
    for name, new, old, flat in (("D", res[0]["D"], D0, tr.D_flat), ("G", res[0]["G"], G0, tr.G_flat)):
        ref, grad = flat.flat.detach().cpu(), flat.grad.detach().cpu()
        close = ((new - old) - (ref - old)).abs() <= 0.05 * LR
        live = grad.abs() > 1e-4 * grad.abs().max()
        frac_live, frac_all = close[live].float().mean().item(), close.float().mean().item()
        print("ddp %s/%s %s: %.4f of the weights above the noise floor (%.1f %% of all) within 0.05 lr; %.4f of all"
              % (backend, exchange, name, frac_live, 100.0 * live.float().mean().item(), frac_all))
        assert frac_live >= 0.995, (name, exchange, frac_live)
        assert frac_all > (0.97 if exchange == "f32" else 0.93), (name, exchange, frac_all)
    lib.delete_all_params()
    slim.reset_scopes()


def _stage2_worker(rank, world, port, q):
    """Stage-II trainer (model 3: frozen encoder + FC mappers / critics, MODE 'wgan') on two ranks: two eager steps, then the same two
    steps replayed from captured graphs.  Under data parallelism the graphs end with the backward pass; the gradient exchange, the
    optimizer step and the weight clip follow eagerly (trainer_stage2._update)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dpig_amd.tflib as lib
        from dpig_amd import slim, synthetic
        from dpig_amd.trainer import Config
        from dpig_amd.trainer_stage2 import DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI
        dev = torch.device("cuda:0")
        half = B // world
        out = {}
        for mode in ("eager", "graphs"):
            lib.delete_all_params(); slim.reset_scopes()
            np.random.seed(0)
            torch.manual_seed(100 + rank); torch.cuda.manual_seed(100 + rank)        # (the samplers' noise: per rank, same in both modes)
            tr = DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI(Config(batch_size=half, conv_hidden_num=16, g_lr=1e-3, d_lr=1e-3), dev)
            b = synthetic.to_device({k: v[rank * half:(rank + 1) * half] for k, v in synthetic.make_batch(B, seed=21).items()}, dev)
            tr.init_net(b)
            assert tr.allreduce.enabled
            tr.step = 1
            if mode == "graphs":
                tr.enable_graphs(b, warmup=1)
                assert tr._graph_update is False                     # exchange + update stay outside the captures
                b = tr.static_batch()
            losses = []
            for _ in range(2):
                o = tr.train_step(b)
                losses.append({k: float(v) for k, v in o.items()})
            torch.cuda.synchronize()
            out[mode] = (losses, {s: [f.flat.detach().cpu().numpy().copy() for f in tr.flats[s]] for s in ("fg", "bg")})
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_stage2_two_rank_captured_steps_equal_eager(dev):
    """Found by `scripts/run_scale.sh --dry` (round 6): the stage-II trainer captured its gradient all-reduce inside the hipGraphs and a
    two-rank launch died in capture_end -- configs[2] (8 x MI355X) would have failed on the 8-GPU day.  Replayed steps must equal the
    eager ones bit for bit on both ranks, and the replicas must stay identical."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stage2_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        (le, we), (lg, wg) = res[r]["eager"], res[r]["graphs"]
        assert le == lg, (le, lg)
        for s in ("fg", "bg"):
            assert all(np.array_equal(a, b_) for a, b_ in zip(we[s], wg[s]))
    for s in ("fg", "bg"):
        assert all(np.array_equal(a, b_) for a, b_ in zip(res[0]["graphs"][1][s], res[1]["graphs"][1][s]))
