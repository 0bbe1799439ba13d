"""The captured two-rank SyncBN step (moved out of test_syncbn_gpu.py in round 6 so that it runs LAST in the suite: file names sort).
It is the one GPU test with a history of an intermittent, unexplained `Memory access fault by GPU` in a rank (rounds 4 and 5; see the test's
docstring) -- it runs in ONE attempt, and should a rank ever die again under `pytest -x`, every other test has already run and been counted."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_syncbn_gpu import ROOT, _free_port, _setup  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def _captured_worker(rank, world, port, q):
    """Two ranks, cross-rank batch-norm statistics, optimizer ops captured as chains of hipGraphs (autograd.SegmentedCapture) with the
    statistics' collectives and the staged gradient exchange between the replays."""
    _setup(rank, world, port)
    try:
        import faulthandler
        crash_dir = os.path.join(ROOT, "gpurun_out", "crash")          # a dying rank leaves its Python stacks here
        os.makedirs(crash_dir, exist_ok=True)
        faulthandler.enable(file=open(os.path.join(crash_dir, "test_captured_pair_r%d.log" % rank), "w"), all_threads=True)
        from dpig_amd import synthetic
        from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
        dev = torch.device("cuda:0")
        B = 4
        half = B // world
        pick = lambda b: {k: v[rank * half:(rank + 1) * half] for k, v in b.items()}     # noqa: E731
        out = {}
        for mode in ("eager", "graphs"):
            import dpig_amd.tflib as lib
            from dpig_amd import slim
            lib.delete_all_params(); slim.reset_scopes()
            np.random.seed(0)
            tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=half, conv_hidden_num=16, z_num=8, sync_bn=True), dev)
            bg = synthetic.to_device(pick(synthetic.make_batch(B, seed=21)), dev)
            bd = synthetic.to_device(pick(synthetic.make_batch(B, seed=22)), dev)
            tr.init_net(bg)
            tr.step = 1
            if mode == "graphs":
                tr.enable_graphs(bg, bd, warmup=1)
                assert type(tr._graphs[0]).__name__ == "SegmentedCapture"
            losses = []
            for _ in range(3):
                o = tr.train_step(bg, bd)
                losses.append((float(o["g_loss"]), float(o["d_loss"])))
            torch.cuda.synchronize()
            out[mode] = (losses, tr.D_flat.flat.detach().cpu().numpy().copy(), tr.G_flat.flat.detach().cpu().numpy().copy())
            dist.barrier()
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_captured_pair():
    """One attempt: two ranks on the one GPU; None if a rank died (reported at once, not after the queue's timeout)."""
    import queue as _queue
    import time
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_captured_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res, deadline = {}, time.time() + 300
    while len(res) < world and time.time() < deadline:
        try:
            r, v = q.get(timeout=2)
            res[r] = v
        except _queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    ok = len(res) == world
    for p in procs:
        p.join(timeout=60 if ok else 10)
        if p.is_alive():
            p.kill()
            p.join(timeout=10)
    return res if ok and all(p.exitcode == 0 for p in procs) else None


def test_two_rank_captured_sync_bn_steps_equal_eager(dev):
    """Three steps replayed as graph chains equal the three eager steps bit for bit on both ranks (losses and every weight), and the
    replicas stay identical -- two ranks sharing the box's one GPU over gloo.
    History: round 4 parked this configuration ("one rank died during the warm-up's gradient all-reduce in 2 of 5 attempts"); round 5 saw
    28 of 28 diagnostic attempts pass, then one full-suite run die the same way (`Memory access fault by GPU` in one rank) and wrapped the
    pair in a retry.  Round 6 took the retry out again: every kernel the step launches was run on guard pages -- each allocation its own
    mapping with unmapped pages on both sides, both placements, NaN-filled, the two ranks' allocations included
    (tests/test_guard_gpu.py, scripts/guard_suite.sh, DPIG_GUARD in scripts/diag_syncbn_graph_2rank.py) -- without one out-of-bounds
    access, and `scripts/diag_syncbn_graph_2rank.py 50 graphs` passed 50 of 50 (profiles/r06_two_rank_captured_50.txt).  ONE attempt:
    a rank that dies fails the test, with its faulthandler trace under gpurun_out/crash/."""
    world = 2
    res = _run_captured_pair()
    assert res is not None, "a rank of the captured two-rank SyncBN step died (no retry: see the docstring)"
    for r in range(world):
        le, De, Ge = res[r]["eager"]
        lg, Dg, Gg = res[r]["graphs"]
        assert le == lg, (le, lg)
        assert np.array_equal(De, Dg) and np.array_equal(Ge, Gg)
    assert np.array_equal(res[0]["graphs"][1], res[1]["graphs"][1]) and np.array_equal(res[0]["graphs"][2], res[1]["graphs"][2])
