"""The data-parallel gradient exchange on 2 processes over gloo (CPU): bucketed all-reduce of the
flat gradient buffer, 1/world folded into the optimizer's grad_scale, rank-0 broadcast of the
initial weights.  (On the GPU box the same code runs over RCCL: backend "nccl".)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from dpig_amd.trainer import FlatParams, GradAllReduce
        torch.manual_seed(rank)                            # ranks start with DIFFERENT weights
        p = torch.nn.Parameter(torch.randn(1000))
        fp = FlatParams([p])
        ar = GradAllReduce(bucket_bytes=1024)              # 256 floats per bucket -> 4 collectives
        assert ar.enabled and ar.world == world
        ar.broadcast(fp.flat)
        fp.grad[:1000].copy_(torch.arange(1000, dtype=torch.float32) * (rank + 1))
        scale = ar(fp.grad)
        exact = fp.grad[:1000].numpy().copy()
        # bf16 exchange ('bf16' mode): staged through a persistent bf16 twin of the flat buffer, two overlapping slices
        # in flight at once (the decoder / encoder halves of the generator-side update), widened back into the fp32 buffer
        enc = lambda s, d: d.copy_(s.to(torch.bfloat16))
        dec = lambda s, d: d.copy_(s.float())
        arc = GradAllReduce(bucket_bytes=512, compress='bf16', codec=(enc, dec))
        fp.grad[:1000].copy_(torch.arange(1000, dtype=torch.float32) * 0.37 * (rank + 1))
        h = arc.start(fp.grad[:600])
        h += arc.start(fp.grad[600:])
        scale_c = arc.finish(h)
        q.put((rank, fp.flat[:1000].detach().numpy().copy(), exact, scale, fp.grad[:1000].numpy().copy(), scale_c))   # by value
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_and_broadcast_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, w0, g0, s0, c0, sc0), (r1, w1, g1, s1, c1, sc1) = [(r, torch.from_numpy(w), torch.from_numpy(g), sc, torch.from_numpy(c), scc)
                                                              for r, w, g, sc, c, scc in res]
    assert torch.equal(w0, w1)                                           # broadcast from rank 0
    expect = torch.arange(1000, dtype=torch.float32) * 3.0               # (1 + 2) * arange
    assert torch.equal(g0, expect) and torch.equal(g1, expect)
    assert s0 == s1 == 0.5                                               # mean = sum * 1/world in Adam
    # bf16 exchange: every rank ends with the same bf16-valued sums, within bf16 rounding of the exact sum
    a = torch.arange(1000, dtype=torch.float32) * 0.37
    want = (a.to(torch.bfloat16) + (2 * a).to(torch.bfloat16)).float()
    assert torch.equal(c0, c1) and sc0 == sc1 == 0.5
    assert (c0 - 3 * a).abs().max() <= 2.0 ** -7 * (3 * a).abs().max() and (c0 - want).abs().max() <= 2.0 ** -7 * want.abs().max()


def test_allreduce_is_identity_without_process_group():
    from dpig_amd.trainer import GradAllReduce
    ar = GradAllReduce()
    g = torch.ones(10)
    assert ar(g) == 1.0 and not ar.enabled and float(g.sum()) == 10.0
