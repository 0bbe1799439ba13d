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
        q.put((rank, fp.flat[:1000].detach().numpy().copy(), fp.grad[:1000].numpy().copy(), scale))   # by value
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
    (r0, w0, g0, s0), (r1, w1, g1, s1) = [(r, torch.from_numpy(w), torch.from_numpy(g), sc) for r, w, g, sc in res]
    assert torch.equal(w0, w1)                                           # broadcast from rank 0
    expect = torch.arange(1000, dtype=torch.float32) * 3.0               # (1 + 2) * arange
    assert torch.equal(g0, expect) and torch.equal(g1, expect)
    assert s0 == s1 == 0.5                                               # mean = sum * 1/world in Adam


def test_allreduce_is_identity_without_process_group():
    from dpig_amd.trainer import GradAllReduce
    ar = GradAllReduce()
    g = torch.ones(10)
    assert ar(g) == 1.0 and not ar.enabled and float(g.sum()) == 10.0
