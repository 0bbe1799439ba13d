import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dpig_amd import hip_ops as H
dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator(device=dev).manual_seed(0)
for (N, Hh, W, C, K) in [(16, 64, 32, 64, 128), (16, 32, 16, 128, 256), (16, 16, 8, 256, 512)]:
    x = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    w = (torch.rand((5, 5, C, K), device=dev, generator=g) * 2 - 1) * 0.02
    b = torch.zeros(K, device=dev); sc = torch.ones(K, device=dev); of = torch.zeros(K, device=dev)
    def fused():
        y, st = H.conv2d_fwd_stats(x, w, b, stride=2)
        return H.bn_fwd(y, sc, of, 1e-5, 2, 0.2, stats=st)
    def plain():
        y = H.conv2d_fwd(x, w, b, stride=2)
        return H.bn_fwd(y, sc, of, 1e-5, 2, 0.2)
    tc = timeit(lambda: H.conv2d_fwd(x, w, b, stride=2)); ts = timeit(lambda: H.conv2d_fwd_stats(x, w, b, stride=2))
    tf, tp = timeit(fused), timeit(plain)
    dy = torch.rand((N, Hh // 2, W // 2, K), device=dev, generator=g)
    td = timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C), stride=2)); tw = timeit(lambda: H.conv2d_wgrad(x, dy, (5, 5, C, K), stride=2))
    fl = 2.0 * N * (Hh // 2) * (W // 2) * 25 * C * K
    print("%dx%d C%d->%d: conv %.1f us (%.0f TF)  conv+stats %.1f us | conv+BN fused-stats %.1f us, plain %.1f us | dgrad %.1f us (%.0f TF) wgrad %.1f us (%.0f TF)" % (
        Hh, W, C, K, tc, fl / tc / 1e6, ts, tf, tp, td, fl / td / 1e6, tw, fl / tw / 1e6))
