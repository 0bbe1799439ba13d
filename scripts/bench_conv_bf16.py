"""Micro-benchmark: fp32 vs bf16 matrix-pipe mode of the conv kernels on decoder layer shapes (B=16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dpig_amd.hip_ops as H
dev = torch.device("cuda:0")
def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e-3
for (N, Hh, W, C, K) in ((16, 128, 64, 256, 256), (16, 64, 32, 512, 512), (16, 128, 64, 128, 128), (16, 32, 16, 768, 768), (16, 16, 8, 1024, 1024)):
    x = torch.randn(N, Hh, W, C, device=dev); w = torch.randn(3, 3, C, K, device=dev) * 0.05; b = torch.randn(K, device=dev)
    y = H.conv2d_fwd(x, w, b, act=1); dy = torch.randn_like(y); fl = 2.0 * y.numel() * 9 * C
    for mode in ("f32", "bf16"):
        H.set_compute(mode)
        tf = timeit(lambda: H.conv2d_fwd(x, w, b, act=1)); td = timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C)))
        tw = timeit(lambda: H.conv2d_wgrad(x, dy, (3, 3, C, K)))
        print("%-4s %s fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF" % (
            mode, (N, Hh, W, C, K), tf * 1e6, fl / tf / 1e12, td * 1e6, fl / td / 1e12, tw * 1e6, fl / tw / 1e12), flush=True)
    H.set_compute("f32")
