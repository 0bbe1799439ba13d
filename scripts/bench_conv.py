"""Micro-benchmark of the conv kernels on Market stage-I layer shapes (B=16). Prints TFLOP/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dpig_amd.hip_ops as H

dev = torch.device("cuda:0")
# name, N, H, W, C, K, k, s
LAYERS = [
    ("dec4   128x64 256->256", 16, 128, 64, 256, 256, 3, 1),
    ("dec3   64x32 512->512 ", 16, 64, 32, 512, 512, 3, 1),
    ("dec2   32x16 768->768 ", 16, 32, 16, 768, 768, 3, 1),
    ("dec1   16x8 1024->1024", 16, 16, 8, 1024, 1024, 3, 1),
    ("dec0   8x4 768->768   ", 16, 8, 4, 768, 768, 3, 1),
    ("enc0   128x64 128->128", 16, 128, 64, 128, 128, 3, 1),
    ("roi b0 48x48 128->128 ", 112, 48, 48, 128, 128, 3, 1),
    ("roi dn 48x48 128->256 ", 112, 48, 48, 128, 256, 3, 2),
    ("roi b4 3x3 640->640   ", 112, 3, 3, 640, 640, 3, 1),
    ("D.1    128x64 3->64   ", 16, 128, 64, 3, 64, 5, 2),
    ("D.2    64x32 64->128  ", 16, 64, 32, 64, 128, 5, 2),
    ("G.out  128x64 256->3  ", 16, 128, 64, 256, 3, 3, 1),
    ("E.stem 128x64 3->128  ", 16, 128, 64, 3, 128, 3, 1),
]

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

for name, N, Hh, W, C, K, k, s in LAYERS:
    x = torch.randn(N, Hh, W, C, device=dev)
    w = torch.randn(k, k, C, K, device=dev) * 0.05
    b = torch.randn(K, device=dev)
    if os.environ.get("ZERO"):      # DVFS probe: all-zero operands draw less power -> higher sustained clock
        x.zero_(); w.zero_(); b.zero_()
    y = H.conv2d_fwd(x, w, b, stride=s, act=1)
    dy = torch.randn_like(y)
    if os.environ.get("ZERO"): dy.zero_()
    flops = 2.0 * y.numel() * k * k * C
    tf = timeit(lambda: H.conv2d_fwd(x, w, b, stride=s, act=1))
    td = timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C), stride=s))
    tw = timeit(lambda: H.conv2d_wgrad(x, dy, (k, k, C, K), stride=s))
    print("%s  fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF" % (
        name, tf * 1e6, flops / tf / 1e12, td * 1e6, flops / td / 1e12, tw * 1e6, flops / tw / 1e12), flush=True)
