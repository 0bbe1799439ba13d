"""A/B of the halo-patch kernel's channel-chunk split (round 6): 3x3 stride-1 layers with fewer 128-pixel patches than resident slots,
bf16 storage, forward + dgrad.  Run twice: DPIG_BF16_HALO_SPLIT=0 (tap-major kernel + split-K, rounds 2-5) and =1 (default)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dpig_amd.hip_ops as H
dev = torch.device("cuda:0")
H.set_compute("bf16")
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e-3
SHAPES = ((16, 32, 16, 384, 384), (16, 16, 8, 512, 512), (16, 32, 16, 768, 768), (16, 16, 8, 1024, 1024), (8, 32, 32, 512, 512), (8, 16, 16, 768, 768),
          (8, 16, 16, 640, 640), (56, 16, 16, 384, 384), (8, 32, 32, 1024, 1024), (16, 64, 32, 256, 256))
print("DPIG_BF16_HALO_SPLIT=%s" % os.environ.get("DPIG_BF16_HALO_SPLIT", "(default 1)"))
for (N, Hh, W, C, K) in SHAPES:
    x = torch.randn(N, Hh, W, C, device=dev).bfloat16(); w = torch.randn(3, 3, C, K, device=dev) * 0.05; b = torch.randn(K, device=dev)
    y = H.conv2d_fwd(x, w, b, act=1); dy = torch.randn_like(y); fl = 2.0 * y.numel() * 9 * C
    tf = timeit(lambda: H.conv2d_fwd(x, w, b, act=1)); td = timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C)))
    print("N%-3d %3dx%-3d C%-4d fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF" % (N, Hh, W, C, tf * 1e6, fl / tf / 1e12, td * 1e6, fl / td / 1e12), flush=True)
