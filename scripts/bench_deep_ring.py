"""A/B of the four-stage k-loop (bg8d_kernel, round 6) on the layers whose plans give the chip one round of <= 256 workgroups: bf16 storage,
forward + dgrad, time and a checksum of the outputs (the two kernels must give the same bits).  Run with DPIG_BF16_DEEP=0 and =1."""
import hashlib, os, sys
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
# (N, H, W, C, K, k, stride)
SHAPES = ((16, 32, 16, 384, 384, 3, 1), (16, 16, 8, 512, 512, 3, 1), (16, 8, 4, 640, 640, 3, 1), (16, 8, 4, 768, 768, 3, 1), (112, 3, 3, 640, 640, 3, 1),
          (16, 64, 32, 256, 384, 3, 2), (16, 32, 16, 384, 512, 3, 2), (16, 16, 8, 512, 640, 3, 2), (16, 64, 32, 64, 128, 5, 2), (16, 32, 16, 128, 256, 5, 2),
          (16, 16, 8, 256, 512, 5, 2), (112, 6, 6, 512, 640, 3, 2), (8, 16, 16, 768, 768, 3, 1), (8, 16, 16, 640, 640, 3, 1), (56, 4, 4, 640, 640, 3, 1),
          (56, 2, 2, 768, 768, 3, 1), (16, 16, 8, 1024, 1024, 3, 1))
print("DPIG_BF16_DEEP=%s" % os.environ.get("DPIG_BF16_DEEP", "(default 1)"))
g = torch.Generator(device="cpu").manual_seed(1)
tot = 0.0
for (N, Hh, W, C, K, R, st) in SHAPES:
    x = (torch.randn(N, Hh, W, C, generator=g)).to(dev).bfloat16(); w = (torch.randn(R, R, C, K, generator=g) * 0.05).to(dev); b = torch.randn(K, generator=g).to(dev)
    y = H.conv2d_fwd(x, w, b, stride=st, act=1); dy = torch.randn(y.shape, generator=g).to(dev).bfloat16(); fl = 2.0 * y.numel() * R * R * C
    dx = H.conv2d_dgrad(dy, w, (N, Hh, W, C), stride=st)
    torch.cuda.synchronize()
    h = hashlib.sha1(y.view(torch.int16).cpu().numpy().tobytes() + dx.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:10]
    tf = timeit(lambda: H.conv2d_fwd(x, w, b, stride=st, act=1)); td = timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C), stride=st))
    tot += tf + td
    print("N%-3d %3dx%-3d C%-4d K%-4d k%d s%d fwd %6.1f us %6.1f TF | dgrad %6.1f us %6.1f TF | bits %s" % (N, Hh, W, C, K, R, st, tf * 1e6, fl / tf / 1e12, td * 1e6, fl / td / 1e12, h), flush=True)
print("sum %.1f us" % (tot * 1e6))
