"""A/B of the class-balanced split plan of the stride-2 dgrad (round 6; DPIG_S2_BALANCE=0 / 1): the step's eleven stride-2 layers, fp32 and bf16
storage, dgrad only; time per call and a checksum (a different split plan changes the summation order of the fp32 partials: results agree to
rounding, not bit for bit -- the printed max difference against the unsplit-order reference is what the kernel tests bound)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dpig_amd.hip_ops as H
dev = torch.device("cuda:0")
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e-3
SHAPES = ((16, 128, 64, 128, 256, 3), (16, 64, 32, 256, 384, 3), (16, 32, 16, 384, 512, 3), (16, 16, 8, 512, 640, 3), (112, 48, 48, 128, 256, 3),
          (112, 24, 24, 256, 384, 3), (112, 12, 12, 384, 512, 3), (112, 6, 6, 512, 640, 3), (16, 64, 32, 64, 128, 5), (16, 32, 16, 128, 256, 5), (16, 16, 8, 256, 512, 5))
print("DPIG_S2_BALANCE=%s" % os.environ.get("DPIG_S2_BALANCE", "(default 1)"))
g = torch.Generator(device="cpu").manual_seed(1)
for mode in ("f32", "bf16"):
    H.set_compute(mode)
    tot = 0.0
    for (N, Hh, W, C, K, R) in SHAPES:
        Ho, Wo = Hh // 2, W // 2
        w = (torch.randn(R, R, C, K, generator=g) * 0.05).to(dev)
        dy = torch.randn(N, Ho, Wo, K, generator=g).to(dev)
        if mode == "bf16": dy = dy.bfloat16()
        fl = 2.0 * N * Ho * Wo * K * R * R * C
        dx = H.conv2d_dgrad(dy, w, (N, Hh, W, C), stride=2)
        t = timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C), stride=2)); tot += t
        print("%-4s N%-3d %3dx%-3d C%-4d<-K%-4d k%d s2 dgrad %7.1f us %6.1f TF  sum %.6e" % (mode, N, Hh, W, C, K, R, t * 1e6, fl / t / 1e12, float(dx.float().double().sum())), flush=True)
    print("%s total %.1f us" % (mode, tot * 1e6))
H.set_compute("f32")
