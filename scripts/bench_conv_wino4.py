"""Winograd F(2x2,3x3) vs F(4x4,3x3), layer by layer (HIP events, stand-alone launches): forward and dgrad.
   python scripts/bench_conv_wino4.py [market|df256]      effective TFLOP/s = DIRECT-conv FLOPs / time; executed = 36/144 of it."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpig_amd import hip_ops as H
dev = torch.device("cuda:0")
MARKET = [("E.res / enc0 128x64 C128", 16, 128, 64, 128), ("roi b0 48x48 C128", 112, 48, 48, 128), ("enc1 64x32 C256", 16, 64, 32, 256),
          ("roi b1 24x24 C256", 112, 24, 24, 256), ("enc2 32x16 C384", 16, 32, 16, 384), ("roi b2 12x12 C384", 112, 12, 12, 384),
          ("enc3 16x8 C512", 16, 16, 8, 512), ("dec1 16x8 C1024", 16, 16, 8, 1024), ("dec2 32x16 C768", 16, 32, 16, 768),
          ("dec3 64x32 C512", 16, 64, 32, 512), ("dec4 128x64 C256", 16, 128, 64, 256)]
DF = [("E.res 256x256 C128", 8, 256, 256, 128), ("enc1 128x128 C256", 8, 128, 128, 256), ("enc2 64x64 C384", 8, 64, 64, 384),
      ("dec2 64x64 C768", 8, 64, 64, 768), ("dec3 128x128 C512", 8, 128, 128, 512), ("dec4 256x256 C256", 8, 256, 256, 256)]
layers = DF if sys.argv[1:] == ["df256"] else MARKET


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


print("%-28s %9s | %8s %8s %8s %6s | %8s %8s %6s | F4 exec TF | diff F2/F4 vs direct | model" % ("layer", "GF", "fwd dir", "fwd F2", "fwd F4", "x", "dg F2", "dg F4", "x"))
g = torch.Generator(device=dev).manual_seed(0)
tot = [0.0] * 4
H.set_wino_mode(2)
for name, N, Hh, W, C in layers:
    x = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    w = (torch.rand((3, 3, C, C), device=dev, generator=g) * 2 - 1) * (1.5 / (9 * C) ** 0.5)
    b = torch.rand((C,), device=dev, generator=g) - 0.5
    dy = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    flops = 2.0 * N * Hh * W * 9 * C * C
    d = H._desc(N, Hh, W, C, C, 3, 3, 1, C, C)
    H.set_compute("f32")
    yd = H.conv2d_fwd(x, w, b, act=1)
    t_fd = timeit(lambda: H.conv2d_fwd(x, w, b, act=1))
    H.set_compute("f32w")
    H.set_wino4_mode(0)
    w._dpig_wino = H.wino_images(w)
    y2 = H.conv2d_fwd(x, w, b, act=1)
    t_f2 = timeit(lambda: H.conv2d_fwd(x, w, b, act=1))
    t_d2 = timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C)))
    H.set_wino4_mode(1)
    pays = bool(H.lib().dpig_conv2d_wino4_eligible(ctypes.byref(d), 0))
    H.set_wino4_mode(2)
    if not H.lib().dpig_conv2d_wino4_eligible(ctypes.byref(d), 0):
        print("%-28s %9.1f | %6.1f TF %6.1f TF   (no F(4x4) block form)" % (name, flops / 1e9, flops / t_fd / 1e12, flops / t_f2 / 1e12))
        H.set_compute("f32")
        continue
    w._dpig_wino4 = H.wino4_images(w)
    y4 = H.conv2d_fwd(x, w, b, act=1)
    t_f4 = timeit(lambda: H.conv2d_fwd(x, w, b, act=1))
    t_d4 = timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C)))
    t_d4m = timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C), mask=x, act=1))          # the form most dgrads of the step have: * act'(mask)
    t_f4r = timeit(lambda: H.conv2d_fwd(x, w, b, act=1, residual=dy))                    # forward with a residual (the encoder's blocks)
    H.set_compute("f32")
    for i, t in enumerate((t_f2, t_f4, t_d2, t_d4)):
        tot[i] += t
    print("%-28s %9.1f | %6.1f TF %6.1f TF %6.1f TF %5.2fx | %6.1f TF %6.1f TF %5.2fx | %6.1f %6.1f | %.1e %.1e | %s" % (
        name, flops / 1e9, flops / t_fd / 1e12, flops / t_f2 / 1e12, flops / t_f4 / 1e12, t_f2 / t_f4, flops / t_d2 / 1e12, flops / t_d4 / 1e12,
        t_d2 / t_d4, flops / 4 / t_f4 / 1e12, flops / 4 / t_d4 / 1e12, float((y2 - yd).abs().max() / yd.abs().max()),
        float((y4 - yd).abs().max() / yd.abs().max()), "F4" if pays else "F2") + " | dgrad*mask %.1f TF, fwd+res %.1f TF" % (flops / t_d4m / 1e12, flops / t_f4r / 1e12))
print("sum of launches on layers with the form [ms]: fwd F2 %.3f F4 %.3f | dgrad F2 %.3f F4 %.3f" % tuple(t * 1e3 for t in tot))
