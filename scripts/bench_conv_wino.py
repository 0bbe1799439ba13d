"""Direct fp32 MFMA conv vs the Winograd F(2x2,3x3) kernel, layer by layer (HIP events, stand-alone launches).
   python scripts/bench_conv_wino.py [market|df256]      effective TFLOP/s = DIRECT-conv FLOPs / time for both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpig_amd import hip_ops as H
dev = torch.device("cuda:0")
H.set_wino4_mode(0)              # this script compares the direct kernels with F(2x2,3x3) / F(3x3,2x2); scripts/bench_conv_wino4.py holds F(4x4,3x3)
MARKET = [("E.res / enc0 128x64 C128", 16, 128, 64, 128), ("roi b0 48x48 C128", 112, 48, 48, 128), ("enc1 64x32 C256", 16, 64, 32, 256),
          ("roi b1 24x24 C256", 112, 24, 24, 256), ("enc2 32x16 C384", 16, 32, 16, 384), ("roi b2 12x12 C384", 112, 12, 12, 384),
          ("enc3 16x8 C512", 16, 16, 8, 512), ("roi b3 6x6 C512", 112, 6, 6, 512), ("enc4 8x4 C640", 16, 8, 4, 640),
          ("dec0 8x4 C768", 16, 8, 4, 768), ("dec1 16x8 C1024", 16, 16, 8, 1024), ("dec2 32x16 C768", 16, 32, 16, 768),
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


print("%-28s %10s | %9s %9s %6s | %9s %9s %6s | %9s %9s %6s | max diff" % ("layer", "GF", "fwd dir", "fwd wino", "x", "dg dir", "dg wino", "x", "wg dir", "wg wino", "x"))
g = torch.Generator(device=dev).manual_seed(0)
tot = [0.0] * 6
for name, N, Hh, W, C in layers:
    x = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    w = (torch.rand((3, 3, C, C), device=dev, generator=g) * 2 - 1) * (1.5 / (9 * C) ** 0.5)
    b = torch.rand((C,), device=dev, generator=g) - 0.5
    dy = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    flops = 2.0 * N * Hh * W * 9 * C * C
    H.set_compute("f32")
    yd = H.conv2d_fwd(x, w, b, act=1)
    t_fd = timeit(lambda: H.conv2d_fwd(x, w, b, act=1))
    t_dd = timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C)))
    dwb, dbb = torch.empty_like(w), torch.empty_like(b)
    t_wd = timeit(lambda: H.conv2d_wgrad(x, dy, (3, 3, C, C), out=dwb, db=dbb))
    dwd = dwb.clone()
    H.set_compute("f32w")
    H.set_wino_mode(2)
    im = H.wino_images(w)
    w._dpig_wino = im                     # persistent images, as a trainer's parameters carry them
    yw = H.conv2d_fwd(x, w, b, act=1)
    t_fw = timeit(lambda: H.conv2d_fwd(x, w, b, act=1))
    t_dw = timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C)))
    t_ww = timeit(lambda: H.conv2d_wgrad(x, dy, (3, 3, C, C), out=dwb, db=dbb))
    wdiff = float((dwb - dwd).abs().max() / dwd.abs().max())
    H.set_wino_mode(1)
    wpays = bool(H.lib().dpig_conv2d_wgrad_wino_eligible(__import__("ctypes").byref(H._desc(N, Hh, W, C, C, 3, 3, 1, C, C))))
    pays = bool(H.lib().dpig_conv2d_wino_eligible(__import__("ctypes").byref(H._desc(N, Hh, W, C, C, 3, 3, 1, C, C)), 0))
    H.set_compute("f32")
    diff = float((yw - yd).abs().max() / yd.abs().max())
    for i, t in enumerate((t_fd, t_fw, t_dd, t_dw, t_wd, t_ww)):
        tot[i] += t
    print("%-28s %10.1f | %7.1f TF %7.1f TF %5.2fx | %7.1f TF %7.1f TF %5.2fx | %7.1f TF %7.1f TF %5.2fx | %.1e %.1e  model:%s/%s" % (
        name, flops / 1e9, flops / t_fd / 1e12, flops / t_fw / 1e12, t_fd / t_fw, flops / t_dd / 1e12, flops / t_dw / 1e12, t_dd / t_dw,
        flops / t_wd / 1e12, flops / t_ww / 1e12, t_wd / t_ww, diff, wdiff, "wino" if pays else "direct", "wino" if wpays else "direct"))
print("sum of launches [ms]: fwd direct %.3f wino %.3f | dgrad direct %.3f wino %.3f | wgrad (+ bias gradient) direct %.3f wino %.3f" % tuple(t * 1e3 for t in tot))
