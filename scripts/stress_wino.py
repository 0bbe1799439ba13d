"""Winograd vs direct kernels over EVERY element of full-size layers, every epilogue the models use, repeated (race hunting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpig_amd import hip_ops as H
dev = torch.device("cuda:0")
LAYERS = [("E.res 128x64 C128", 16, 128, 64, 128), ("roi b0 48x48 C128", 112, 48, 48, 128), ("enc1 64x32 C256", 16, 64, 32, 256),
          ("roi b1 24x24 C256", 112, 24, 24, 256), ("enc2 32x16 C384", 16, 32, 16, 384), ("roi b2 12x12 C384", 112, 12, 12, 384),
          ("dec2 32x16 C768", 16, 32, 16, 768), ("dec3 64x32 C512", 16, 64, 32, 512), ("dec4 128x64 C256", 16, 128, 64, 256)]
g = torch.Generator(device=dev).manual_seed(0)
bad = 0
for name, N, Hh, W, C in LAYERS:
    x = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    w = (torch.rand((3, 3, C, C), device=dev, generator=g) * 2 - 1) * (1.5 / (9 * C) ** 0.5)
    b = torch.rand((C,), device=dev, generator=g) - 0.5
    res = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    dy = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    msk = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    big = torch.zeros((N, Hh, W, 2 * C), device=dev)

    def run():
        outs = {}
        outs["fwd relu"] = H.conv2d_fwd(x, w, b, act=1)
        o, oa = torch.empty_like(x), torch.empty_like(x)
        H.conv2d_fwd(x, w, b, act=1, residual=res, res_after_act=True, out=o, out_act=oa)
        outs["tail out"], outs["tail act"] = o, oa
        bb = big.clone()
        H.conv2d_fwd(x, w, b, act=1, out=bb[..., C:])
        outs["slice"] = bb
        outs["dgrad"] = H.conv2d_dgrad(dy, w, (N, Hh, W, C))
        outs["dgrad mask"] = H.conv2d_dgrad(dy, w, (N, Hh, W, C), mask=msk, act=1)
        outs["dgrad acc mask"] = H.conv2d_dgrad(dy, w, (N, Hh, W, C), accum=res, mask=msk, act=1)
        outs["dgrad acc"] = H.conv2d_dgrad(dy, w, (N, Hh, W, C), accum=res)
        return outs
    H.set_compute("f32")
    ref = run()
    H.set_compute("f32w"); H.set_wino_mode(2)
    w._dpig_wino = H.wino_images(w)
    runs = [run() for _ in range(3)]
    H.set_wino_mode(1); H.set_compute("f32")
    line = []
    for k in ref:
        scale = float(ref[k].abs().max())
        err = max(float((r[k] - ref[k]).abs().max()) for r in runs) / scale
        nbad = int(((runs[0][k] - ref[k]).abs() > 2e-5 * scale).sum())
        same = all(torch.equal(runs[0][k], r[k]) for r in runs[1:])
        if err > 2e-5 or not same:
            bad += 1
        line.append("%s %.1e%s%s" % (k, err, "" if same else " NONDET", " bad=%d" % nbad if nbad else ""))
    print("%-20s %s" % (name, " | ".join(line)), flush=True)
print("FAILURES:", bad)
