"""GPU check + timing of bhq32_kernel (csrc/dpig_conv_bf16_q.hip), the halo-staged 512 x 128 kernel of the 128-column 3x3 layers:

    timeout 120 python scripts/check_bhq32.py        # (own timeout: a hang must not cost a strike)

Forward (+ bias + ReLU, + residual) and the stride-1 dgrad (* mask) of 128-column 3x3 layers with the 512 x 128 variant forced -- which the
switch routes to bhq32_kernel where the layer is eligible -- against the 128 x 128 kernels (large tiles off) on the same operands: the k
orders differ ((32-chunk, tap) vs (64-chunk, tap) / (tap, chunk)), so the bar is one bf16 ulp of the stored output; then timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dpig_amd.hip_ops as H
dev = torch.device("cuda:0"); BF = torch.bfloat16


def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e-3


def close(a, b, what):
    a, b = a.float(), b.float()
    tol = 2.0 ** -7 * b.abs().clamp_min(2.0 ** -6)            # one bf16 ulp of the value (values below 2^-6: absolute)
    bad = ((a - b).abs() > tol).float().mean().item()
    print("  %-28s max |diff| %.3g, outside one ulp: %.4f %%" % (what, (a - b).abs().max().item(), 100 * bad))
    return bad < 1e-3


ok = True
for (N, Hh, W, C, K) in [(2, 32, 16, 128, 128), (2, 64, 48, 64, 128), (1, 96, 32, 192, 104), (8, 256, 256, 128, 128)]:
    g = torch.Generator(device=dev).manual_seed(N * 7 + C)
    x = torch.randn(N, Hh, W, C, device=dev, generator=g).to(BF)
    w = torch.randn(3, 3, C, K, device=dev, generator=g) * (1.0 / (9 * C) ** 0.5)
    b = torch.randn(K, device=dev, generator=g)
    r = torch.randn(N, Hh, W, K, device=dev, generator=g).to(BF)
    dy = torch.randn(N, Hh, W, K, device=dev, generator=g).to(BF)
    m = torch.randn(N, Hh, W, C, device=dev, generator=g).to(BF)
    w._dpig_shadow = H.filter_shadows(w)
    print("layer N%d %dx%d C%d K%d" % (N, Hh, W, C, K))
    res = {}
    for name, mode, var in (("128-tile", 0, 0), ("bhq32 (variant 2 forced)", 2, 2)):
        H.set_large_tile(mode, var)
        res[name] = (H.conv2d_fwd(x, w, b, act=1), H.conv2d_fwd(x, w, b, act=1, residual=r),
                     H.conv2d_dgrad(dy, w, (N, Hh, W, C), mask=m, act=1))
    a, c = res["128-tile"], res["bhq32 (variant 2 forced)"]
    ok &= close(c[0], a[0], "fwd + bias + relu") & close(c[1], a[1], "fwd + residual + relu") & close(c[2], a[2], "dgrad * relu'(mask)")
    fl = 2.0 * N * Hh * W * K * 9 * C
    for name, mode, var in (("128-tile", 0, 0), ("q512 / bhq32", 2, 2)):
        H.set_large_tile(mode, var)
        print("  %-14s fwd %7.1f TF   dgrad %7.1f TF" % (name, fl / timeit(lambda: H.conv2d_fwd(x, w, b, act=1)) / 1e12,
                                                          fl / timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C), mask=m, act=1)) / 1e12))
H.set_large_tile(1, 0)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
