"""Micro-benchmark of the bf16-STORAGE conv kernels (dpig_conv2d_*_bf16) on the 3x3 layer shapes of the DeepFashion
256x256 (B=8) and Market 128x64 (B=16) stage-I graphs, next to the legacy 'bf16c' mode (fp32 tensors, bf16 pipe).
Random operands (not zeros: the clock depends on the data).  DPIG_BF16_DMA=flat|buffer selects the LDS-DMA form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dpig_amd.hip_ops as H
dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, it=8):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e-3


LAYERS = [  # (tag, N, H, W, C, K, k, stride)
    ("df.E.res      ", 8, 256, 256, 128, 128, 3, 1),
    ("df.dec4       ", 8, 256, 256, 256, 256, 3, 1),
    ("df.dec3       ", 8, 128, 128, 512, 512, 3, 1),
    ("df.dec2       ", 8, 64, 64, 768, 768, 3, 1),
    ("df.dec1       ", 8, 32, 32, 1024, 1024, 3, 1),
    ("df.dec0       ", 8, 16, 16, 768, 768, 3, 1),
    ("df.roi.b0     ", 56, 64, 64, 128, 128, 3, 1),
    ("df.roi.b1     ", 56, 32, 32, 256, 256, 3, 1),
    ("df.roi.down0  ", 56, 64, 64, 128, 256, 3, 2),
    ("mk.dec4       ", 16, 128, 64, 256, 256, 3, 1),
    ("mk.dec3       ", 16, 64, 32, 512, 512, 3, 1),
    ("mk.roi.b0     ", 112, 48, 48, 128, 128, 3, 1),
    ("mk.roi.b2     ", 112, 12, 12, 384, 384, 3, 1),
    ("st2.roi.b0 B64", 448, 48, 48, 128, 128, 3, 1),
    ("df.D.2        ", 16, 128, 128, 64, 128, 5, 2),
    ("df.up1x1      ", 8, 128, 128, 512, 256, 1, 1),
]
quick = "--quick" in sys.argv
print("DMA form: %s" % os.environ.get("DPIG_BF16_DMA", "buffer"))
for (tag, N, Hh, W, C, K, k, s) in (LAYERS[:4] + LAYERS[9:11] if quick else LAYERS):
    up = tag.startswith("df.up")
    x = torch.randn(N, Hh, W, C, device=dev); w = torch.randn(k, k, C, K, device=dev) * 0.05; b = torch.randn(K, device=dev)
    xb = x.to(BF)
    H.set_compute("f32")
    y = H.conv2d_fwd(xb, w, b, stride=s, act=1, upsample2x=up)
    dy = torch.randn(y.shape, device=dev).to(BF)
    fl = 2.0 * y.numel() // (4 if up else 1) * k * k * C
    sh = H.filter_shadows(w)
    w._dpig_shadow = sh                      # persistent shadows, as in a trainer
    tf = timeit(lambda: H.conv2d_fwd(xb, w, b, stride=s, act=1, upsample2x=up))
    td = timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C), stride=s, upsample2x=up))
    dw = torch.empty((k, k, C, K), device=dev)
    tw = timeit(lambda: H.conv2d_wgrad(xb, dy, (k, k, C, K), stride=s, upsample2x=up, out=dw))
    line = "bf16  %s fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF" % (
        tag, tf * 1e6, fl / tf / 1e12, td * 1e6, fl / td / 1e12, tw * 1e6, fl / tw / 1e12)
    if "--legacy" in sys.argv:
        del w._dpig_shadow
        H.set_compute("bf16c")
        dy32 = dy.float()
        tf = timeit(lambda: H.conv2d_fwd(x, w, b, stride=s, act=1, upsample2x=up)); td = timeit(lambda: H.conv2d_dgrad(dy32, w, (N, Hh, W, C), stride=s, upsample2x=up))
        tw = timeit(lambda: H.conv2d_wgrad(x, dy32, (k, k, C, K), stride=s, upsample2x=up))
        line += "  || bf16c fwd %6.1f dgrad %6.1f wgrad %6.1f TF" % (fl / tf / 1e12, fl / td / 1e12, fl / tw / 1e12)
        H.set_compute("f32")
    if "--x3" in sys.argv:                   # fp32 tensors: exact fp32 pipe next to the split-bf16 pipe (effective TF/s)
        if hasattr(w, "_dpig_shadow"): del w._dpig_shadow
        dy32 = dy.float()
        for mode in ("f32", "bf16x3", "bf16x3+shadows", "bf16x3+shadows+planes"):
            H.set_compute(mode.split("+")[0])
            shx = H.FilterShadows([w], split=True) if "+" in mode else None      # persistent two-term shadows, as in a trainer
            H.X3_PLANES[0] = mode.endswith("planes")      # + the activation's split32 image (made once, outside the timed calls)
            tf = timeit(lambda: H.conv2d_fwd(x, w, b, stride=s, act=1, upsample2x=up)); td = timeit(lambda: H.conv2d_dgrad(dy32, w, (N, Hh, W, C), stride=s, upsample2x=up))
            tw = timeit(lambda: H.conv2d_wgrad(x, dy32, (k, k, C, K), stride=s, upsample2x=up, out=dw))
            line += "  || %s fwd %6.1f dgrad %6.1f wgrad %6.1f TF" % (mode, fl / tf / 1e12, fl / td / 1e12, fl / tw / 1e12)
            if shx is not None: shx.detach()
        H.X3_PLANES[0] = False
        H.set_compute("f32")
    print(line, flush=True)
    del x, xb, y, dy, w, dw
    torch.cuda.empty_cache()
