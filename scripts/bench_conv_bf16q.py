"""A/B of the bf16-storage forward / stride-1 dgrad tile families on the 3x3 layers of the DeepFashion 256x256 (B=8) and
Market 128x64 (B=16) graphs: 128 x 128 kernels (bh / bg) vs the 8-wave large-tile kernels (bq<2,4> = 256 x 256,
bq<4,2> = 512 x 128).  Random operands; interleaved rounds, median of the rounds (guide rule 24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dpig_amd.hip_ops as H
dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, it=6):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e-3


LAYERS = [  # (tag, N, H, W, C, K)
    ("df.E.res  256^2 C128      ", 8, 256, 256, 128, 128),
    ("df.dec4   256^2 C256      ", 8, 256, 256, 256, 256),
    ("df.enc1   128^2 C256      ", 8, 128, 128, 256, 256),
    ("df.dec3   128^2 C512      ", 8, 128, 128, 512, 512),
    ("df.enc2   64^2  C384      ", 8, 64, 64, 384, 384),
    ("df.dec2   64^2  C768      ", 8, 64, 64, 768, 768),
    ("df.dec1   32^2  C1024     ", 8, 32, 32, 1024, 1024),
    ("df.roi.b0 N56 64^2 C128   ", 56, 64, 64, 128, 128),
    ("df.roi.b1 N56 32^2 C256   ", 56, 32, 32, 256, 256),
    ("df.roi.b2 N56 16^2 C384   ", 56, 16, 16, 384, 384),
    ("mk.E.res  128x64 C128     ", 16, 128, 64, 128, 128),
    ("mk.dec4   128x64 C256     ", 16, 128, 64, 256, 256),
    ("mk.dec3   64x32 C512      ", 16, 64, 32, 512, 512),
    ("mk.dec2   32x16 C768      ", 16, 32, 16, 768, 768),
    ("mk.roi.b0 N112 48^2 C128  ", 112, 48, 48, 128, 128),
    ("mk.roi.b1 N112 24^2 C256  ", 112, 24, 24, 256, 256),
    ("st2.roi.b0 N448 48^2 C128 ", 448, 48, 48, 128, 128),
]
quick = "--quick" in sys.argv
rounds = 3
MODES = [("128", 0, 0), ("q256", 2, 1), ("q512", 2, 2)]
for (tag, N, Hh, W, C, K) in (LAYERS[:4] + LAYERS[11:13] if quick else LAYERS):
    x = torch.randn(N, Hh, W, C, device=dev).to(BF); w = torch.randn(3, 3, C, K, device=dev) * 0.05; b = torch.randn(K, device=dev)
    dy = torch.randn(N, Hh, W, K, device=dev).to(BF)
    m = torch.randn(N, Hh, W, C, device=dev).to(BF)
    fl = 2.0 * N * Hh * W * K * 9 * C
    w._dpig_shadow = H.filter_shadows(w)
    res = {name: ([], [], []) for name, _, _ in MODES}
    dw = torch.empty((3, 3, C, K), device=dev)
    for r in range(rounds):
        for name, mode, var in MODES:
            H.set_large_tile(mode, var)
            res[name][0].append(timeit(lambda: H.conv2d_fwd(x, w, b, act=1)))
            res[name][1].append(timeit(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C), mask=m, act=1)))
            H.set_large_tile_wgrad(mode, var)
            res[name][2].append(timeit(lambda: H.conv2d_wgrad(x, dy, (3, 3, C, K), out=dw)))
    H.set_large_tile(1, 0)
    H.set_large_tile_wgrad(1, 0)
    line = tag
    for name, _, _ in MODES:
        tf = sorted(res[name][0])[rounds // 2]; td = sorted(res[name][1])[rounds // 2]; tw = sorted(res[name][2])[rounds // 2]
        line += " | %s fwd %6.1f dgrad %6.1f wgrad %6.1f" % (name, fl / tf / 1e12, fl / td / 1e12, fl / tw / 1e12)
    print(line, flush=True)
    del x, dy, m, w
    torch.cuda.empty_cache()
