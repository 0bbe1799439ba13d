"""Per-layer A/B of the four- and eight-wave forms of the 128 x 128 bf16 kernel (bg_kernel / bg8_kernel) on the layers of the Market,
stage-II and DeepFashion graphs the 128-tile family serves: forward and dgrad, each replayed from a hipGraph of 20 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dpig_amd.hip_ops as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
LAYERS = [  # (N, H, W, C, K, k, stride)
    (16, 32, 16, 384, 384, 3, 1), (16, 16, 8, 512, 512, 3, 1), (16, 8, 4, 640, 640, 3, 1), (16, 64, 32, 256, 256, 3, 1),
    (112, 12, 12, 384, 384, 3, 1), (112, 6, 6, 512, 512, 3, 1), (112, 3, 3, 640, 640, 3, 1), (112, 24, 24, 256, 256, 3, 1),
    (16, 16, 8, 1024, 1024, 3, 1), (16, 8, 4, 768, 768, 3, 1), (16, 32, 16, 768, 768, 3, 1), (16, 64, 32, 512, 512, 3, 1),
    (16, 64, 32, 256, 384, 3, 2), (16, 32, 16, 384, 512, 3, 2), (16, 16, 8, 512, 640, 3, 2), (112, 12, 12, 384, 512, 3, 2),
    (16, 64, 32, 64, 128, 5, 2), (16, 32, 16, 128, 256, 5, 2), (16, 16, 8, 256, 512, 5, 2),
    (448, 12, 12, 384, 384, 3, 1), (448, 6, 6, 512, 512, 3, 1), (448, 3, 3, 640, 640, 3, 1), (64, 16, 8, 512, 512, 3, 1),
    (8, 32, 32, 512, 512, 3, 1), (8, 16, 16, 640, 640, 3, 1), (8, 16, 16, 768, 768, 3, 1), (56, 8, 8, 512, 512, 3, 1), (56, 4, 4, 640, 640, 3, 1),
    (56, 16, 16, 384, 384, 3, 1), (8, 32, 32, 1024, 1024, 3, 1), (8, 64, 64, 384, 384, 3, 1),
]
def graph_time(fn, n=20, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3
tot = [[0.0, 0.0], [0.0, 0.0]]
for (N, Hh, W, C, K, k, s) in LAYERS:
    x = torch.randn(N, Hh, W, C, device=dev).to(BF); w = torch.randn(k, k, C, K, device=dev) * 0.05; b = torch.randn(K, device=dev)
    w._dpig_shadow = H.filter_shadows(w)
    y = H.conv2d_fwd(x, w, b, stride=s, act=1); dy = torch.randn(y.shape, device=dev).to(BF)
    out_y = torch.empty_like(y); out_dx = torch.empty_like(x)
    t = {}
    for mode in (0, 1):
        H.set_wave8(mode)
        t[mode] = (graph_time(lambda: H.conv2d_fwd(x, w, b, stride=s, act=1, out=out_y)),
                   graph_time(lambda: H.conv2d_dgrad(dy, w, (N, Hh, W, C), stride=s, mask=x, act=1, out=out_dx)))
    H.set_wave8(3)
    M = N * (-(-Hh // s)) * (-(-W // s))
    for i in (0, 1):
        tot[i][0] += t[0][i]; tot[i][1] += t[1][i]
    print("N%-3d %3dx%-3d C%-4d K%-4d k%d s%d  M %7d | fwd %6.1f -> %6.1f us (%+5.1f %%) | dgrad %6.1f -> %6.1f us (%+5.1f %%)" % (
        N, Hh, W, C, K, k, s, M, t[0][0], t[1][0], (t[1][0] / t[0][0] - 1) * 100, t[0][1], t[1][1], (t[1][1] / t[0][1] - 1) * 100))
print("sum fwd %.1f -> %.1f us, dgrad %.1f -> %.1f us" % (tot[0][0], tot[0][1], tot[1][0], tot[1][1]))
