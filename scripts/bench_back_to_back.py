import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpig_amd import hip_ops as H
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
H.set_compute("f32w"); H.set_wino_mode(2)
for (N, Hh, W, C) in [(16, 128, 64, 128), (16, 64, 32, 512)]:
    x = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    w = (torch.rand((3, 3, C, C), device=dev, generator=g) * 2 - 1) * 0.02
    b = torch.rand((C,), device=dev, generator=g)
    w._dpig_wino = H.wino_images(w)
    for _ in range(3): H.conv2d_fwd(x, w, b, act=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): H.conv2d_fwd(x, w, b, act=1)
    e1.record(); torch.cuda.synchronize()
    print("C%d: %.1f us per launch by events (40 back to back)" % (C, e0.elapsed_time(e1) / 40 * 1e3))
    gr = torch.cuda.CUDAGraph()
    y = torch.empty((N, Hh, W, C), device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        H.conv2d_fwd(x, w, b, act=1)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(40): H.conv2d_fwd(x, w, b, act=1)
    gr.replay(); torch.cuda.synchronize()
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print("C%d: %.1f us per launch inside one hipGraph of 40" % (C, e0.elapsed_time(e1) / 40 * 1e3))
