"""Stand-alone timings of the thin (HBM / latency bound) kernels at the DeepFashion and Market sizes: crop_and_resize backward,
border-class sums, the critic's first-layer dgrad.  usage: python scripts/ubench_thin.py [df|market]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__
__graft_entry__.build()
from dpig_amd import hip_ops as H, synthetic
which = sys.argv[1] if len(sys.argv) > 1 else "df"
dev = torch.device("cuda:0")
B, Hh, W, C, crop, bf = (8, 256, 256, 128, 64, True) if which == "df" else (16, 128, 64, 128, 48, False)
batch = synthetic.make_batch(B, img_H=Hh, img_W=W, seed=100)
bbox = torch.as_tensor(batch["part_bbox"]).to(dev)
boxes, ind = H.roi_boxes(bbox, 7, Hh, W)
dt = torch.bfloat16 if bf else torch.float32
H.set_compute("bf16" if bf else "f32")
dout = torch.randn(7 * B, crop, crop, C, device=dev).to(dt)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(which, "crop_resize_bwd  %.1f us" % timeit(lambda: H.crop_resize_bwd(dout, boxes, ind, (B, Hh, W, C))))
img = torch.randn(B, Hh, W, C, device=dev).to(dt)
print(which, "crop_resize_fwd  %.1f us" % timeit(lambda: H.crop_resize_fwd(img, boxes, ind, crop, crop)))
print(which, "border_class_sum %.1f us" % timeit(lambda: H.border_class_sum(img)))
dy = torch.randn(B, Hh // 2, W // 2, 64, device=dev).to(dt)
w = torch.randn(5, 5, 3, 64, device=dev) * 0.02
print(which, "fewc dgrad 5x5s2 %.1f us" % timeit(lambda: H.conv2d_dgrad(dy, w, (B, Hh, W, 3), stride=2)))
x3 = torch.randn(B, Hh, W, 3, device=dev)
print(which, "fewc fwd 5x5s2   %.1f us" % timeit(lambda: H.conv2d_fwd(x3, w, None, stride=2, act=2)))
xw = torch.randn(B, Hh, W, 2 * 128, device=dev).to(dt)
w3 = torch.randn(3, 3, 256, 3, device=dev) * 0.02
print(which, "thin3 fwd        %.1f us" % timeit(lambda: H.conv2d_fwd(xw, w3, None)))
dy3 = torch.randn(B, Hh, W, 3, device=dev)
print(which, "thin3 dgrad      %.1f us" % timeit(lambda: H.conv2d_dgrad(dy3, w3, (B, Hh, W, 256))))
