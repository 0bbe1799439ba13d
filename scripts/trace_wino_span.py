"""Where a persistent Winograd launch loses time OUTSIDE its workgroups' items: per XCD (s_memtime is per XCD), from the
dpig_debug_wino_trace stamps of one launch: start ramp (first item's start after the XCD's earliest), gaps between a workgroup's items,
tail (the XCD's last end minus the workgroup's own last end).   python scripts/trace_wino_span.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpig_amd import hip_ops as H
dev = torch.device("cuda:0")
lib = H.lib()
lib.dpig_debug_wino_trace.argtypes = [ctypes.c_void_p]
lib.dpig_debug_wino_trace.restype = ctypes.c_int
g = torch.Generator(device=dev).manual_seed(0)
H.set_compute("f32w"); H.set_wino_mode(2)
G = 256
for name, N, Hh, W, C in [("C128 128x64", 16, 128, 64, 128), ("C256 128x64", 16, 128, 64, 256), ("C512 64x32", 16, 64, 32, 512)]:
    x = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    w = (torch.rand((3, 3, C, C), device=dev, generator=g) * 2 - 1) * 0.02
    b = torch.rand((C,), device=dev, generator=g)
    w._dpig_wino = H.wino_images(w)
    for _ in range(200):
        H.conv2d_fwd(x, w, b, act=1)
    items = (N * (Hh // 2) * (W // 2) + 63) // 64 * (C // 64)
    buf = torch.zeros(items * 8, dtype=torch.int64, device=dev)
    lib.dpig_debug_wino_trace(ctypes.c_void_p(buf.data_ptr()))
    H.conv2d_fwd(x, w, b, act=1)
    torch.cuda.synchronize()
    lib.dpig_debug_wino_trace(None)
    t = buf.view(items, 8).cpu().double()
    per = items // G
    T = 10.0                                        # ns per tick of s_memrealtime (100 MHz)
    s0 = t[:, 5].min().item()
    e1 = t[:, 6].max().item()
    first = [(t[wg, 5].item() - s0) * T / 1e3 for wg in range(G)]
    last = [(e1 - t[wg + (per - 1) * G, 6].item()) * T / 1e3 for wg in range(G)]
    busy = [sum((t[wg + i * G, 6] - t[wg + i * G, 5]).item() for i in range(per)) * T / 1e3 for wg in range(G)]
    gaps = [(t[wg + (i + 1) * G, 5] - t[wg + i * G, 6]).item() * T / 1e3 for wg in range(G) for i in range(per - 1)]
    m = lambda a: sum(a) / max(len(a), 1)
    busy_by_wg = list(busy)
    first.sort(); last.sort()
    print("%-12s items %5d (%d per workgroup) | first start -> last end %7.1f us | inside items, mean per workgroup %7.1f us | start after the "
          "first workgroup: mean %5.1f median %5.1f max %5.1f us | gap between items mean %4.2f us | idle before the last end: mean %5.1f "
          "median %5.1f max %5.1f us" % (name, items, per, (e1 - s0) * T / 1e3, m(busy), m(first), first[G // 2], first[-1], m(gaps),
                                         m(last), last[G // 2], last[-1]))
    # is the spread systematic?  mean time inside items and mean finish time (before the launch's last end) per XCD (= blockIdx % 8)
    for xcd in range(8):
        wgs = [wg for wg in range(G) if wg % 8 == xcd]
        cyc = m([sum((t[wg + i * G, 4] - t[wg + i * G, 0]).item() for i in range(per)) for wg in wgs])
        print("    XCD %d: %8.0f shader cycles in items = %4.0f MHz |" % (xcd, cyc, cyc / m([busy_by_wg[wg] for wg in wgs])), end="")
        print(" inside items %7.1f us   finishes %5.1f us before the last end (min %5.1f max %5.1f)" % (
            m([busy_by_wg[wg] for wg in wgs]), m([(e1 - t[wg + (per - 1) * G, 6].item()) * T / 1e3 for wg in wgs]),
            min((e1 - t[wg + (per - 1) * G, 6].item()) * T / 1e3 for wg in wgs), max((e1 - t[wg + (per - 1) * G, 6].item()) * T / 1e3 for wg in wgs)))
H.set_wino_mode(1); H.set_compute("f32")
