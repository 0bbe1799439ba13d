"""Where a Winograd F(4x4,3x3) workgroup spends its life: s_memtime stamps (dpig_debug_wino_trace) of one launch per layer.
   python scripts/trace_wino4.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpig_amd import hip_ops as H
dev = torch.device("cuda:0")
lib = H.lib()
lib.dpig_debug_wino_trace.argtypes = [ctypes.c_void_p]
lib.dpig_debug_wino_trace.restype = ctypes.c_int
g = torch.Generator(device=dev).manual_seed(0)
H.set_compute("f32w"); H.set_wino4_mode(2)
print("%-22s %6s %5s | %8s %9s %9s %9s %9s | %9s %8s" % ("layer", "items", "nch", "arrive", "prologue", "loop", "outxf y0", "rows 1-3", "total", "cyc/chunk"))
for name, N, Hh, W, C in [("C128 128x64", 16, 128, 64, 128), ("C128 48x48", 112, 48, 48, 128), ("C256 128x64", 16, 128, 64, 256), ("C256 24x24", 112, 24, 24, 256),
                          ("C512 64x32", 16, 64, 32, 512), ("C768 32x16", 16, 32, 16, 768)]:
    x = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
    w = (torch.rand((3, 3, C, C), device=dev, generator=g) * 2 - 1) * 0.02
    b = torch.rand((C,), device=dev, generator=g)
    w._dpig_wino4 = H.wino4_images(w)
    H.conv2d_fwd(x, w, b, act=1)
    d = H._desc(N, Hh, W, C, C, 3, 3, 1, C, C)
    split = max(1, int(lib.dpig_conv2d_wino4_workspace_bytes(ctypes.byref(d), 0)) // (N * Hh * W * C * 4))
    wgs = N * (Hh // 4) * (W // 4) // 32 * (C // 64) * split
    buf = torch.zeros(wgs * 8, dtype=torch.int64, device=dev)
    lib.dpig_debug_wino_trace(ctypes.c_void_p(buf.data_ptr()))
    H.conv2d_fwd(x, w, b, act=1)
    torch.cuda.synchronize()
    lib.dpig_debug_wino_trace(None)
    t = buf.view(wgs, 8).cpu().double()
    seg = [(t[:, i + 1] - t[:, i]).mean().item() for i in range(4)]
    tot = (t[:, 4] - t[:, 0]).mean().item()
    span = (t[:, 4].max() - t[:, 0].min()).item()
    rounds = (wgs + 255) // 256
    nch = C // 8 // split
    if os.environ.get("DPIG_WINO4_KO") == "32":      # epilogue stamp mode: row 1 of the output phase
        m = lambda a, b: (t[:, b] - t[:, a]).mean().item()
        print("%-22s row 1: transform + staging stores %6.0f | barrier %6.0f | sums + epilogue + store issue %6.0f | barrier %6.0f" % (name, m(1, 2), m(2, 3), m(3, 7), m(7, 4)))
        continue
    if os.environ.get("DPIG_WINO4_KO") == "192":     # prologue stamp mode
        m = lambda a, b: (t[:, b] - t[:, a]).mean().item()
        print("%-22s prologue: setup %6.0f | issue of 6 gathers + 9 fragment loads %6.0f | own loads home %6.0f | barrier %6.0f | transform + barrier %6.0f" % (
            name, m(0, 1), m(1, 2), m(2, 3), m(3, 7), m(7, 4)))
        continue
    arrive = (t[:, 7] - t[:, 0]).mean().item()
    print("%-22s %6d %5d | %8.0f %9.0f %9.0f %9.0f %9.0f | %9.0f %8.0f  (split %d)" % (name, wgs, nch, arrive, seg[0], seg[1], seg[2], seg[3], tot, seg[1] / nch, split))
H.set_wino4_mode(1); H.set_compute("f32")
