"""s_memtime segment sums of the large-tile bf16 gather-GEMM k-loop (bq_kernel), one lane per wave of the first 256 WGs.
Build:  hipcc ... -DDPIG_TRACE -> scripts/ubench/libdpig_trace.so (scripts/ubench/build.sh)
run:    DPIG_LIB_PATH=scripts/ubench/libdpig_trace.so python scripts/ubench/trace_q.py [C] [K] [variant]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dpig_amd.hip_ops as H
from dpig_amd import _lib
dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
var = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N, Hh, W = 8, 128, 128
x = torch.randn(N, Hh, W, C, device=dev).to(torch.bfloat16)
w = torch.randn(3, 3, C, K, device=dev) * 0.05
w._dpig_shadow = H.filter_shadows(w)
H.set_large_tile(2, var)
for _ in range(3):
    y = H.conv2d_fwd(x, w, None, act=1)
torch.cuda.synchronize()
n = 256 * 8 * 16
buf = (ctypes.c_ulonglong * n)()
h = _lib.lib()
h.dpig_debug_bq_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert h.dpig_debug_bq_prof_read(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 16).astype(np.float64)
names = ["PA issue (16 reads + NA DMA)", "PA counted vmcnt", "PA fragments home (lgkmcnt 0)", "barrier 1 (both phases)", "16 MFMAs issued (both phases)",
         "barrier 2 (both phases)", "PB issue (8 reads + cursor + NA+2NB DMA)", "PB vmcnt + fragments home"]
nkt = a[0, 0, 8]
print("layer N%d %dx%d C%d K%d variant %d: %d k-tiles per workgroup; s_memtime ticks per k-tile and wave (two phases)" % (N, Hh, W, C, K, var, nkt))
for g, sl in (("group 0 (waves 0-3)", slice(0, 4)), ("group 1 (waves 4-7)", slice(4, 8))):
    print(g)
    tot = 0.0
    for i, nm in enumerate(names):
        v = a[:, sl, i].reshape(-1) / nkt
        tot += np.median(v)
        print("  %-44s median %7.0f  p10 %7.0f  p90 %7.0f" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
    print("  sum of medians %.0f; prologue + loop per k-tile %.0f; epilogue total %.0f ticks" % (
        tot, np.median(a[:, sl, 9]) / nkt, np.median(a[:, sl, 10])))
