"""s_memtime timeline of the bf16 gather-GEMM k-loop (bg_kernel), one lane per wave of the first 256 workgroups.
Build: hipcc ... -DDPIG_TRACE -> scripts/ubench/libdpig_trace.so (scripts/ubench/build.sh);
run:   DPIG_BF16_HALO=0 DPIG_LIB_PATH=scripts/ubench/libdpig_trace.so python scripts/ubench/trace_bf16.py [C] [K]
Stamps per k-tile: 1 tile start (behind the barrier), 2 fragments requested + next tile's DMA issued, 3 last MFMA issued,
4 DMA landed (vmcnt(0)), [barrier], next 1."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dpig_amd.hip_ops as H
from dpig_amd import _lib
dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N, Hh, W = 8, 128, 128
x = torch.randn(N, Hh, W, C, device=dev).to(torch.bfloat16)
w = torch.randn(3, 3, C, K, device=dev) * 0.05
w._dpig_shadow = H.filter_shadows(w)
for _ in range(3):
    y = H.conv2d_fwd(x, w, None, act=1)
torch.cuda.synchronize()
n = 256 * 4 * 160
buf = (ctypes.c_ulonglong * n)()
h = _lib.lib()
h.dpig_debug_bf_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert h.dpig_debug_bf_trace_read(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256 * 4, 160)
slot = (a >> np.uint64(56)).astype(np.int64)
t = (a & np.uint64((1 << 56) - 1)).astype(np.int64)
seg = {"1->2 frag0 + DMA issue": [], "2->3 MFMAs (16) + frag reads": [], "3->4 wait for DMA": [], "4->1 barrier": [], "tile period 1->1": []}
life = []
for r in range(a.shape[0]):
    s, tt = slot[r], t[r]
    if s[0] != 0:
        continue
    idx = {k: np.where(s == k)[0] for k in range(1, 7)}
    for i in range(len(idx[1]) - 1):
        j = idx[1][i]
        if j + 4 < 160 and s[j + 1] == 2 and s[j + 2] == 3 and s[j + 3] == 4 and s[j + 4] == 1:
            seg["1->2 frag0 + DMA issue"].append(tt[j + 1] - tt[j]); seg["2->3 MFMAs (16) + frag reads"].append(tt[j + 2] - tt[j + 1])
            seg["3->4 wait for DMA"].append(tt[j + 3] - tt[j + 2]); seg["4->1 barrier"].append(tt[j + 4] - tt[j + 3])
            seg["tile period 1->1"].append(tt[j + 4] - tt[j])
    if len(idx[5]) and len(idx[6]):
        life.append((tt[idx[1][0]] - tt[0], tt[idx[6][0]] - tt[idx[5][0]], tt[idx[6][0]] - tt[0]))
print("layer N%d %dx%d C%d K%d; ticks of s_memtime (100 MHz? calibrate against the MFMA segment)" % (N, Hh, W, C, K))
for k, v in seg.items():
    v = np.array(v)
    print("%-32s n=%6d  median %7.0f  mean %7.0f  p10 %7.0f  p90 %7.0f" % (k, len(v), np.median(v), v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
life = np.array(life)
print("prologue (start -> first tile) median %.0f; epilogue median %.0f; whole workgroup median %.0f" % tuple(np.median(life, axis=0)))
