"""Dev aid: k-loop phase timeline (s_memtime) of the 4 waves of one workgroup of the wgrad kernel."""
import ctypes, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dpig_amd.hip_ops as H
from dpig_amd._lib import lib
dev = torch.device("cuda:0")
N, Hh, W, C, K = 16, 128, 64, int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(N, Hh, W, C, device=dev); dy = torch.randn(N, Hh, W, K, device=dev)
for _ in range(3): H.conv2d_wgrad(x, dy, (3, 3, C, K))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8000)()
lib().dpig_debug_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib().dpig_debug_trace_read(buf, 8000)
a = np.array(buf[:], dtype=np.uint64)
for wv in range(4):
    aw = a[wv * 2000:(wv + 1) * 2000]
    slot = (aw >> np.uint64(56)).astype(int); t = (aw & np.uint64((1 << 56) - 1)).astype(np.int64)
    n = int(np.argmax((slot == 0) & (np.arange(2000) > 0))) or 2000
    slot, t = slot[:n], t[:n]
    d = collections.OrderedDict()
    for i in range(40, n - 1): d.setdefault((slot[i], slot[i + 1]), []).append(t[i + 1] - t[i])
    print("wave %d: stamps %d | " % (wv, n) + " | ".join("%d->%d: %.0f (n=%d)" % (k[0], k[1], np.mean(v), len(v)) for k, v in d.items()))
