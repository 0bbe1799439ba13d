"""Dev aid: phase timeline (s_memtime) of one wave of one workgroup of the conv-forward kernel."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dpig_amd.hip_ops as H
from dpig_amd._lib import lib
dev = torch.device("cuda:0")
N, Hh, W, C, K = 16, 128, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 256
x = torch.randn(N, Hh, W, C, device=dev); w = torch.randn(3, 3, C, K, device=dev) * 0.05; b = torch.randn(K, device=dev)
for _ in range(3): y = H.conv2d_fwd(x, w, b, act=1)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); y = H.conv2d_fwd(x, w, b, act=1); e1.record(); torch.cuda.synchronize()
wall_us = e0.elapsed_time(e1) * 1e3
NB = 512 * 256
buf = (ctypes.c_ulonglong * NB)()
lib().dpig_debug_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib().dpig_debug_trace_read(buf, NB)
a = np.array(buf[:], dtype=np.uint64).reshape(512, 256)
hw = a[:, 0]
key = [(int(h >> np.uint64(32)) & 15, (int(h) >> 13) & 7, (int(h) >> 8) & 15) for h in hw]      # (xcc, se, cu)
simd = [(int(h) >> 4) & 3 for h in hw]; wid = [int(h) & 15 for h in hw]
slot = (a >> np.uint64(56)).astype(int); t = (a & np.uint64((1 << 56) - 1)).astype(np.int64)
import collections
d = collections.OrderedDict()
for b in range(512):
    sl, tt = slot[b][1:], t[b][1:]
    for i in range(len(sl) - 1):
        if sl[i + 1] == 0 and tt[i + 1] == 0: break
        d.setdefault((sl[i], sl[i + 1]), []).append(tt[i + 1] - tt[i])
print("epilogue phases (mean ticks over 512 workgroups): " + " | ".join("%d->%d: %.0f" % (k[0], k[1], np.mean(v)) for k, v in d.items()))
tiles = (N * Hh * W // 128) * ((K + 127) // 128)
se = (ctypes.c_ulonglong * (4 * tiles))()
lib().dpig_debug_trace_read_se.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib().dpig_debug_trace_read_se(se, 4 * tiles)
se = np.array(se[:], dtype=np.uint64).reshape(tiles, 4)
st, en, hwv, fin = se[:, 0].astype(np.int64), se[:, 1].astype(np.int64), se[:, 2], se[:, 3].astype(np.int64)
bycu = collections.defaultdict(list)
for b in range(tiles):
    h = int(hwv[b]); bycu[((h >> 32) & 15, (h >> 13) & 7, (h >> 8) & 15)].append(b)
rates = []; lifes = []; gaps = []
for k, v in bycu.items():
    s0, e0 = st[v], en[v]
    rates.append((e0.max() - s0.min()) / wall_us); lifes.append((e0 - s0).mean())
    o = np.argsort(s0)
print("kernel wall %.1f us; %d workgroups on %d CUs (%.2f per CU)" % (wall_us, tiles, len(bycu), tiles / len(bycu)))
print("per-CU (first start -> last k-loop end) / kernel wall: mean %.0f ticks/us (min %.0f, max %.0f); mean workgroup life %.0f ticks" % (np.mean(rates), np.min(rates), np.max(rates), np.mean(lifes)))
k0 = list(bycu.keys())[0]; v = bycu[k0]; o = np.argsort(st[v])
print("one CU timeline (start, k-loop end, kernel end; relative):", [(int(st[v][i] - st[v].min()), int(en[v][i] - st[v].min()), int(fin[v][i] - st[v].min())) for i in o])
print("mean epilogue (k-loop end -> last instruction) %.0f ticks" % (fin - en).mean())
