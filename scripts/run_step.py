"""Run a few stage-I steps at the full Market config and print timings (dev aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dpig_amd import synthetic
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
np.random.seed(0)
cfg = Config(batch_size=B)
tr = DPIG_Encoder_GAN_BodyROI_FgBg(cfg, dev)
b0 = synthetic.to_device(synthetic.make_batch(B, seed=1), dev)
b1 = synthetic.to_device(synthetic.make_batch(B, seed=2), dev)
t0 = time.time(); tr.init_net(b0); torch.cuda.synchronize(); print("init %.2fs; G params %d, D params %d" % (time.time() - t0, tr.G_flat.numel, tr.D_flat.numel))
for i in range(2):
    out = tr.train_step(b0, b1)
torch.cuda.synchronize()
print({k: float(v) for k, v in out.items() if v.numel() == 1})
def ev(): return torch.cuda.Event(enable_timing=True)
for name, fn in (("g_optim", lambda: tr.g_optim(b0)), ("d_optim", lambda: tr.d_optim(b1))):
    e0, e1 = ev(), ev(); t0 = time.time(); e0.record()
    for _ in range(steps): fn()
    e1.record(); tcpu = time.time() - t0; torch.cuda.synchronize()
    print("%s: gpu %.2f ms/iter, cpu-issue %.2f ms/iter" % (name, e0.elapsed_time(e1) / steps, tcpu / steps * 1e3))
e0, e1 = ev(), ev(); e0.record()
for _ in range(steps): out = tr.train_step(b0, b1)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
print("train_step: %.2f ms -> %.1f img/s ; mem %.2f GB" % (ms, B / ms * 1e3, torch.cuda.max_memory_allocated() / 2**30))
print({k: float(v) for k, v in out.items() if v.numel() == 1})
