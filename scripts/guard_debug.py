"""Dev aid for the guard-page harness: one w16 stage-I forward/backward on guarded memory with per-tap / per-gradient NaN counts.
    python scripts/guard_debug.py hi|lo|none      (DPIG_GUARD_FILL / DPIG_TWO_STREAM / AMD_SERIALIZE_KERNEL from the environment)"""
import faulthandler
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
faulthandler.enable(all_threads=True)
mode = sys.argv[1]
if mode != "none":
    import conftest
    conftest.install_guard_allocator(mode)
import numpy as np
import torch
import dpig_amd.hip_ops as H
import dpig_amd.tflib as lib
from dpig_amd import models, slim, synthetic, autograd as A
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg, gan_loss
dev = torch.device("cuda:0")
np.random.seed(0)
tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=2, conv_hidden_num=16, z_num=8), dev)
bg = synthetic.to_device(synthetic.make_batch(2, seed=21), dev)
for k, v in bg.items():
    print("batch", k, tuple(v.shape), v.dtype, "nan", int(torch.isnan(v.float()).sum()), "ptr %x" % v.data_ptr(), flush=True)
tr.init_net(bg)
print("init ok; params nan:", sum(int(torch.isnan(p).sum()) for p in tr.G_flat.params), flush=True)
models.TAPS = {}
tr.G_flat.zero_grad()
embs, _ = tr.encode(bg)
torch.cuda.synchronize()
print("embs nan", int(torch.isnan(embs).sum()), flush=True)
G, _ = tr.generate(embs, bg["pose"])
torch.cuda.synchronize()
for k, v in models.TAPS.items():
    print("tap %-14s %-22s nan %d" % (k, tuple(v.shape), int(torch.isnan(v.float()).sum())), flush=True)
models.TAPS = None
print("G nan", int(torch.isnan(G).sum()), flush=True)
l1 = A.l1_mean(G, bg["x"])
d_fake = tr.discriminate(G)
print("d_fake nan", int(torch.isnan(d_fake).sum()), "l1", float(l1), flush=True)
(l1 + d_fake.mean()).backward()
tr.G_flat.finalize()
torch.cuda.synchronize()
bad = [(n, int(torch.isnan(p._dpig_grad).sum())) for n, p in lib._params.items() if hasattr(p, "_dpig_grad") and torch.isnan(p._dpig_grad).any()]
print("grads with nan:", len(bad), bad[:12], flush=True)
print("DEBUG DONE", flush=True)
