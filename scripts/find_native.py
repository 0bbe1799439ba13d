"""Which host lines still launch torch-native kernels inside a training step?  One eager step under torch.profiler with
stacks; prints every non-dpig device kernel with the python frames that issued it.
usage: python scripts/find_native.py [--workload df256] [--dtype bf16]"""
import argparse, os, sys, importlib, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="market128"); ap.add_argument("--dtype", default="f32")
a = ap.parse_args()
from dpig_amd import synthetic
from dpig_amd.trainer import Config
mod, cls, cfgk, B, _ = bench.WORKLOADS[a.workload]
dev = torch.device("cuda:0")
np.random.seed(0)
cfg = Config(batch_size=B, compute_dtype=a.dtype, **cfgk)
tr = getattr(importlib.import_module("dpig_amd." + mod), cls)(cfg, dev)
bg = synthetic.to_device(synthetic.make_batch(B, img_H=cfg.img_H, img_W=cfg.img_W, seed=100), dev)
bd = synthetic.to_device(synthetic.make_batch(B, img_H=cfg.img_H, img_W=cfg.img_W, seed=101), dev)
tr.init_net(bg); tr.step = 1
for _ in range(2):
    tr.train_step(bg, bd)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(bg, bd)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue            # count the leaf op only
    frames = [f for f in (ev.stack or []) if "dpig_amd" in f or "bench.py" in f or "disentangled" in f][:3]
    key = (ev.name, " <- ".join(f.split("/")[-1] for f in frames))
    agg[key][0] += 1
    agg[key][1] += ev.device_time_total
for (name, where), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.1f us %4d  %-28s %s" % (us, n, name, where))

# ---- autograd nodes of the generator loss that are not this package's Functions ------------------------------------------------
g_loss, embs, out = tr._g_forward(bg)
seen, stack, names = set(), [g_loss.grad_fn], collections.Counter()
while stack:
    fn = stack.pop()
    if fn is None or fn in seen:
        continue
    seen.add(fn)
    names[type(fn).__name__] += 1
    for nf, _ in fn.next_functions:
        stack.append(nf)
print({k: v for k, v in names.items() if not k.startswith("_") or "Backward" in k and not k.startswith("_")})
print(sorted(names.items(), key=lambda kv: -kv[1])[:60])
