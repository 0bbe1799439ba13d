"""The torch-native device launches (element-wise aten ops, copies with their calling op) left in one stage-I step (eager, torch profiler).
DPIG_DTYPE=bf16 DPIG_WORKLOAD=market128|df256 python scripts/diag_copies.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from dpig_amd import synthetic
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
from dpig_amd.trainer_256 import DPIG_Encoder_GAN_BodyROI_256
dev = torch.device("cuda:0"); np.random.seed(0)
DT = os.environ.get('DPIG_DTYPE', 'bf16')
if os.environ.get('DPIG_WORKLOAD', 'market128') == 'df256':
    B = 8
    tr = DPIG_Encoder_GAN_BodyROI_256(Config(batch_size=B, img_H=256, img_W=256, compute_dtype=DT), dev)
    mk = lambda seed: synthetic.keypoints_only(synthetic.to_device(synthetic.make_batch(B, img_H=256, img_W=256, seed=seed), dev))
else:
    B = 16
    tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, compute_dtype=DT), dev)
    mk = lambda seed: synthetic.keypoints_only(synthetic.to_device(synthetic.make_batch(B, seed=seed), dev))
b0, b1 = mk(1), mk(2)
tr.init_net(b0); tr.step = 1
for _ in range(2): tr.train_step(b0, b1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train_step(b0, b1)
    torch.cuda.synchronize()
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.Counter(); tim = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU: continue
    n = ev.name
    if not (n in ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::cat", "aten::sum", "aten::mean", "aten::div", "aten::sub",
                                             "aten::neg", "aten::fill_", "aten::zero_", "aten::clone", "aten::_to_copy", "aten::mm", "aten::addmm", "aten::sigmoid", "aten::abs",
                                             "aten::index", "aten::where", "aten::clamp", "aten::exp", "aten::log", "aten::sqrt", "aten::pow", "aten::binary_cross_entropy_with_logits")):
        continue
    ks = getattr(ev, "kernels", None) or []
    dt = sum(k.duration for k in ks)
    if not ks and n != "aten::copy_": continue
    site = "?"
    if n == "aten::copy_":
        par = ev.cpu_parent; chain = []
        while par is not None and len(chain) < 4: chain.append(par.name); par = par.cpu_parent
        n = "copy_ <- " + " <- ".join(chain) + " " + str(ev.input_shapes)[:60]
    for fr in (ev.stack or []):
        if ("generation_amd/" in fr or "dpig_amd" in fr or "bench.py" in fr) and "scripts/" not in fr:
            site = fr.split("/")[-1]; break
    agg[(n, site)] += 1; tim[(n, site)] += dt
print("device launches from aten ops in one eager step (count x op, device time):")
for k, c in sorted(agg.items(), key=lambda kv: -tim[kv[0]]):
    print("%4d x %-100s %8.1f us  %s" % (c, k[0], tim[k], k[1]))
