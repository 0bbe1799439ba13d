"""Per-layer-shape timing table of one stage-I step (eager, HIP events around every conv launch)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dpig_amd import hip_ops as H, synthetic
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
from dpig_amd.trainer_256 import DPIG_Encoder_GAN_BodyROI_256
dev = torch.device("cuda:0"); np.random.seed(0)
DT = os.environ.get('DPIG_DTYPE', 'f32')
if os.environ.get('DPIG_WORKLOAD', 'market128') == 'df256':      # DPIG_WORKLOAD=df256 DPIG_DTYPE=bf16 python scripts/layer_table.py
    B = 8
    tr = DPIG_Encoder_GAN_BodyROI_256(Config(batch_size=B, img_H=256, img_W=256, compute_dtype=DT), dev)
    mk = lambda seed: synthetic.keypoints_only(synthetic.to_device(synthetic.make_batch(B, img_H=256, img_W=256, seed=seed), dev))   # (bench.py's default: keypoint-fed stem)
else:
    B = 16
    tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, compute_dtype=DT), dev)
    mk = lambda seed: synthetic.keypoints_only(synthetic.to_device(synthetic.make_batch(B, seed=seed), dev))
b0, b1 = mk(1), mk(2)
tr.init_net(b0); tr.step = 1
for _ in range(2): tr.train_step(b0, b1)
torch.cuda.synchronize(); H.PROFILE = []
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(3): tr.train_step(b0, b1)
e1.record(); torch.cuda.synchronize()
recs = H.PROFILE; H.PROFILE = None
agg = collections.OrderedDict()
for k, f, a, b, lab in recs:
    key = (k, lab); d = agg.setdefault(key, [0, 0.0, 0.0]); d[0] += 1; d[1] += f; d[2] += a.elapsed_time(b) * 1e-3
tot = sum(v[2] for v in agg.values())
print("step %.2f ms (instrumented); conv total %.2f ms" % (e0.elapsed_time(e1) / 3, tot / 3 * 1e3))
CEIL = {'f32': 130e12, 'f32w': 110e12, 'bf16c': 600e12, 'bf16': 1000e12, 'bf16x3': 330e12}[DT]   # (f32w: executed FLOPs of the Winograd launches)   # reference rate for the 'lost' column
tf = sum(v[1] for v in agg.values())
print("executed %.2f TFLOP/step -> %.1f TF average; at %.0f TF everywhere: %.2f ms" % (tf / 3 / 1e12, tf / tot / 1e12, CEIL / 1e12, tf / 3 / CEIL * 1e3))
lost = lambda v: (v[2] - v[1] / CEIL) / 3 * 1e3
for (k, lab), v in sorted(agg.items(), key=lambda kv: -lost(kv[1])):
    print("%-16s N%-3d %3dx%-3d C%-4d K%-4d k%d s%d u%d | n/step %4.1f  ms/step %6.3f  %6.1f TF  lost %6.3f ms" % ((k,) + lab + (v[0] / 3, v[2] / 3 * 1e3, v[1] / v[2] / 1e12, lost(v))))
