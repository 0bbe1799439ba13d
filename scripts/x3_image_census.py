"""Which tensors still get their split32 image from a dpig_split32 pass (rather than from a producing epilogue) in one
bf16x3 stage-I step, and what the passes move."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dpig_amd import hip_ops as H, synthetic
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
dev = torch.device("cuda:0"); np.random.seed(0)
tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=16, compute_dtype='bf16x3'), dev)
b0 = synthetic.to_device(synthetic.make_batch(16, seed=1), dev)
tr.init_net(b0); tr.step = 1
made = collections.Counter(); emitted = collections.Counter(); byt = [0, 0]
orig_split, orig_room = H.split32, H._image_room
def spy_split(t, reuse=1):
    had = getattr(t, "_dpig_s32", None) is not None
    r = orig_split(t, reuse)
    if r is not None and not had:
        made[tuple(t.shape)] += 1; byt[0] += t.numel() * 8
    return r
H.split32 = spy_split
import traceback
tr.train_step(b0, b0)
torch.cuda.synchronize()
print("dpig_split32 passes in one step: %d, %.2f GB moved" % (sum(made.values()), byt[0] / 1e9))
for k, v in sorted(made.items(), key=lambda kv: -kv[1] * np.prod(kv[0]))[:14]:
    print("  %3d x %s" % (v, k))
