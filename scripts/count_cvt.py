"""Which call sites convert between fp32 and bf16 in one replay-able step (DPIG_WORKLOAD=df256|market128): count by caller."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dpig_amd import hip_ops as H, synthetic
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
from dpig_amd.trainer_256 import DPIG_Encoder_GAN_BodyROI_256
dev = torch.device("cuda:0"); np.random.seed(0)
if os.environ.get('DPIG_WORKLOAD', 'df256') == 'df256':
    B = 8
    tr = DPIG_Encoder_GAN_BodyROI_256(Config(batch_size=B, img_H=256, img_W=256, compute_dtype='bf16'), dev)
    mk = lambda seed: synthetic.keypoints_only(synthetic.to_device(synthetic.make_batch(B, img_H=256, img_W=256, seed=seed), dev))
else:
    B = 16
    tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, compute_dtype='bf16'), dev)
    mk = lambda seed: synthetic.keypoints_only(synthetic.to_device(synthetic.make_batch(B, seed=seed), dev))
b0, b1 = mk(1), mk(2)
tr.init_net(b0); tr.step = 1
tr.train_step(b0, b1)
cnt = collections.Counter()
def wrap(name, fn):
    def f(t, out=None):
        if t is not None and ((name == "to_f32" and t.dtype == torch.bfloat16) or (name == "to_bf16" and t.dtype == torch.float32)):
            st = traceback.extract_stack(limit=6)[:-1]
            key = name + " " + str(tuple(t.shape)) + " <- " + " <- ".join("%s:%d" % (os.path.basename(s.filename), s.lineno) for s in reversed(st[-4:]))
            cnt[key] += 1
        return fn(t, out)
    return f
H.to_f32, H.to_bf16 = wrap("to_f32", H.to_f32), wrap("to_bf16", H.to_bf16)
tr.train_step(b0, b1)
torch.cuda.synchronize()
for k, v in cnt.most_common(60):
    print(v, k)
print("total", sum(cnt.values()))
