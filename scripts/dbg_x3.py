import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dpig_amd import hip_ops as H, synthetic, _lib
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
dev = torch.device("cuda:0"); np.random.seed(0)
tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=16, compute_dtype='bf16x3'), dev)
b0 = synthetic.to_device(synthetic.make_batch(16, seed=1), dev)
tr.init_net(b0); tr.step = 1
print("init done", flush=True)
torch.cuda.synchronize()
tr.train_step(b0, b0)
torch.cuda.synchronize()
print("step done", flush=True)
