"""Single process, world-size-1 gloo group, SyncBN path forced: does the trainer's segmented capture work?"""
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable(); faulthandler.dump_traceback_later(100, exit=True)
import numpy as np, torch, torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29577"
dist.init_process_group("gloo", rank=0, world_size=1)
from dpig_amd import synthetic, autograd as A
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
dev = torch.device("cuda:0")
# force the cross-rank BN op although the world has one rank
orig = A.batchnorm
def forced(x, scale, offset, eps=1e-5, act=0, alpha=0.2, stats=None):
    return A._SyncBatchNormFn.apply(x, scale, offset, eps, act, alpha, None)
A.batchnorm = forced
import dpig_amd.tflib.ops.batchnorm as BNmod
B = 2
np.random.seed(0)
tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, conv_hidden_num=16, z_num=8, sync_bn=True, split_backward=(sys.argv[1:] == ["split"])), dev)
bg = synthetic.to_device(synthetic.make_batch(B, seed=21), dev); bd = synthetic.to_device(synthetic.make_batch(B, seed=22), dev)
tr.init_net(bg); tr.step = 1
tr._sync_bn_active = lambda: True
tr.allreduce.enabled = False
o = tr.train_step(bg, bd); torch.cuda.synchronize(); print("eager ok", float(o["d_loss"])); sys.stdout.flush()
tr.enable_graphs(bg, bd, warmup=1)
print("captured", type(tr._graphs[0]).__name__, tr._graphs[0].segments, tr._graphs[2].segments); sys.stdout.flush()
o = tr.train_step(bg, bd); torch.cuda.synchronize(); print("replay ok", float(o["d_loss"]))
