"""Loss trajectories of the headline workload in 'f32' and 'f32w' mode, step by step (eager and graph replay): do they track?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dpig_amd.tflib as lib
from dpig_amd import slim, synthetic
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
dev = torch.device("cuda:0")
B = 16
res = {}
for mode, graph in (("f32", False), ("f32w", False), ("f32w", True), ("f32", True)):
    lib.delete_all_params(); slim.reset_scopes()
    np.random.seed(0)
    tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, compute_dtype=mode), dev)
    bg = synthetic.keypoints_only(synthetic.to_device(synthetic.make_batch(B, seed=100), dev))
    bd = synthetic.keypoints_only(synthetic.to_device(synthetic.make_batch(B, seed=101), dev))
    tr.init_net(bg); tr.step = 1
    if graph:
        tr.enable_graphs(bg, bd)
    out = []
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        o = tr.train_step(bg, bd)
        out.append((float(o["g_loss"]), float(o["L1Loss"]), float(o["g_loss_only"]), float(o["d_loss"])))
    res[(mode, graph)] = out
    print(mode, "graph" if graph else "eager")
    for i, t in enumerate(out):
        print("  step %2d g_loss %.6f L1 %.6f g_only %.6f d_loss %.6f" % ((i,) + t))
