import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dpig_amd import hip_ops as H, synthetic, autograd as A
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
dev = torch.device("cuda:0"); np.random.seed(0)
B = 2
tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, conv_hidden_num=16, z_num=8), dev)
gb = synthetic.to_device(synthetic.make_batch(B, seed=3), dev)
tr.init_net(gb)
grads = lambda: [p._dpig_grad.clone() for p in tr.G_flat.params]
tr.config.split_backward = False
tr._g_optim_eager(gb, update=False); ref = grads()
tr.config.split_backward = True
tr.G_flat.grad.fill_(7.0)
tr._g_optim_eager(gb, update=False); got = grads()
for st in tr._stages:
    for i in list(range(st[1], st[2]))[:4]:
        a, b = got[i], ref[i]
        print(st[0], tr.G_flat.params[i].dpig_name, "equal" if torch.equal(a, b) else "max|diff| %.3e  |ref| %.3e  ratio of norms %.4f  cos %.4f" % (
            (a - b).abs().max(), b.abs().max(), a.norm() / b.norm(), (a * b).sum() / (a.norm() * b.norm())))
