"""Diagnostic: stage-I graph in 'bf16' storage mode against the fp64 oracle, piece by piece (forward, D logits, losses,
parameter gradients)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dpig_amd.hip_ops as H
import dpig_amd.tflib as lib
from dpig_amd import slim, synthetic
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg, gan_loss
from oracle import models as OM

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B, HID, ZN = 2, 64, 16
np.random.seed(0)
batch_np = synthetic.make_batch(B, seed=31)
ob = OM.batch_to_torch(batch_np)
P = OM.ParamStore(seed=12)
embs_o, G_o = OM.stage1_forward(P, ob, hidden_num=HID, z_num=ZN)
gl_o, aux = OM.stage1_g_loss(P, ob, hidden_num=HID, z_num=ZN)
gnames = OM.g_var_names(P)
gg = dict(zip(gnames, torch.autograd.grad(gl_o, [P.p[n] for n in gnames], allow_unused=True)))
dl_o, auxd = OM.stage1_d_loss(P, ob, hidden_num=HID, z_num=ZN)
dnames = OM.d_var_names(P)
dg = dict(zip(dnames, torch.autograd.grad(dl_o, [P.p[n] for n in dnames], allow_unused=True)))
lib.set_device(dev)
for n, v in P.state_numpy().items():
    lib.param(n, v, trainable=P.trainable[n])
tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, conv_hidden_num=HID, z_num=ZN, compute_dtype=mode), dev)
batch = synthetic.to_device(batch_np, dev)
tr.init_net(batch)
rel = lambda a, b: (a.double().cpu() - b.double()).abs().max().item() / max(b.abs().max().item(), 1e-12)
with torch.no_grad():
    embs, _ = tr.encode(batch)
    G, _ = tr.generate(embs, batch["pose"])
    Dp, Dn = tr.disc_pair(batch["x"], G)
print("embs %.3e  G %.3e  D_pos %.3e  D_neg %.3e" % (rel(embs, embs_o), rel(G, G_o), rel(Dp, auxd["D_z_pos"]), rel(Dn, auxd["D_z_neg"])))
print("D_pos", Dp.cpu().numpy(), auxd["D_z_pos"].detach().numpy())
print("D_neg", Dn.cpu().numpy(), auxd["D_z_neg"].detach().numpy())
out = tr._d_optim_eager(batch, update=False)
print("d_loss %.6f oracle %.6f" % (float(out["d_loss"]), float(dl_o)))
worst = []
for p, o in zip(tr.D_flat.params, tr.D_flat.offsets):
    n = p.dpig_name
    if dg.get(n) is None:
        continue
    worst.append((rel(tr.D_flat.grad[o:o + p.numel()].view(p.shape), dg[n]), n))
print("D grads worst:", sorted(worst)[-5:])
out = tr._g_optim_eager(batch, update=False)
print("g_loss %.6f oracle %.6f" % (float(out["g_loss"]), float(gl_o)))
worst = []
for p, o in zip(tr.G_flat.params, tr.G_flat.offsets):
    n = p.dpig_name
    if gg.get(n) is None:
        continue
    worst.append((rel(tr.G_flat.grad[o:o + p.numel()].view(p.shape), gg[n]), n))
worst.sort()
print("G grads median %.3e" % worst[len(worst) // 2][0])
for w in worst[-12:]:
    print("   %.3e %s" % w)
