import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_model_gpu as T
from dpig_amd import autograd as A
from dpig_amd.trainer import gan_loss
dev = torch.device("cuda:0")
tr, gb, P, ob, OM = T._setup(dev)
tr.D_flat.set_requires_grad(False)
with torch.no_grad():
    _, Go = OM.stage1_forward(P, ob, hidden_num=T.HID, z_num=T.ZNUM)
    embs, _ = tr.encode(gb); G, _ = tr.generate(embs, gb["pose"])
Gh = G.double().cpu()
print("G_o stats: max %.3e mean|.| %.3e ; err max %.3e mean %.3e" % (Go.abs().max(), Go.abs().mean(), (Gh-Go).abs().max(), (Gh-Go).abs().mean()))
def ograd(Gin):
    Gi = Gin.clone().requires_grad_(True)
    taps = {}
    l = OM.gan_loss("dcgan", None, OM.dcgan_discriminator(P, Gi, "dcgan", taps=taps))[0]; l.backward(); return Gi.grad, taps
def hgrad(Gin):
    Gi = Gin.float().to(dev).requires_grad_(True)
    l = gan_loss(tr.wgan_gp, None, tr.discriminate(Gi))[0]; l.backward(); return Gi.grad
g0, taps = ograd(Go)
print("D.1 act stats: mean|.| %.3e  frac |y|<1e-6: %.3e" % (taps["D.1"].abs().mean(), (taps["D.1"].abs() < 1e-6).double().mean()))
print("oracle(Go) vs hip(Go)      ", T._rel(hgrad(Go), g0))
print("oracle(Go) vs hip(G_hip)   ", T._rel(hgrad(Gh), g0))
print("oracle(Go) vs oracle(G_hip)", T._rel(ograd(Gh)[0], g0))
noise = torch.randn_like(Go) * (Gh-Go).abs().mean()
print("oracle(Go) vs oracle(Go+n) ", T._rel(ograd(Go + noise)[0], g0))
print("oracle(Go) vs oracle(Go.float())", T._rel(ograd(Go.float().double())[0], g0))
d = Gh - Go
print("delta per-image mean", d.mean(dim=(1,2,3)).numpy(), "per-channel mean", d.mean(dim=(0,1,2)).numpy())
print("delta corr with Go", (d*Go).sum().item() / (d.norm()*Go.norm()).item())
print("rel delta/|Go| elementwise median", (d.abs()/(Go.abs()+1e-30)).median().item())
for t in (1e-3, 1e-2, 1e-1, 0.5, 1.0, 2.0):
    print("t=%g  rel grad change %.3e" % (t, T._rel(ograd(Go + t*d)[0], g0)))
