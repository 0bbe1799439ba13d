import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_model_gpu as T
dev = torch.device("cuda:0")
tr, gb, P, ob, OM = T._setup(dev)
with torch.no_grad():
    embs_o, G_o = OM.stage1_forward(P, ob, hidden_num=T.HID, z_num=T.ZNUM)
    taps = {}
    d_o = OM.dcgan_discriminator(P, G_o, taps=taps)
    embs, _ = tr.encode(gb); G, _ = tr.generate(embs, gb["pose"])
    print("embs", T._rel(embs, embs_o), "G", T._rel(G, G_o))
    d = tr.discriminate(G_o.float().to(dev))
    print("D(G_o)", T._rel(d, d_o), d.cpu().numpy(), d_o.numpy())
# D input-gradient with identical input
Gi = G_o.clone().requires_grad_(True)
do = OM.dcgan_discriminator(P, Gi); lo = OM.gan_loss("dcgan", None, do)[0]; lo.backward()
import dpig_amd.autograd as A
Gh = G_o.float().to(dev).requires_grad_(True)
tr.D_flat.set_requires_grad(False)
dh = tr.discriminate(Gh); lh = A.sce_mean(dh, 1.0); lh.backward()
print("loss", lh.item(), lo.item(), "dD/dG rel", T._rel(Gh.grad, Gi.grad))
