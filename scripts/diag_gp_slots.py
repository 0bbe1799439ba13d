"""Level tensors the bf16 fused gradient-penalty call leaves in its workspace against the bf16-emulating sweeps (oracle/gp_sweeps.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpig_amd import hip_ops as H
from dpig_amd._lib import workspace
from oracle import models as OM, gp_sweeps as GS
dev = torch.device("cuda:0")
shape, dim = (2, 128, 64, 3), 64
g = torch.Generator().manual_seed(7)
B = shape[0]
P = OM.ParamStore(seed=17)
real = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1).float().double()
fake = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1).float().double()
alpha = torch.rand(B, generator=g, dtype=torch.float64).float().double()
D_o = lambda t: OM.dcgan_discriminator(P, t, "wgan-gp", dim=dim)
D_o(real[:1])
names = [n for n in OM.d_var_names(P)]
with torch.no_grad():
    for n in names:
        if not n.endswith(("Filters", "Output.W")):
            P.p[n].add_(0.2 * (torch.rand(P.p[n].shape, generator=g, dtype=torch.float64) - 0.5))
        P.p[n].copy_(P.p[n].float().double())
        if n.endswith("Filters") and not n.endswith("Discriminator.1.Filters"):
            P.p[n].copy_(P.p[n].float().to(torch.bfloat16).double())
params = {n: P.p[n].detach().float().to(dev).contiguous() for n in names}
rd, fd, ad = real.float().to(dev), fake.float().to(dev), alpha.float().to(dev)
pen, slopes, grads = H.gp_double_backward(params, rd, fd, ad, 10.0, dim=dim, grads=True, compute=H.COMPUTE_BF16_STORE)
torch.cuda.synchronize()
ws = workspace.get(1, dev)[0]
taps = {}
GS.gp_sweeps(P.p, real, fake, alpha, 10.0, dim=dim, store=GS.bf16_round, taps=taps)
up = lambda v: (v + 255) & ~255
Hs, Ws, Cs = [128], [64], [3]
for l in range(1, 5):
    Hs.append((Hs[-1] + 1) // 2); Ws.append((Ws[-1] + 1) // 2); Cs.append(dim << (l - 1))
n = [B * Hs[l] * Ws[l] * Cs[l] for l in range(5)]
off = 0
def take(nbytes):
    global off
    o = off; off += up(nbytes); return o
img = {}
for nm in ("xhat", "gin", "u0"):
    o = take(n[0] * 4); img[nm] = ws[o:o + n[0] * 4].view(torch.float32).reshape(B, Hs[0], Ws[0], 3)
slots = ["Z", "A", "DA", "DZ", "V", "UB", "ZB", "T"]
lev = {}
for l in range(1, 5):
    for s in slots:
        o = take(n[l] * 2); lev["%s%d" % (s, l)] = ws[o:o + n[l] * 2].view(torch.bfloat16).reshape(B, Hs[l], Ws[l], Cs[l])
def cmp(nm, got, ref):
    got = got.double().cpu(); e = got - ref
    ulp = (e.abs() > 1e-6 * ref.abs().max()).double().mean().item()
    print("%-6s rel-L2 %.2e  max-err/max %.2e  elements that differ %.3f %%" % (nm, e.norm() / ref.norm(), e.abs().max() / ref.abs().max(), 100 * ulp))
order = ["xhat", "A1", "Z2", "A2", "Z3", "A3", "Z4", "A4", "DA4", "DZ4", "DA3", "DZ3", "DA2", "DZ2", "DZ1", "gin", "u0", "V1", "UB1", "V2", "UB2", "V3", "UB3", "V4", "UB4", "ZB4", "ZB3", "ZB2", "ZB1"]
for nm in order:
    got = img[nm] if nm in img else lev[nm]
    if nm in ("ZB4",):
        pass
    cmp(nm, got, taps[nm])
