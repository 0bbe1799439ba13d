"""Per-tensor error table of dpig_gp_double_backward in bf16 storage (fused), the taped bf16 path and the fp32 fused call against the fp64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dpig_amd.tflib as lib
from dpig_amd import hip_ops as H, slim
from dpig_amd.trainer import gradient_penalty
from dpig_amd.wgan_gp import WGAN_GP
from oracle import models as OM
dev = torch.device("cuda:0")
shape, dim = tuple(int(v) for v in (sys.argv[1:5] or (2, 128, 64, 3))), 64
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
gp_ref = OM.gradient_penalty(D_o, real, fake, alpha, 10.0)
refs = dict(zip(names, torch.autograd.grad(gp_ref, [P.p[n] for n in names], allow_unused=True)))
params = {n: P.p[n].detach().float().to(dev).contiguous() for n in names}
rd, fd, ad = real.float().to(dev), fake.float().to(dev), alpha.float().to(dev)
res = {}
for tag, comp in (("fused f32", H.COMPUTE_F32), ("fused bf16", H.COMPUTE_BF16_STORE)):
    pen, slopes, grads = H.gp_double_backward(params, rd, fd, ad, 10.0, dim=dim, grads=True, compute=comp)
    res[tag] = (pen.item(), grads)
lib.delete_all_params(); slim.reset_scopes(); lib.set_device(dev)
for n, v in P.state_numpy().items():
    lib.param(n, v, trainable=P.trainable[n])
H.set_compute("bf16")
wg = WGAN_GP(MODE="wgan-gp", BATCH_SIZE=B)
gp = gradient_penalty(lambda t: wg.DCGANDiscriminator(t.permute(0, 3, 1, 2), input_dim=3), rd, fd, 10.0, ad)
gp.backward()
res["taped bf16"] = (gp.item(), {n: lib._params[n].grad for n in names})
H.set_compute("f32")
print("oracle penalty %.6f" % gp_ref.item())
for tag, (pv, gr) in res.items():
    print("%s: penalty %.6f (rel %.2e)" % (tag, pv, abs(pv - gp_ref.item()) / abs(gp_ref.item())))
    for n in names:
        if refs[n] is None or gr.get(n) is None:
            continue
        r = refs[n]; e = gr[n].double().cpu() - r
        print("   %-28s max-err/max-ref %.2e   rel-L2 %.2e   cos %.6f" % (n, e.abs().max() / r.abs().max(), e.norm() / r.norm(),
              float((gr[n].double().cpu() * r).sum() / (gr[n].double().cpu().norm() * r.norm()))))
from oracle import gp_sweeps as GS
import time
t0 = time.time()
pen_e, g_e = GS.gp_sweeps(P.p, real, fake, alpha, 10.0, dim=dim, store=GS.bf16_round)
print("emulation (fp64 sweeps, bf16 stores) took %.1f s; penalty %.6f; HIP fused bf16 vs emulation: penalty rel %.2e" % (
    time.time() - t0, pen_e.item(), abs(res["fused bf16"][0] - pen_e.item()) / abs(pen_e.item())))
for n in names:
    if n.endswith("Output.b") or n not in g_e: continue
    r = g_e[n]; e = res["fused bf16"][1][n].double().cpu() - r
    if float(r.abs().max()) == 0: continue
    print("   %-28s max-err/max-ref %.2e   rel-L2 %.2e" % (n, e.abs().max() / r.abs().max(), e.norm() / r.norm()))
a, b = res["fused bf16"][1], res["taped bf16"][1]
print("fused bf16 vs taped bf16:")
for n in names:
    if a.get(n) is None or b.get(n) is None: continue
    e = (a[n] - b[n]).double(); print("   %-28s max-err/max %.2e  rel-L2 %.2e" % (n, e.abs().max() / b[n].abs().max(), e.norm() / b[n].double().norm()))
