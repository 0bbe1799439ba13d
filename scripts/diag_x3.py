"""Per-parameter difference of the E+G trunk gradients (linear read-out) between the exact fp32 kernels and the
split-bf16 pipe ('bf16x3') on the same weights and batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dpig_amd.hip_ops as H
import dpig_amd.tflib as lib
from dpig_amd import slim, synthetic
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
dev = torch.device("cuda:0")
B, HID, ZN = 2, 64, 16
np.random.seed(0)
batch = synthetic.to_device(synthetic.make_batch(B, seed=33), dev)
tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=B, conv_hidden_num=HID, z_num=ZN), dev)
tr.init_net(batch)
r = torch.randn(tuple(batch["x"].shape), device=dev)
grads = {}
for mode in ("f32", "bf16x3"):
    H.set_compute(mode)
    tr.G_flat.zero_grad()
    embs, _ = tr.encode(batch)
    G, _ = tr.generate(embs, batch["pose"])
    G.backward(r)
    tr.G_flat.finalize()
    grads[mode] = {n: p._dpig_grad.detach().clone() for n, p in lib._params.items() if hasattr(p, "_dpig_grad") and p._dpig_grad is not None}
    print(mode, "G", float(G.abs().max()))
H.set_compute("f32")
# sensitivity of the exact path: the same weights perturbed by 4e-6 relative (the split's operand truncation is <= 3.8e-6)
with torch.no_grad():
    w0 = tr.G_flat.flat.detach().clone()
    tr.G_flat.flat.mul_(1.0 + 4e-6 * torch.randn_like(tr.G_flat.flat))
tr.G_flat.zero_grad()
embs, _ = tr.encode(batch); G, _ = tr.generate(embs, batch["pose"]); G.backward(r); tr.G_flat.finalize()
pert = {n: p._dpig_grad.detach().clone() for n, p in lib._params.items() if hasattr(p, "_dpig_grad") and p._dpig_grad is not None}
with torch.no_grad():
    tr.G_flat.flat.copy_(w0)
rows = []
rel2 = lambda a, b: ((a - b).double().norm() / b.double().norm().clamp_min(1e-30)).item()
tot = lambda gd: torch.cat([gd[n].flatten() for n in grads["f32"]])
print("whole-gradient L2 relative difference: bf16x3 vs f32 %.3e ; f32 with 4e-6-perturbed weights vs f32 %.3e" % (
    rel2(tot(grads["bf16x3"]), tot(grads["f32"])), rel2(tot(pert), tot(grads["f32"]))))
mx = lambda gd: max((gd[n] - grads["f32"][n]).abs().max().item() / max(grads["f32"][n].abs().max().item(), 1e-12) for n in grads["f32"])
print("worst per-parameter max-norm difference: bf16x3 %.3e ; perturbed f32 %.3e" % (mx(grads["bf16x3"]), mx(pert)))
for n, g in grads["f32"].items():
    d = (grads["bf16x3"][n] - g).abs().max().item() / max(g.abs().max().item(), 1e-12)
    rows.append((d, n, tuple(g.shape)))
for d, n, sh in sorted(rows, reverse=True)[:25]:
    print("%.3e  %-50s %s" % (d, n, sh))
