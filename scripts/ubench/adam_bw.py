"""Adam update rate on the generator-sized flat buffer (118.5 M parameters, 28 B each per step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dpig_amd.hip_ops as H
dev = torch.device("cuda:0")
n = 118_500_000
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-3; m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
lr = torch.full((1,), 2e-5, device=dev)
for i in range(3): H.adam_step(p, g, m, v, lr, 0.5, 0.999, 1e-8, i + 1)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
R = 20
for i in range(R): H.adam_step(p, g, m, v, lr, 0.5, 0.999, 1e-8, i + 4)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / R * 1e-3
print("%.1f us per update, %.2f TB/s" % (t * 1e6, 28.0 * n / t / 1e12))
