"""Sanity soak: N stage-I training steps on fresh synthetic batches; prints the losses every 10 steps and checks that
nothing diverges (finite losses, weights moving, L1 falling from its start as G learns the synthetic statistics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dpig_amd import synthetic
from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0"); np.random.seed(0)
tr = DPIG_Encoder_GAN_BodyROI_FgBg(Config(batch_size=16, compute_dtype=dtype, g_lr=2e-4, d_lr=2e-4), dev)
mk = lambda s: synthetic.to_device(synthetic.make_batch(16, seed=s), dev)
tr.init_net(mk(0))
w0 = tr.G_flat.flat.clone()
hist = []
for it in range(steps):
    out = tr.train_step(mk(2 * it + 1), mk(2 * it + 2))
    if it % 10 == 0 or it == steps - 1:
        vals = {k: float(v) for k, v in out.items() if hasattr(v, "numel") and v.numel() == 1}
        hist.append(vals)
        print("step %3d " % it + " ".join("%s=%.4f" % kv for kv in sorted(vals.items())), flush=True)
torch.cuda.synchronize()
assert all(np.isfinite(list(h.values())).all() for h in hist)
print("weights moved by (max abs) %.3e; L1 first -> last: %.4f -> %.4f" % (float((tr.G_flat.flat - w0).abs().max()),
      hist[1].get("L1Loss", float("nan")), hist[-1].get("L1Loss", float("nan"))))
