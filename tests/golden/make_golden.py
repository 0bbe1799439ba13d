"""Generate tests/golden/stage1_market_b2.npz -- golden vectors of the FULL-WIDTH Market stage-I
model (conv_hidden_num=128, z_num=64, 128x64, bs=2) from the fp64 CPU oracle.

The reference (python2 / TF-1.4) cannot run here, so these are oracle outputs, not reference
outputs ("parity unpinned", oracle/__init__.py).  Inputs and weights are NOT stored: both are
regenerated bit-identically from seeds (dpig_amd.synthetic.make_batch(seed) and
oracle.models.ParamStore(seed), numpy Generator streams), only expected outputs are committed.

    python tests/golden/make_golden.py            (~5 min on 8 cores)
    python tests/golden/make_golden.py --small    the width-16 model (seconds): the fixture the CPU suite re-derives
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from dpig_amd import synthetic  # noqa: E402
from oracle import models as OM  # noqa: E402

BATCH_SEED, PARAM_SEED, READOUT_SEED, B = 21, 22, 23, 2
GRAD_PARAMS = ["Encoder/G_encoder/Conv/weights", "Encoder/G_encoder/Conv_16/weights",
               "Encoder/G_encoder/fully_connected_1/weights", "ID_AE/G/Conv/weights", "ID_AE/G/Conv_15/biases",
               "ID_AE/G/Conv_27/weights", "ID_AE/G/Conv_29/weights", "ID_AE/G/fully_connected/biases"]


def subsample(t, n=4096):
    """Deterministic strided sample of a tensor (flattened)."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].to(torch.float64).numpy().copy()


def small_outputs():
    """Width-16 / z-8 model, same seeds: forward values, losses and two kink-robust gradient vectors.  Shared by the
    generator (--small) and tests/test_oracle.py::test_oracle_reproduces_small_golden."""
    ob = OM.batch_to_torch(synthetic.make_batch(B, seed=BATCH_SEED))
    P = OM.ParamStore(seed=PARAM_SEED)
    embs, G = OM.stage1_forward(P, ob, hidden_num=16, z_num=8)
    d_fake = OM.dcgan_discriminator(P, G, "dcgan")
    d_real = OM.dcgan_discriminator(P, ob["x"], "dcgan")
    g_only, d_loss = OM.gan_loss("dcgan", d_real, d_fake)
    l1 = (G - ob["x"]).abs().mean()
    gen = torch.Generator().manual_seed(READOUT_SEED)
    r = torch.randn(tuple(G.shape), generator=gen, dtype=torch.float64)
    names = ["Encoder/G_encoder/Conv/weights", "ID_AE/G/Conv/weights"]
    grads = torch.autograd.grad((G * r).sum(), [P.p[n] for n in names])
    out = {"embs": embs.detach().numpy(), "G": subsample(G, 8192), "d_real": d_real.detach().numpy(),
           "d_fake": d_fake.detach().numpy(), "g_loss": np.array((g_only + 20.0 * l1).item()),
           "L1Loss": np.array(l1.item()), "d_loss": np.array(d_loss.item())}
    for n, g in zip(names, grads):
        out["grad/" + n] = subsample(g)
    return out


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    t0 = time.time()
    if "--small" in sys.argv:
        out = small_outputs()
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage1_market_b2_w16.npz")
        np.savez_compressed(path, **out)
        print("wrote %s (%.0f KB) in %.1fs" % (path, os.path.getsize(path) / 1024.0, time.time() - t0))
        return
    ob = OM.batch_to_torch(synthetic.make_batch(B, seed=BATCH_SEED))
    P = OM.ParamStore(seed=PARAM_SEED)
    taps = {}
    embs, G = OM.stage1_forward(P, ob, taps=taps)
    print("forward %.1fs" % (time.time() - t0), flush=True)
    d_fake = OM.dcgan_discriminator(P, G, "dcgan")
    d_real = OM.dcgan_discriminator(P, ob["x"], "dcgan")
    g_only, d_loss = OM.gan_loss("dcgan", d_real, d_fake)
    l1 = (G - ob["x"]).abs().mean()
    g_loss = g_only + 20.0 * l1
    # kink-robust gradient vectors: linear read-out <G, r> (see tests/test_model_gpu.py docstring)
    gen = torch.Generator().manual_seed(READOUT_SEED)
    r = torch.randn(tuple(G.shape), generator=gen, dtype=torch.float64)
    grads = torch.autograd.grad((G * r).sum(), [P.p[n] for n in GRAD_PARAMS])
    print("backward %.1fs" % (time.time() - t0), flush=True)
    out = {
        "meta": np.array([BATCH_SEED, PARAM_SEED, READOUT_SEED, B]),
        "embs": embs.detach().numpy(), "G": G.detach().numpy(),
        "d_real": d_real.detach().numpy(), "d_fake": d_fake.detach().numpy(),
        "g_loss": np.array(g_loss.item()), "g_loss_only": np.array(g_only.item()),
        "L1Loss": np.array(l1.item()), "d_loss": np.array(d_loss.item()),
    }
    for k, v in taps.items():
        out["tap/" + k] = subsample(v)
        out["tapstat/" + k] = np.array([v.detach().abs().mean().item(), v.detach().abs().max().item()])
    for n, g in zip(GRAD_PARAMS, grads):
        out["grad/" + n] = subsample(g)
        out["gradstat/" + n] = np.array([g.abs().mean().item(), g.abs().max().item()])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage1_market_b2.npz")
    np.savez_compressed(path, **{k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in out.items()})
    print("wrote %s (%.0f KB) in %.1fs" % (path, os.path.getsize(path) / 1024.0, time.time() - t0))


if __name__ == "__main__":
    main()
