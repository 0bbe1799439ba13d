"""DevicePrefetcher (dpig_amd/prefetch.py): batches arrive in order and intact, also when the consumer lags far behind
the uploader (slot recycling must wait for the consumer's reads), and a trainer step fed through it equals a step fed
with the resident batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host_batches(n, seed=0):
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        out.append({"x": torch.from_numpy(rng.randn(4, 33, 17, 3).astype(np.float32)).pin_memory(),
                    "ids": torch.from_numpy(rng.randint(0, 1 << 30, size=(4, 7)).astype(np.int32)).pin_memory()})
    return out


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("depth", [1, 2, 3])
def test_order_values_and_exhaustion(depth, packed):
    from dpig_amd.prefetch import DevicePrefetcher
    host = _host_batches(7)
    got = []
    for b in DevicePrefetcher(host, "cuda:0", depth=depth, packed=packed):
        assert b["x"].is_cuda and b["ids"].dtype == torch.int32
        got.append({k: v.clone() for k, v in b.items()})
    assert len(got) == len(host)
    for g, h in zip(got, host):
        for k in h:
            assert torch.equal(g[k].cpu(), h[k])


@pytest.mark.parametrize("packed", [False, True])
def test_slow_consumer_never_sees_a_recycled_slot(packed):
    from dpig_amd.prefetch import DevicePrefetcher
    host = _host_batches(12, seed=3)
    big = torch.randn(4096, 4096, device="cuda:0")
    got = []
    for b in DevicePrefetcher(host, "cuda:0", depth=1, packed=packed):
        for _ in range(6):                      # keep the compute stream busy well past the next uploads
            big = torch.tanh(big @ big * 1e-2)
        got.append(b["x"] + 0)                  # read the slot late on the compute stream
    torch.cuda.synchronize()
    for g, h in zip(got, host):
        assert torch.equal(g.cpu(), h["x"])


def test_trainer_step_through_prefetcher_matches_resident_batch():
    from dpig_amd import slim, synthetic, tflib
    from dpig_amd.prefetch import DevicePrefetcher
    from dpig_amd.trainer import Config, DPIG_Encoder_GAN_BodyROI_FgBg
    dev = torch.device("cuda:0")
    cfg = Config(batch_size=2, conv_hidden_num=16, z_num=8)
    losses = []
    for fed in (False, True):
        tflib.delete_all_params()
        slim.reset_scopes()
        np.random.seed(0)
        tr = DPIG_Encoder_GAN_BodyROI_FgBg(cfg, dev)
        bg = synthetic.to_device(synthetic.make_batch(2, seed=5), dev)
        bd = synthetic.to_device(synthetic.make_batch(2, seed=6), dev)
        tr.init_net(bg)
        tr.step = 1
        if fed:
            hg = {k: v.cpu().pin_memory() for k, v in bg.items()}
            hd = {k: v.cpu().pin_memory() for k, v in bd.items()}
            fg, fd = DevicePrefetcher([hg, hg], dev), DevicePrefetcher([hd, hd], dev, packed=True)
            outs = [tr.train_step(next(fg), next(fd)) for _ in range(2)]
        else:
            outs = [tr.train_step(bg, bd) for _ in range(2)]
        losses.append([float(o["g_loss"]) for o in outs] + [float(o["d_loss"]) for o in outs])
    tflib.delete_all_params()
    assert np.allclose(losses[0], losses[1], rtol=1e-5, atol=0), losses
