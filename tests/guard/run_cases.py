"""Guard-page harness (TEST TOOL): every device allocation of this process is its own virtual-memory reservation with unmapped pages
on both sides (tests/guard/guard_alloc.cpp), so an out-of-bounds access of ANY kernel operand -- input, output, workspace, flat
parameter / gradient / Adam buffer, channel slice -- is a deterministic `Memory access fault by GPU` in the case that made it.

    AMD_SERIALIZE_KERNEL=3 python tests/guard/run_cases.py hi|lo [first_case [last_case]]

Prints `CASE <i> <name>` before and `OK <i> <name>` after each case (device synchronised), `DONE <n>` at the end; faulthandler is on, so
a fault also prints the Python stack of the launching call.  tests/test_guard_gpu.py runs this in a sub-process for both placements and
restarts after a faulting case so that ONE run lists every faulting case.
"""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
faulthandler.enable(all_threads=True)


def model_case(trainer_mod, cls, steps=2, batch=2, prep=None, wino4=None, **cfg):
    def run(dev):
        import importlib
        import numpy as np
        import dpig_amd.hip_ops as H
        import dpig_amd.tflib as lib
        from dpig_amd import slim, synthetic
        from dpig_amd.trainer import Config
        lib.delete_all_params(); slim.reset_scopes()
        np.random.seed(0)
        prev = H.set_wino_mode(2)
        prev4 = H.set_wino4_mode(wino4) if wino4 is not None else H.get_wino4_mode()     # (2: the F(4x4,3x3) kernel wherever the layer has the form)
        try:
            c = Config(batch_size=batch, **cfg)
            tr = getattr(importlib.import_module("dpig_amd." + trainer_mod), cls)(c, dev)
            bg = synthetic.to_device(synthetic.make_batch(batch, img_H=c.img_H, img_W=c.img_W, seed=21), dev)
            bd = synthetic.to_device(synthetic.make_batch(batch, img_H=c.img_H, img_W=c.img_W, seed=22), dev)
            if prep:
                bg, bd = prep(bg), prep(bd)
            tr.init_net(bg)
            tr.step = 1
            for _ in range(steps):
                out = tr.train_step(bg) if trainer_mod == "trainer_stage2" else tr.train_step(bg, bd)
            vals = [float(v) for v in out.values() if hasattr(v, "numel") and v.numel() == 1]
            assert all(v == v for v in vals), out          # (NaN check: with DPIG_GUARD_FILL=255 fresh memory is NaN)
        finally:
            H.set_wino_mode(prev)
            H.set_wino4_mode(prev4)
            H.set_compute("f32")
            lib.delete_all_params(); slim.reset_scopes()
    return run


def conv_case(N, Hh, W, C, K, R, stride, mode="f32", up=False, slack=0):
    """forward + dgrad + wgrad of one layer through dpig_amd.autograd (what the models call).  slack > 0: x, y are channel slices
    [slack:] of wider buffers, so the LAST pixel's channels end on the buffer's last byte and ld > C."""
    def run(dev):
        import torch
        import dpig_amd.hip_ops as H
        from dpig_amd import autograd as A
        H.set_compute("f32w" if mode == "f32w4" else mode)
        prev = H.set_wino_mode(2)
        prev4 = H.set_wino4_mode(2 if mode == "f32w4" else 0)       # "f32w4": F(4x4,3x3); "f32w": F(2x2,3x3)
        try:
            g = torch.Generator().manual_seed(N * 1000 + C * 7 + K)
            dt = torch.bfloat16 if (mode == "bf16" and C % 8 == 0) else torch.float32
            xb = (torch.rand((N, Hh, W, C + slack), generator=g) - 0.5).to(dev).to(dt)
            x = xb[..., slack:].requires_grad_(True) if slack else xb.requires_grad_(True)
            w = ((torch.rand((R, R, C, K), generator=g) - 0.5) * 0.1).to(dev).requires_grad_(True)
            b = (torch.rand((K,), generator=g) - 0.5).to(dev).requires_grad_(True)
            y = A.conv2d(x, w, b, stride=stride, act=H.ACT_RELU, upsample2x=up) if up else A.conv2d(x, w, b, stride=stride, act=H.ACT_RELU)
            y.float().sum().backward()
            torch.cuda.synchronize()
            assert torch.isfinite(y.float()).all() and torch.isfinite(w.grad).all() and torch.isfinite(x.grad.float()).all()
        finally:
            H.set_wino_mode(prev)
            H.set_wino4_mode(prev4)
            H.set_compute("f32")
    return run


def keypoints(b):
    from dpig_amd import synthetic
    return synthetic.keypoints_only(b)


def build_cases():
    cases = []
    M = "DPIG_Encoder_GAN_BodyROI_FgBg"
    # ---- whole optimizer steps, eager: every kernel the trainers launch, on the widths the suite uses ------------------------------
    cases.append(("step market w16 f32 dcgan", model_case("trainer", M, conv_hidden_num=16, z_num=8)))
    cases.append(("step market w16 f32 dcgan keypoint-fed", model_case("trainer", M, prep=keypoints, conv_hidden_num=16, z_num=8)))
    cases.append(("step market w16 f32 wgan-gp", model_case("trainer", M, conv_hidden_num=16, z_num=8, gan_mode="wgan-gp")))
    cases.append(("step market w16 f32 wgan", model_case("trainer", M, conv_hidden_num=16, z_num=8, gan_mode="wgan")))
    cases.append(("step market w16 bf16", model_case("trainer", M, conv_hidden_num=16, z_num=8, compute_dtype="bf16")))
    cases.append(("step market w16 bf16 wgan-gp", model_case("trainer", M, conv_hidden_num=16, z_num=8, compute_dtype="bf16", gan_mode="wgan-gp")))
    cases.append(("step market w16 bf16x3", model_case("trainer", M, conv_hidden_num=16, z_num=8, compute_dtype="bf16x3")))
    cases.append(("step market w64 f32w", model_case("trainer", M, conv_hidden_num=64, z_num=16, compute_dtype="f32w")))
    cases.append(("step market w64 f32w F(4x4)", model_case("trainer", M, wino4=2, conv_hidden_num=64, z_num=16, compute_dtype="f32w")))
    cases.append(("step market w64 bf16", model_case("trainer", M, conv_hidden_num=64, z_num=16, compute_dtype="bf16")))
    cases.append(("step market w24 f32 B=1", model_case("trainer", M, batch=1, conv_hidden_num=24, z_num=8)))
    cases.append(("step market w16 f32 B=3", model_case("trainer", M, batch=3, conv_hidden_num=16, z_num=8)))
    cases.append(("step stage2 w16 f32", model_case("trainer_stage2", "DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI", conv_hidden_num=16)))
    cases.append(("step stage2 w16 bf16", model_case("trainer_stage2", "DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI", conv_hidden_num=16, compute_dtype="bf16")))
    cases.append(("step df256 w16 f32", model_case("trainer_256", "DPIG_Encoder_GAN_BodyROI_256", batch=1, img_H=256, img_W=256, conv_hidden_num=16, z_num=8)))
    cases.append(("step df256 w16 bf16 wgan-gp", model_case("trainer_256", "DPIG_Encoder_GAN_BodyROI_256", batch=1, img_H=256, img_W=256,
                                                          conv_hidden_num=16, z_num=8, compute_dtype="bf16", gan_mode="wgan-gp")))

    def syncbn(dev):           # cross-rank batch-norm statistics path (world of ONE rank over gloo: the same kernels, no peer)
        import socket
        import torch.distributed as dist
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
        try:
            model_case("trainer", M, conv_hidden_num=16, z_num=8, sync_bn=True)(dev)
            model_case("trainer", M, conv_hidden_num=16, z_num=8, sync_bn=True, compute_dtype="bf16")(dev)
        finally:
            dist.destroy_process_group()
    cases.append(("step market w16 sync_bn (world 1)", syncbn))

    # ---- single layers on odd shapes (C = 3, 6, 20, 72; 1x1 .. 3x3 maps; N = 1), plain and as channel slices ------
    for mode in ("f32", "bf16"):
        for (N, Hh, W) in ((1, 1, 1), (1, 2, 2), (2, 3, 3), (1, 5, 7), (2, 8, 4)):
            for (C, K) in ((3, 6), (6, 20), (20, 72), (72, 3), (64, 64)):
                if mode == "bf16" and (C % 8 or K % 8):
                    continue
                for (R, st) in ((1, 1), (3, 1), (3, 2), (5, 2)):
                    for slack in (0, 8):
                        cases.append(("conv %s N%d %dx%d C%d K%d k%d s%d slack%d" % (mode, N, Hh, W, C, K, R, st, slack),
                                      conv_case(N, Hh, W, C, K, R, st, mode, slack=slack)))
    for (N, Hh, W, C, K) in ((1, 2, 2, 64, 64), (2, 4, 2, 64, 128), (1, 6, 6, 128, 64)):       # Winograd forms
        for slack in (0, 64):
            cases.append(("conv f32w N%d %dx%d C%d K%d slack%d" % (N, Hh, W, C, K, slack), conv_case(N, Hh, W, C, K, 3, 1, "f32w", slack=slack)))
    # F(4x4,3x3) forms: one 4 x 8 block, a 2 x 16 block holding eight images, block rows that straddle images, a split plan
    # (+ the 3-wide form with its unused tile slots and partial last block, a partial 4 x 8 block, a 2 x 16 block with one live tile row)
    for (N, Hh, W, C, K) in ((1, 32, 16, 64, 64), (8, 8, 8, 64, 128), (2, 16, 16, 128, 64), (2, 16, 16, 448, 64),
                             (5, 12, 12, 64, 128), (7, 24, 24, 64, 64), (3, 8, 16, 64, 64), (1, 4, 8, 64, 64)):
        for slack in (0, 64):
            cases.append(("conv f32w F(4x4) N%d %dx%d C%d K%d slack%d" % (N, Hh, W, C, K, slack), conv_case(N, Hh, W, C, K, 3, 1, "f32w4", slack=slack)))
    for (N, Hh, W, C, K) in ((1, 1, 1, 20, 6), (2, 3, 3, 72, 24), (1, 4, 2, 64, 16)):           # upsample-fused 1x1 / 3x3
        cases.append(("conv f32 up2x N%d %dx%d C%d K%d" % (N, Hh, W, C, K), conv_case(N, Hh, W, C, K, 1, 1, "f32", up=True)))
    return cases


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "hi"
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    last = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
    cases = build_cases()
    if mode == "list":
        for i, (n, _) in enumerate(cases):
            print(i, n)
        return
    import conftest
    conftest.install_guard_allocator(mode)
    import torch
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    print("GUARD mode=%s cases=%d" % (mode, len(cases)), flush=True)
    for i, (name, fn) in enumerate(cases):
        if i < first or i > last:
            continue
        print("CASE %d %s" % (i, name), flush=True)
        try:
            fn(dev)
            torch.cuda.synchronize()
            print("OK %d %s" % (i, name), flush=True)
        except Exception as e:        # a Python-level failure is reported and the sweep goes on; a GPU fault kills the process
            import traceback
            traceback.print_exc()
            print("ERR %d %s: %s: %s" % (i, name, type(e).__name__, str(e)[:300]), flush=True)
    print("DONE %d" % len(cases), flush=True)


if __name__ == "__main__":
    main()
