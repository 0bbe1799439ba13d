#!/usr/bin/env python
"""bench.py -- DPIG stage-I training throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

One "step" = one iteration of the reference training loop (trainer.py:336-347) in dcgan mode:
g_optim on one synthetic batch + d_optim on another, Market-1501 128x64, bs=16 per GPU, fp32
(BASELINE configs[1]); inputs are resident in HBM before the timed region.  Weak scaling: the
per-GPU batch is fixed, gradients are all-reduced over RCCL each optimizer call.

Prints ONE compact JSON line (< 4 KB, the last line of stdout) on rank 0 with the driver's contract plus
  roofline     -- the dominant kernel (conv forward implicit GEMM), executed FLOPs per launch over its
                  mean launch duration, HIP events on the launch stream, same steps as the timed ones
  cpu_baseline -- the CPU oracle ("port" of the reference graph, torch-CPU fp32) timed on the host
                  cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: "~2.5 PF dense" (v_mfma_f32_32x32x16_bf16; measured 2495 TF)
ALG_GFLOP_PER_IMG = 549.9         # stage-I Market G+D step, dense reference formulation


# name -> (trainer module, class, Config overrides, default per-GPU batch, description)
WORKLOADS = {
    "market128": ("trainer", "DPIG_Encoder_GAN_BodyROI_FgBg", {}, 16,
                  "Market-1501 128x64 stage-I (Fg/Bg/Pose enc + U-Net decoder + DCGAN D), g_optim + d_optim per step"),
    "market128-wgan-gp": ("trainer", "DPIG_Encoder_GAN_BodyROI_FgBg", {"gan_mode": "wgan-gp"}, 16,
                          "Market-1501 128x64 stage-I with MODE='wgan-gp' (LayerNorm critic, gradient penalty, "
                          "g_optim + 5 critic iterations per step, trainer.py:336-347)"),
    "market128-stage2": ("trainer_stage2", "DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI", {}, 64,
                         "Market-1501 128x64 stage-II embedding GAN (model 3: frozen encoder forward x10 per step + "
                         "Gaussian FC mappers / FC critics, MODE='wgan', 5 critic iterations per side)"),
    "df256": ("trainer_256", "DPIG_Encoder_GAN_BodyROI_256", {"img_H": 256, "img_W": 256}, 8,
              "DeepFashion 256x256 stage-I (trainer_256.py path), g_optim + d_optim per step"),
    "market128-sampling": ("tester", "DPIG_FourNetsFgBg_testOnly", {}, 32,
                           "Market-1501 128x64 re-ID data generation (tester.py:256-417, batch 32 as tester.py:67): pose auto-encoder + "
                           "Fg/Bg encoder + both Gaussian appearance mappers + generator + critic score + SSIM, forward only, "
                           "appearance and pose sampled"),
    "df256-wgan-gp": ("trainer_256", "DPIG_Encoder_GAN_BodyROI_256", {"img_H": 256, "img_W": 256, "gan_mode": "wgan-gp"}, 8,
                      "DeepFashion 256x256 stage-I with MODE='wgan-gp' (LayerNorm critic with 8 logit rows per image, gradient "
                      "penalty, g_optim + 5 critic iterations per step): the per-GPU workload of BASELINE configs[4]"),
}


def cpu_baseline(target_seconds=20.0):
    """Time the oracle's G+D step (torch-CPU, fp32, all host cores) on a bounded sample."""
    import torch
    from dpig_amd import synthetic
    from oracle import models as OM
    import torch.nn.functional as F
    ncpu = os.cpu_count() or 1
    # Pool size = the PHYSICAL cores this process may run on (round 4's probe-picked pool gave 0.37 .. 0.90 img/s from box to box:
    # torch-CPU conv throughput collapses when the pool oversubscribes the cores, and two SMT threads of a core share its FMA units).
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or ncpu
    except Exception:
        phys = max(1, ncpu // 2)
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = ncpu
    cores = max(1, min(phys, avail, 64))          # (beyond 64 threads the bs=4 convs have too few rows per thread to scale)
    torch.set_num_threads(cores)

    P = OM.ParamStore(seed=1, dtype=torch.float32)
    opt = {}

    def one_step(ob):
        # trainer.py:337-345: g_optim (loss, gradients, TF-Adam update) then d_optim (the same for the critic)
        for key, loss_fn, names_fn in (("g", OM.stage1_g_loss, OM.g_var_names), ("d", OM.stage1_d_loss, OM.d_var_names)):
            loss, _ = loss_fn(P, ob)
            names = names_fn(P)
            grads = torch.autograd.grad(loss, [P.p[n] for n in names], allow_unused=True)
            if key not in opt:
                opt[key] = OM.OracleAdam(P, names, 2e-5)
            opt[key].step(dict(zip(names, grads)))

    ob1 = OM.batch_to_torch(synthetic.make_batch(1, seed=7), dtype=torch.float32)
    t0 = time.time()
    one_step(ob1)              # untimed: creates the 122.9 M parameters, pages torch's CPU kernels in
    t0 = time.time()
    one_step(ob1)              # bs=1 probe that sizes the sample
    t1 = time.time() - t0
    B = 4                      # BASELINE configs[0]: the reference's CPU-runnable case is bs=4
    reps = int(max(3, min(8, round(target_seconds / max(4 * t1, 1e-3)))))     # >= 3 timed steps
    ob = OM.batch_to_torch(synthetic.make_batch(B, seed=8), dtype=torch.float32)
    t0 = time.time()
    for _ in range(reps):
        one_step(ob)
    t = time.time() - t0
    # one oracle step = g_loss fwd+bwd + TF-Adam update, d_loss fwd+bwd + TF-Adam update; threads = physical cores available, capped at 64
    return {"value": round(B * reps / t, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d oracle G+D steps (torch-CPU fp32) at bs=%d on %d threads of %d logical CPUs: %.1f s" % (reps, B, cores, ncpu, t)}


INFO_RUNS = [   # (key, BASELINE config it informs, bench.py arguments)
    ("market128_direct_f32", "configs[1] with EVERY conv on the direct implicit-GEMM fp32 kernels (exact fp32 products in the direct summation "
     "order): rounds 1-4's headline, kept as the like-for-like line", ["--workload", "market128", "--dtype", "f32", "--steps", "20", "--warmup", "5"]),
    ("df256_f32", "configs[3]'s graph (DeepFashion 256x256, trainer_256.py path, bs=8) in the headline's arithmetic: fp32 with Winograd where it pays",
     ["--workload", "df256", "--dtype", "f32w", "--steps", "10", "--warmup", "2"]),
    ("df256_bf16", "configs[3]: DeepFashion 256x256 (trainer_256.py path) bs=8 bf16 on 1 MI355X",
     ["--workload", "df256", "--dtype", "bf16", "--steps", "20", "--warmup", "3"]),
    ("market128_stage2_bf16", "configs[2]: Market-1501 stage-II adversarial sampling bs=64 bf16 (this GPU's share of the job)",
     ["--workload", "market128-stage2", "--dtype", "bf16", "--steps", "10", "--warmup", "2"]),
    ("market128_bf16", "configs[1]'s graph in bf16", ["--workload", "market128", "--dtype", "bf16", "--steps", "30", "--warmup", "3"]),
    ("market128_split_bf16", "configs[1] (the headline workload) with the conv products on the bf16 pipe as two-term splits of the fp32 "
     "operands (DPIG_COMPUTE_BF16X3: fp32 tensors, <= 2e-5 max|ref| per kernel); the headline itself stays exact fp32",
     ["--workload", "market128", "--dtype", "bf16x3", "--steps", "30", "--warmup", "3"]),
    ("df256_split_bf16", "configs[3]'s graph (DeepFashion 256x256 bs=8) with fp32 tensors and split-bf16 conv products",
     ["--workload", "df256", "--dtype", "bf16x3", "--steps", "10", "--warmup", "2"]),
    ("market128_wgan_gp_f32", "configs[1] with MODE='wgan-gp' (LayerNorm critic, gradient penalty, 5 critic iterations per step)",
     ["--workload", "market128-wgan-gp", "--dtype", "f32", "--steps", "10", "--warmup", "2"]),
    ("market128_sampling_bf16", "SURVEY 8(f-3): the inference / sampling harness at bf16 (generated images per second)",
     ["--workload", "market128-sampling", "--dtype", "bf16", "--steps", "20", "--warmup", "3"]),
    ("df256_wgan_gp_bf16", "configs[4]: DeepFashion 256x256 bs=8 per GPU, MODE='wgan-gp', bf16 -- one GPU's share of the 8-GPU job",
     ["--workload", "df256-wgan-gp", "--dtype", "bf16", "--steps", "10", "--warmup", "2"]),
    ("df256_wgan_gp_bf16_bs4", "configs[4] at its OWN per-GPU batch: global 32 over 8 GPUs = 4 per GPU (SURVEY 8e) -- half the rows per layer of the "
     "bs=8 line above", ["--workload", "df256-wgan-gp", "--dtype", "bf16", "--batch", "4", "--steps", "10", "--warmup", "2"]),
    ("market128_bs2_f32", "SURVEY 8(d)'s strong-scaling point: global B=16 over 8 GPUs = 2 images per GPU (small-batch CU fill of the headline graph)",
     ["--workload", "market128", "--dtype", "f32w", "--batch", "2", "--steps", "20", "--warmup", "5"]),
    ("market128_host_input_f32", "SURVEY 8(f-1): the headline step with both batches starting every step in pinned HOST memory as the records' "
     "keypoints + image + mask + boxes, packed into one buffer, uploaded one step ahead on a copy stream (prefetch.DevicePrefetcher)",
     ["--workload", "market128", "--host-input", "keypoints-packed", "--steps", "20", "--warmup", "5"]),
    ("market128_tfrecord_f32", "SURVEY 8(f-1): the headline step fed from serialized tf.train.Example records (the reference's schema, "
     "datasets/market1501.py:79-141): tfrecord.RecordFeeder decodes on 4 host threads -> DevicePrefetcher -> step",
     ["--workload", "market128", "--host-input", "tfrecord", "--steps", "20", "--warmup", "5"]),
    ("market128_tfrecord_bf16", "the same record-fed pipeline at bf16 (a 5x shorter step for the host decode to keep up with)",
     ["--workload", "market128", "--dtype", "bf16", "--host-input", "tfrecord", "--steps", "30", "--warmup", "5"]),
]


INFO_FILE = os.environ.get("DPIG_BENCH_INFO_FILE", os.path.join("gpurun_out", "bench_info.jsonl"))


def csrc_sha():
    """sha1 over the kernel sources + the C header: what a profiles/roofline_traffic.json entry was measured on."""
    import glob
    import hashlib
    h = hashlib.sha1()
    pkg = os.path.join(ROOT, "disentangled-person-image-generation_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(pkg, "*.hip")) + glob.glob(os.path.join(pkg, "*.h")) + [os.path.join(ROOT, "include", "dpig_hip.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def info_lines():
    """The other BASELINE configurations that fit one GPU, each measured by this same script in a sub-process (fresh parameter
    registry, same kernels) AFTER the headline's timed region.  Every sub-process's full record goes to INFO_FILE (one JSON object per
    line, with the configuration it informs) and to stderr; the headline object only carries {key: [images/sec, ms/step, roofline.frac]}
    (round 5 nested the full records and the 36-KB line no longer parsed)."""
    import subprocess
    out = {}
    path = os.path.join(ROOT, INFO_FILE)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        fh = open(path, "w")
    except OSError:
        fh = None
    only = [k for k in os.environ.get("DPIG_BENCH_INFO_ONLY", "").split(",") if k]      # (tests: a bounded subset)
    for key, informs, extra in INFO_RUNS:
        if only and key not in only:
            continue
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-info-lines"] + extra,
                               capture_output=True, text=True, timeout=300)
            js = [l for l in r.stdout.splitlines() if l.startswith("{")]
            d = json.loads(js[-1])
            d["key"], d["informs"] = key, informs
            out[key] = [d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac")]
        except Exception as e:          # an information line must never take the headline down
            d = {"key": key, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            out[key] = None
        txt = json.dumps(d)
        print("[info] " + txt, file=sys.stderr, flush=True)
        if fh is not None:
            fh.write(txt + "\n")
            fh.flush()
    if fh is not None:
        fh.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 50 steps after 10 warm-up (3.7 s + 0.7 s on one GPU)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 16 for Market, BASELINE configs[1]; 8 for df256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly (no hipGraph replay)")
    ap.add_argument("--dtype", default="f32w", choices=["f32", "bf16", "bf16c", "bf16x3", "f32w"],
                    help="f32w (default since round 5, the headline) = the BASELINE metric's arithmetic TYPE -- fp32 tensors, fp32 products, fp32 "
                         "accumulation -- with the 3x3 stride-1 convs (forward, dgrad, wgrad) evaluated by Winograd minimal filtering "
                         "F(4x4,3x3) (maps with a 4x4-tile block form: 4x fewer multiplies) / F(2x2,3x3) / F(3x3,2x2) (2.25x fewer) on the fp32 "
                         "matrix pipe where the library's cost model says it pays "
                         "(every golden activation within 1e-4 of max|ref| of the fp64 oracle, tests/test_golden_gpu.py); f32 = the same "
                         "with every conv on the direct implicit-GEMM kernels (rounds 1-4's headline, now the information line "
                         "market128_direct_f32).  bf16 (information lines; BASELINE configs 3-5): activations, "
                         "their gradients and the filter shadows stored as bf16, bf16 matrix pipe, fp32 accumulation / master "
                         "weights / gradients / optimizer.  bf16c: round 1's intermediate mode (fp32 tensors, bf16 pipe).  bf16x3 "
                         "(information line): fp32 tensors, conv operands split into two bf16 terms, three bf16 MFMAs per product "
                         "block -- the exact kernels' own 2e-5 accuracy bar on the bf16 pipe.")
    ap.add_argument("--no-info-lines", action="store_true",
                    help="headline run only: skip the information lines (df256 / stage-II / Market in bf16, Market wgan-gp) that "
                         "are measured in sub-processes after the headline; full records -> gpurun_out/bench_info.jsonl + stderr, summary under `info`")
    ap.add_argument("--host-input", nargs="?", const="prefetch", default=None, choices=["prefetch", "serial", "keypoints", "keypoints-serial", "keypoints-packed", "packed", "tfrecord"],
                    help="information line: both batches start every step in pinned HOST memory, so the timed region "
                         "includes their PCIe upload (the BASELINE value is quoted with inputs resident in HBM). "
                         "'prefetch' uploads on a copy stream one step ahead (dpig_amd.prefetch), 'serial' on the compute stream, "
                         "'keypoints' = prefetch of the 18 (row, col, visibility) triplets instead of the dense pose maps, "
                         "rasterised on the device (what tfrecord.batch_from_examples feeds); 'keypoints-serial' = the same upload on "
                         "the compute stream, no second queue; 'packed' / 'keypoints-packed' = prefetch with the batch packed "
                         "into one pinned buffer and moved by one copy; 'tfrecord' = the batches are DECODED every step from serialized "
                         "tf.train.Example records (tfrecord.RecordFeeder, 4 host threads) and uploaded packed, one step ahead")
    ap.add_argument("--copy-input", action="store_true",
                    help="resident batches are separate device tensors copied into the graphs' static inputs every step (rounds 1-3's "
                         "methodology; the default since round 4 hands the graphs' own input buffers to the step: no per-step copy)")
    ap.add_argument("--pose", default="keypoints", choices=["keypoints", "map"],
                    help="keypoints (default): the resident batch holds pose_rcv, the [B,18,3] keypoints of the records, and the "
                         "generator's first conv consumes them directly (the reference rasterises the target map inside the graph, "
                         "trainer.py:556-560); map: the batch holds the rasterised [B,H,W,18] map (rounds 1-2)")
    ap.add_argument("--workload", default="market128", choices=sorted(WORKLOADS),
                    help="market128 = the BASELINE metric (configs[1]); the others are information lines for DESIGN.md")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ndev = torch.cuda.device_count()
        local_rank = local_rank % max(ndev, 1)            # (only the 1-GPU gloo smoke test shares a device)
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("DPIG_DIST_BACKEND", "nccl")             # "nccl" IS RCCL on ROCm
        dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from dpig_amd import hip_ops as H
    from dpig_amd import synthetic
    import importlib
    from dpig_amd.trainer import Config
    wl_mod, wl_cls, wl_cfg, wl_batch, wl_desc = WORKLOADS[args.workload]
    headline = args.workload == "market128" and args.dtype == "f32w" and not args.host_input
    if not headline:                          # information lines: no CPU leg
        args.no_cpu_baseline = True

    np.random.seed(0)                         # identical initial weights on every rank (+ broadcast)
    B = args.batch or wl_batch
    cfg_kw = dict(wl_cfg)
    # A/B switches of the data-parallel path for scripts/run_scale.sh (defaults: wire format follows the dtype, staged backward
    # in data-parallel runs, per-rank batch-norm statistics)
    if os.environ.get("DPIG_GRAD_EXCHANGE") in ("f32", "bf16"):
        cfg_kw["grad_exchange"] = os.environ["DPIG_GRAD_EXCHANGE"]
    if os.environ.get("DPIG_SPLIT_BACKWARD") in ("0", "1"):
        cfg_kw["split_backward"] = os.environ["DPIG_SPLIT_BACKWARD"] == "1"
    if os.environ.get("DPIG_SYNC_BN") == "1":
        cfg_kw["sync_bn"] = True
    cfg = Config(batch_size=B, compute_dtype=args.dtype, **cfg_kw)
    sampling = args.workload == "market128-sampling"
    if sampling:
        args.no_graph = True
        tr = getattr(importlib.import_module("dpig_amd." + wl_mod), wl_cls)(cfg, dev, sample_app=True, sample_pose=True)
    else:
        tr = getattr(importlib.import_module("dpig_amd." + wl_mod), wl_cls)(cfg, dev)
    batch_g = synthetic.to_device(synthetic.make_batch(B, img_H=cfg.img_H, img_W=cfg.img_W, seed=100 + 2 * rank), dev)
    batch_d = synthetic.to_device(synthetic.make_batch(B, img_H=cfg.img_H, img_W=cfg.img_W, seed=101 + 2 * rank), dev)
    kp_host = bool(args.host_input) and (args.host_input == "tfrecord" or args.host_input.startswith("keypoints"))
    if args.pose == "keypoints" and (not args.host_input or kp_host) and args.workload not in ("market128-stage2", "market128-sampling"):
        batch_g, batch_d = synthetic.keypoints_only(batch_g), synthetic.keypoints_only(batch_d)
    if sampling:
        tr.run(batch_g, batch_g["pose_rcv"])  # builds the graph's variables (random init: there are no checkpoints here)
    else:
        tr.init_net(batch_g)
        tr.step = 1                           # steady state: g_optim is only skipped at step 0
    if not args.no_graph:
        if args.workload == "market128-stage2":
            tr.enable_graphs(batch_g)         # one hipGraph per (side, optimizer op): frozen-encoder forward + mapper / critic update
            if not args.copy_input:
                batch_g = tr.static_batch()   # (the resident batch is the graphs' input buffer)
        else:
            tr.enable_graphs(batch_g, batch_d)    # fwd+bwd+all-reduce+Adam of each optimizer op = one hipGraph
            if not args.host_input and not args.copy_input and hasattr(tr, "static_batches"):
                # the resident batches ARE the graphs' input buffers (what a device-side producer fills): no per-step device copy
                batch_g, batch_d = tr.static_batches()

    if args.host_input == "serial" and args.no_graph:
        raise SystemExit("--host-input needs the hipGraph path (the eager path takes device batches)")
    feed_g = feed_d = None
    feeders = []
    if args.host_input == "tfrecord":
        # serialized records of the reference's schema (datasets/market1501.py:79-141; raw image bytes) built once from synthetic
        # batches; EVERY step decodes 2 x B of them on host threads (tfrecord.RecordFeeder), packs, uploads one step ahead
        from dpig_amd import tfrecord as T
        from dpig_amd.prefetch import DevicePrefetcher

        def records(seed):
            nb = synthetic.make_batch(4 * B, img_H=cfg.img_H, img_W=cfg.img_W, seed=seed)
            out = []
            for i in range(4 * B):
                ex = {"image_format": [b"raw"], "image_height": np.array([cfg.img_H]), "image_width": np.array([cfg.img_W])}
                rcv = np.asarray(nb["pose_rcv"][i])
                for sfx in ("0", "1"):
                    img = np.clip(np.rint(np.asarray(nb["x"][i]) * 127.5 + 127.5), 0, 255).astype(np.uint8)
                    ex.update({"image_raw_" + sfx: [img.tobytes()], "pose_peaks_%s_rcv" % sfx: rcv.reshape(-1).astype(np.float32),
                               "pose_mask_r6_" + sfx: np.asarray(nb["mask_r6"][i]).reshape(-1).astype(np.int64),
                               "part_bbox_" + sfx: np.asarray(nb["part_bbox"][i]).reshape(-1).astype(np.int64),
                               "part_vis_" + sfx: np.asarray(nb["part_vis"][i]).reshape(-1).astype(np.int64)})
                out.append(T.encode_example(ex))
            return out
        for seed in (400 + 2 * rank, 401 + 2 * rank):
            feeders.append(T.RecordFeeder(records(seed), B, which=0, img_H=cfg.img_H, img_W=cfg.img_W, workers=4, depth=4, pin=True))
        feed_g = DevicePrefetcher(feeders[0], dev, packed=True)
        feed_d = DevicePrefetcher(feeders[1], dev, packed=True)
    elif args.host_input:                     # the replayed graphs read their own static buffers; _feed copies into them
        batch_g = {k: v.cpu().pin_memory() for k, v in batch_g.items()}
        batch_d = {k: v.cpu().pin_memory() for k, v in batch_d.items()}
        if args.host_input != "serial":
            import itertools
            from dpig_amd import utils
            from dpig_amd.prefetch import DevicePrefetcher
            if args.host_input.startswith("keypoints"):
                def sparse(b, seed):
                    rs = np.random.RandomState(seed)
                    rcv = np.stack([rs.uniform(0, cfg.img_H, (B, 18)), rs.uniform(0, cfg.img_W, (B, 18)),
                                    (rs.uniform(size=(B, 18)) < 0.9).astype(np.float64)], axis=-1)
                    b = {k: v for k, v in b.items() if k != "pose"}
                    b["pose_rcv"] = torch.from_numpy(rcv.reshape(B, 54).astype(np.float32)).pin_memory()
                    return b

                def rasterised(feed):             # --pose map: the graphs consume the dense target map, rasterised on the device
                    if "pose" not in tr.static_batches()[0]:
                        yield from feed               # (default: the keypoints feed the generator's first conv directly)
                        return
                    for b in feed:
                        b = dict(b)
                        b["pose"] = utils.pose_target_from_rcv(b.pop("pose_rcv"), 18, False, cfg.img_H, cfg.img_W)
                        yield b
                def uploaded(host):
                    while True:
                        yield {k: v.to(dev, non_blocking=True) for k, v in host.items()}
                batch_g, batch_d = sparse(batch_g, 1), sparse(batch_d, 2)
                if args.host_input != "keypoints-serial":
                    pk = args.host_input.endswith("packed")
                    feed_g = rasterised(DevicePrefetcher(itertools.repeat(batch_g), dev, packed=pk))
                    feed_d = rasterised(DevicePrefetcher(itertools.repeat(batch_d), dev, packed=pk))
                else:
                    feed_g, feed_d = rasterised(uploaded(batch_g)), rasterised(uploaded(batch_d))
            else:
                feed_g = DevicePrefetcher(itertools.repeat(batch_g), dev, packed=args.host_input == "packed")
                feed_d = DevicePrefetcher(itertools.repeat(batch_d), dev, packed=args.host_input == "packed")

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    get_g = (lambda: next(feed_g)) if feed_g is not None else (lambda: batch_g)
    get_d = (lambda: next(feed_d)) if feed_d is not None else (lambda: batch_d)
    if sampling:
        step_fn = lambda: tr.run(batch_g, batch_g["pose_rcv"])
    elif args.workload == "market128-stage2":
        step_fn = lambda: tr.train_step(get_g())
    elif args.workload.endswith("wgan-gp") and feed_d is None:
        # every critic iteration of a step dequeues its own batch (trainer.py:340-345, 553-555): 5 distinct resident batches
        critic_batches = [batch_d] + [synthetic.to_device(synthetic.make_batch(B, img_H=cfg.img_H, img_W=cfg.img_W,
                                                                             seed=300 + 10 * rank + i), dev) for i in range(4)]
        if args.pose == "keypoints":
            critic_batches = [synthetic.keypoints_only(b) for b in critic_batches]
        step_fn = lambda: tr.train_step(get_g(), critic_batches)
    else:
        step_fn = lambda: tr.train_step(get_g(), get_d())
    for _ in range(args.warmup):
        step_fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step_fn()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # ---- N > 1: what the gradient exchange costs (rank-max like the headline) ---------------------------------------------
    comm, comm_error = None, None
    if world > 1 and hasattr(tr, "G_flat") and hasattr(tr, "D_flat") and getattr(tr, "allreduce", None) is not None and tr.allreduce.enabled:
        try:
            def rank_max(sec):
                t = torch.tensor([sec], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return t.item()
            # (a) the step's collectives alone, back to back on an otherwise idle GPU: one slice per backward stage of the generator
            #     side (generator, background tower, ROI tower, stem: trainer._backward_stages) + the critic's
            if hasattr(tr, "_stages"):
                slices = [tr._stage_slice(st) for st in tr._stages] + [tr.D_flat.grad]
            else:
                eo = getattr(tr, "_enc_off", 0)
                slices = [tr.G_flat.grad[:eo], tr.G_flat.grad[eo:], tr.D_flat.grad]
            slices = [s for s in slices if s.numel()]
            reps = 10
            for _ in range(2):
                tr.allreduce.finish(sum((tr.allreduce.start(s) for s in slices), []))
            sync()
            t1 = time.perf_counter()
            for _ in range(reps):
                tr.allreduce.finish(sum((tr.allreduce.start(s) for s in slices), []))
            sync()
            ar_ms = rank_max(time.perf_counter() - t1) / reps * 1e3
            # (b) the same K steps with the exchange switched off (every rank keeps its local gradient): the difference to the
            #     timed region is the part of the exchange that the backward pass did not hide
            tr.allreduce.enabled = False
            for _ in range(min(args.warmup, 3)):
                step_fn()
            sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step_fn()
            sync()
            local_ms = rank_max(time.perf_counter() - t1) / args.steps * 1e3
            tr.allreduce.enabled = True
            nbytes = sum(s.numel() for s in slices) * (2 if tr.allreduce.compress else 4)
            comm = {"allreduce_ms": round(ar_ms, 3), "exposed_ms": round(ms_per_step - local_ms, 3),
                    "ms_per_step_without_exchange": round(local_ms, 3), "bytes_per_step": nbytes,
                    "collectives_per_step": sum(-(-s.numel() // tr.allreduce.bucket) for s in slices),
                    "wire_dtype": "bf16" if tr.allreduce.compress else "f32"}
            # allreduce_ms: the step's gradient all-reduces alone on an idle GPU; exposed_ms: timed step minus the same step with the
            # exchange off (what the backward pass did not overlap)
        except Exception as e:      # a failure of this extra measurement must not take the bench line down (it is the same on every rank)
            tr.allreduce.enabled = True
            comm = None
            comm_error = "%s: %s" % (type(e).__name__, str(e)[:200])

    roofline = None
    if not args.no_roofline:
        # instrumented replay of the same step: HIP events around every conv-forward launch.  EVERY rank replays
        # (the steps contain the gradient all-reduce); rank 0's numbers are the ones reported.
        H.PROFILE = []
        nrep = max(1, min(3, args.steps))
        graphs = getattr(tr, "_graphs", None)
        tr._graphs = None                              # eager launches so that each one can be bracketed
        from dpig_amd import autograd as _A
        side = (_A.TWO_STREAM[0], _A.D_OVERLAP[0])
        _A.TWO_STREAM[0] = _A.D_OVERLAP[0] = False     # ... and on ONE stream, so that every launch is timed alone (the timed steps
        try:                                           # above run independent towers / critic passes side by side)
            for _ in range(nrep):
                step_fn()
            torch.cuda.synchronize()
        finally:
            _A.TWO_STREAM[0], _A.D_OVERLAP[0] = side
        tr._graphs = graphs
        recs = [(r[0], r[1], r[2].elapsed_time(r[3]) * 1e-3) for r in H.PROFILE]
        labels = [(r[0], r[4]) for r in H.PROFILE if len(r) > 4]
        H.PROFILE = None
        # dominant kernel class of this configuration: the conv-forward implicit GEMM on the pipe the dtype selects
        dom, peak, kname, fmul = {
            "f32": ("conv_fwd_mfma", PEAK_F32_MFMA_TFLOPS, "dpig::gather_gemm_kernel<false,true,false,0> (conv fwd implicit GEMM, v_mfma_f32_32x32x2_f32)", 1.0),
            "bf16": ("conv_fwd_bf16", PEAK_BF16_MFMA_TFLOPS, "dpig::bfk::{bhq,bhq32,bq,bh,bg8,sk}_kernel (conv fwd implicit GEMM on bf16 tensors, v_mfma_f32_32x32x16_bf16)", 1.0),
            "bf16c": ("conv_fwd_mfma", PEAK_BF16_MFMA_TFLOPS, "conv fwd implicit GEMM, fp32 tensors rounded to bf16 on the way into LDS", 1.0),
            # fp32 tensors as two-term bf16 splits: 3 bf16 MFMAs per product block (executed FLOPs = 3 x algorithmic)
            "bf16x3": ("conv_fwd_mfma", PEAK_BF16_MFMA_TFLOPS, "conv fwd implicit GEMM, two-term bf16 splits of fp32 operands (3 MFMAs per product block)", 3.0),
            # 3x3 stride-1 conv fwd as Winograd F(2x2,3x3): 16 position GEMMs, transforms fused; FLOPs = the executed 16/36 of the direct count
            "f32w": ("conv_fwd_wino", PEAK_F32_MFMA_TFLOPS, "dpig::wino::wino_block_kernel / wino_kernel (Winograd F(2x2,3x3) conv fwd, v_mfma_f32_32x32x2_f32)", 1.0),
        }[args.dtype]
        wino4_dom = False
        if args.dtype == "f32w":
            # two Winograd forward classes: the dominant kernel is the one with the larger share of the step
            t2 = sum(t for (k, f, t) in recs if k == "conv_fwd_wino")
            t4 = sum(t for (k, f, t) in recs if k == "conv_fwd_wino4")
            if t4 > t2:
                wino4_dom = True
                dom = "conv_fwd_wino4"
                kname = "dpig::wino4::wino4_kernel (Winograd F(4x4,3x3) conv fwd, v_mfma_f32_32x32x2_f32)"
        fwd = [(f, t) for (k, f, t) in recs if k == dom]
        nl = max(len(fwd), 1)
        flops = sum(f for f, _ in fwd) * fmul
        secs = max(sum(t for _, t in fwd), 1e-12)
        achieved = flops / secs / 1e12
        # ALGORITHMIC bytes of the same launches: each operand once -- input map + filter + output map at the storage width
        esz = 2 if args.dtype == "bf16" else 4
        def alg_bytes_of(classes):
            a = [esz * (n * h * w * c + (n * 4 * h * w * kk if up else n * (-(-h // st)) * (-(-w // st)) * kk) + r * r * c * kk)
                 for (k, (n, h, w, c, kk, r, st, up)) in labels if k in classes]
            return sum(a) / max(len(a), 1)
        alg_bytes = alg_bytes_of((dom,))
        # HBM bytes per launch of that kernel class from the FETCH_SIZE / WRITE_SIZE passes of THIS command under rocprofv3
        # (scripts/pmc_traffic.sh -> profiles/roofline_traffic.json; separate --pmc passes, FETCH_SIZE doubled per the guide's gfx950 note)
        traffic, traffic_head, traffic_stale, wg_traffic, wg_alg = None, None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                ent = json.load(open(tpath)).get("entries", {}).get("%s/%s" % (args.workload, args.dtype))
                if ent and B == wl_batch:
                    traffic_head = ent.get("head")
                    # a figure measured on other kernel sources than the ones this run executes is not "this run": report null
                    traffic_stale = ent.get("csrc_sha") != csrc_sha()
                    if not traffic_stale:
                        traffic = ent.get("hbm_bytes_per_launch")
                        alg_bytes = alg_bytes_of(tuple(ent.get("classes", [dom])))      # (the bf16 kernels serve forward AND dgrad launches)
                        if ent.get("wgrad"):          # the Winograd filter-gradient class beside it: x + dy + the [3][3][C][K] gradient once
                            wg_traffic = ent["wgrad"].get("hbm_bytes_per_launch")
                            wg_alg = alg_bytes_of(tuple(ent["wgrad"].get("classes", ["conv_wgrad_wino"])))
            except Exception:
                traffic = None
        roofline = {"bound": "mfma", "kernel": kname,
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_head": traffic_head, "traffic_stale": traffic_stale,
                    "algorithmic_bytes_per_launch": int(alg_bytes),
                    "traffic_over_algorithmic": round(traffic / alg_bytes, 2) if (traffic and alg_bytes) else None,
                    "launches_per_step": len(fwd) // nrep,
                    "flops_per_launch": round(flops / nl), "avg_launch_us": round(secs / nl * 1e6, 2),
                    "time_share_of_step": round(secs / nrep / (ms_per_step * 1e-3), 3),
                    "flop_basis": ("executed (36/144 of direct)" if wino4_dom else "executed (16/36 of direct)") if args.dtype == "f32w" else "executed = direct"}
        if wg_traffic:
            roofline["wgrad_traffic"], roofline["wgrad_algorithmic_bytes_per_launch"] = wg_traffic, int(wg_alg)
        if args.dtype == "f32w":
            deq = 4.0 if wino4_dom else 36.0 / 16.0
            roofline["achieved_direct_equivalent"] = round(achieved * deq, 2)
            roofline["frac_direct_equivalent"] = round(achieved * deq / peak, 4)
        by = {}
        for k, f, t in recs:
            a = by.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1; a[1] += f; a[2] += t
        roofline["per_kernel_class_fields"] = "launches/step, ms/step, TFLOP/s"
        roofline["per_kernel_class"] = {k: [v[0] // nrep, round(v[2] / nrep * 1e3, 3), round(v[1] / v[2] / 1e12, 1) if v[2] > 0 and v[1] > 0 else None]
                                        for k, v in sorted(by.items())}
        # whole-step matrix-pipe utilisation on executed FLOPs: every FLOP-carrying launch of the step over the timed step
        tot_flops = sum(v[1] for v in by.values()) * fmul / nrep
        roofline["step_executed_tflop"] = round(tot_flops / 1e12, 3)
        roofline["step_frac"] = round(tot_flops / (ms_per_step * 1e-3) / 1e12 / peak, 4)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    MODE_DESC = {"f32": "fp32 direct", "f32w": "fp32, Winograd on 3x3 s1 where the cost model picks it", "bf16": "bf16 storage + bf16 MFMA, fp32 accumulate/master/optimizer",
                 "bf16c": "fp32 tensors, bf16 MFMA", "bf16x3": "fp32 tensors, split-bf16 products"}[args.dtype]
    rccl_ranks_seen = None
    if world > 1:
        # what the collective library actually connected: every rank contributes one bit, summed by a real all-reduce on the device
        t = torch.zeros(world, dtype=torch.float32, device=dev)
        t[rank] = 1.0
        dist.all_reduce(t)
        rccl_ranks_seen = int((t > 0.5).sum().item())
    if rank == 0:
        line = {
            "metric": "training images/sec (G+D step) Market-1501 128x64 bs=16" if headline else
                      ("generated images/sec (%s, %s%s) [information line]" if sampling else "training images/sec (%s, %s%s) [information line]") % (
                          args.workload, args.dtype, ", host input %s" % args.host_input if args.host_input else ""),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.dtype == "f32w" else args.dtype, "compute_mode": args.dtype, "data": "synthetic",
            "conv_algorithm": "winograd F(4x4,3x3)/F(2x2,3x3)/F(3x3,2x2) on 3x3 s1 + direct implicit GEMM" if args.dtype == "f32w" else "direct implicit GEMM",
            "config": {"workload": "%s bs=%d/GPU (%s; pose as %s)" % (args.workload, B, MODE_DESC, "map" if "pose" in batch_g else "keypoints"),
                       "global_batch": world * B, "parallelism": "dp%d" % world},
            "achieved_alg_tflops": round(ALG_GFLOP_PER_IMG * value / 1e3, 2) if headline else None,
            "losses": {k: round(float(v), 5) for k, v in out.items() if hasattr(v, "numel") and v.numel() == 1 and not sampling},
            "input_feed": ("host:%s" % args.host_input) if args.host_input else ("copy" if (args.copy_input or args.no_graph) else "static"),
            "roofline": roofline, "cpu_baseline": cpu,
            "build_mode": __graft_entry__.BUILD_MODE,      # "compiled": this process rebuilt the library; "reused": the shipped .so was fresh
        }
        if world > 1:
            line["rccl_ranks_seen"] = rccl_ranks_seen
            line["dist_backend"] = dist.get_backend()
        if comm is not None:
            line["allreduce_ms"], line["exposed_ms"], line["comm"] = comm["allreduce_ms"], comm["exposed_ms"], comm
        elif comm_error is not None:
            line["comm"] = {"error": comm_error}
        if headline and world == 1 and not args.no_info_lines:
            line["info_fields"] = "images/sec, ms/step, roofline.frac"
            line["info"] = info_lines()
            line["info_file"] = INFO_FILE
        # ONE compact line, the last of stdout (the driver stores a bounded tail and parses the last line)
        print(json.dumps(line, separators=(",", ":")), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
