"""bench.py contract on the GPU box: the N=1 line carries roofline + config; the N=2 launch (the driver's
`torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) completes and reports the whole-job rate.  The two ranks
share the box's one GPU and talk over gloo here (DPIG_DIST_BACKEND); production uses RCCL (backend "nccl")."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_single_gpu_line(dev):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["unit"] == "images/sec" and j["scaling"] == "weak"
    assert j["dtype"] == "f32" and "workload" in j["config"] and j["value"] > 0
    rf = j["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # the headline runs the 3x3 stride-1 convs by Winograd: FLOPs are the EXECUTED ones, the direct-equivalent rate is carried beside them
    assert "winograd" in j["conv_algorithm"] and "EXECUTED" in rf["flop_basis"]
    assert abs(rf["frac_direct_equivalent"] - rf["frac"] * 36.0 / 16.0) < 2e-3
    assert rf["per_kernel_class"]["conv_fwd_wino"]["launches_per_step"] > 0 and rf["per_kernel_class"]["conv_wgrad_wino"]["launches_per_step"] > 0
    assert rf["traffic"] is None or rf["traffic_over_algorithmic"] > 0.5


def test_bench_two_ranks(dev):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DPIG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                        "--steps", "1", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 32 and j["config"]["parallelism"] == "dp2"
    assert j["roofline"] is not None and j["cpu_baseline"] is None
