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
    # the driver's command shape (information lines off to keep the test short; test_bench_headline_line_is_compact covers them)
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-info-lines"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.rstrip("\n").splitlines()[-1]
    # round 5's 36-KB line left the driver's record unparsed: the LAST stdout line is the compact headline object
    assert last.startswith("{") and len(last) < 8192, len(last)
    j = json.loads(last)
    assert "roofline" in j and "cpu_baseline" in j and j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["kind"] == "port"
    assert j["compute_mode"] == "f32w"
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["unit"] == "images/sec" and j["scaling"] == "weak"
    assert j["dtype"] == "f32" and "workload" in j["config"] and j["value"] > 0
    rf = j["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # the headline runs the 3x3 stride-1 convs by Winograd: FLOPs are the EXECUTED ones, the direct-equivalent rate is carried beside them
    assert "winograd" in j["conv_algorithm"] and "executed" in rf["flop_basis"]
    deq = 4.0 if "36/144" in rf["flop_basis"] else 36.0 / 16.0      # dominant kernel: F(4x4,3x3) or F(2x2,3x3), whichever holds more of the step
    assert abs(rf["frac_direct_equivalent"] - rf["frac"] * deq) < 2e-3
    pk = rf["per_kernel_class"]
    assert pk["conv_fwd_wino4"][0] > 0 and pk["conv_dgrad_wino4"][0] > 0           # the large maps run the F(4x4,3x3) kernel ...
    assert pk["conv_fwd_wino"][0] > 0 and pk["conv_wgrad_wino"][0] > 0            # ... the small ones F(2x2,3x3), filter gradients F(3x3,2x2)
    assert rf["traffic"] is None or rf["traffic_over_algorithmic"] > 0.5
    assert rf["traffic"] is None or rf["traffic_stale"] is False      # a figure from other kernel sources is reported as null
    assert 0 < rf["step_frac"] < 1


def test_bench_headline_line_is_compact(dev):
    """The information lines go to gpurun_out/bench_info.jsonl + stderr; stdout carries exactly one JSON line with their summary."""
    keys = {"market128_bf16", "market128_bs2_f32"}
    env = dict(os.environ, DPIG_BENCH_INFO_ONLY=",".join(sorted(keys)), DPIG_BENCH_INFO_FILE=os.path.join("gpurun_out", "bench_info_test.jsonl"))
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8192 and r.stdout.rstrip("\n").splitlines()[-1] == lines[0]
    j = json.loads(lines[0])
    assert set(j["info"]) == keys and all(v and v[0] > 0 for v in j["info"].values()), j["info"]
    recs = [json.loads(l) for l in open(os.path.join(ROOT, j["info_file"]))]
    assert {r_["key"] for r_ in recs} == keys and all("roofline" in r_ and "informs" in r_ for r_ in recs)


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
    assert j["rccl_ranks_seen"] == 2 and j["dist_backend"] == "gloo"
