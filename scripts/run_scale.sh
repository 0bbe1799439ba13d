#!/bin/bash
# The 8-GPU day in one command: the driver's own launch line at N = 1, 2, 4, 8 on ONE node, then the sweeps that
# DESIGN section 6 could only size from byte counts.  NOTHING here has been executed on more than one device: every box this
# repository has seen had one MI355X (SCALE_r01..r04.json: skipped).  Output: gpurun_out/scale/*.jsonl (one bench line per run).
#
#   bash scripts/run_scale.sh            # curve + sweeps (~15 min on an 8-GPU node)
#   bash scripts/run_scale.sh curve      # the 1/2/4/8 curve only (fp32 headline + Market bf16)
#   bash scripts/run_scale.sh --dry      # rehearsal on ONE GPU: every branch of this script with N clamped to 2 ranks sharing the device
#                                        # over gloo, 2 steps each -- so that a typo does not burn the node's minutes
#
# What to read in each line: value (whole-job img/s), ms_per_step, allreduce_ms (the step's gradient slices all-reduced back to back on
# an idle GPU), exposed_ms (timed step minus the same step with the exchange off = what the staged backward did not hide), comm{}.
# Bars stated in DESIGN section 6: fp32 weak-scaling efficiency >= 0.75 at N = 8 (expected 0.91 .. 0.97), exposed_ms <= 8.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
DRY=0; if [ "${1:-}" = "--dry" ]; then DRY=1; shift; fi
OUT=gpurun_out/scale; [ $DRY -eq 1 ] && OUT=gpurun_out/scale_dry; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0            # dmabuf IPC: without it RCCL fails with hipIpcGetMemHandle: invalid argument on this driver
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
STEPS=${STEPS:-30}; WARM=${WARM:-5}; PORT=29540
if [ $DRY -eq 1 ]; then STEPS=2; WARM=1; export DPIG_DIST_BACKEND=gloo; fi     # (gloo: the ranks share the one device; RCCL needs one device per rank)
run() {   # run <tag> <ngpus> [env assignments ...] -- [bench args ...]
  local tag=$1 n=$2; shift 2
  local envs=(); while [ "$#" -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$#" -gt 0 ] && shift
  local want=$n
  if [ $DRY -eq 1 ] && [ "$n" -gt 2 ]; then n=2; fi
  if [ $DRY -eq 0 ] && [ "$n" -gt "$NGPU" ]; then echo "skip $tag: needs $n GPUs, node has $NGPU"; return; fi
  PORT=$((PORT + 1))
  echo "== $tag: N=$n ${envs[*]:-} $*"
  if [ "$n" -eq 1 ]; then
    env "${envs[@]}" timeout 900 python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-info-lines --no-cpu-baseline "$@" 2> $OUT/$tag.err | grep '^{' | tail -1 > $OUT/$tag.json
  else
    env "${envs[@]}" timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $n --steps $STEPS --warmup $WARM --no-info-lines "$@" 2> $OUT/$tag.err | grep '^{' | tail -1 > $OUT/$tag.json
  fi
  python - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    seen = d.get("rccl_ranks_seen")
    # the collective library's own count of the ranks (a device all-reduce of one-hot rows, bench.py) must be the N that was launched
    flag = "" if (d["n_gpus"] == 1 or seen == d["n_gpus"]) else "  **RANKS SEEN %s != n_gpus %s**" % (seen, d["n_gpus"])
    print("   %s: N=%d (%s) %.1f img/s, %.2f ms/step, allreduce %s ms, exposed %s ms%s" % (
        sys.argv[2], d["n_gpus"], d.get("dist_backend", "-"), d["value"], d["ms_per_step"], d.get("allreduce_ms"), d.get("exposed_ms"), flag))
except Exception as e:
    print("   %s: NO RESULT (%s) -- see the .err file" % (sys.argv[2], e))
PY
}
# ---- 1. the curve: fp32 headline (configs[1], weak scaling, bs = 16 per GPU) and the same graph in bf16 (bf16 wire) ----
for n in 1 2 4 8; do
  run f32_n$n $n --
  run bf16_n$n $n -- --dtype bf16
done
# ---- 1b. SURVEY 8(d)'s strong-scaling information points: the reference's global batch 16 over the node = 2 images per GPU at N = 8
#          (one-GPU twin: bench.py's market128_bs2_f32 information line), and configs[4]'s own per-GPU batch of 4 ----
for n in 1 8; do
  run f32_strong_bs2_n$n $n -- --batch 2
done
[ "${1:-all}" = "curve" ] && exit 0
# ---- 2. configs[2] / configs[4] at their OWN per-GPU batches: stage-II global 64 = 8 per GPU, DeepFashion wgan-gp global 32 = 4 per GPU ----
run stage2_bf16_n8 8 -- --workload market128-stage2 --dtype bf16 --batch 8
run df256_wgan_gp_bf16_n8 8 -- --workload df256-wgan-gp --dtype bf16 --batch 4 --steps 10 --warmup 2
run df256_wgan_gp_bf16_bs4_n1 1 -- --workload df256-wgan-gp --dtype bf16 --batch 4 --steps 10 --warmup 2
# ---- 3. RCCL sweeps at N = 8 (RCCL honours the NCCL_* names).  xGMI is a full mesh of point-to-point links: Ring is the expected
#         winner at 32-MB buckets; the channel count trades CUs taken from the backward's kernels against exchange bandwidth ----
for algo in Ring Tree; do
  run bf16_n8_algo$algo 8 NCCL_ALGO=$algo -- --dtype bf16
done
for ch in 4 8 16 32; do
  run bf16_n8_ch$ch 8 NCCL_MIN_NCHANNELS=$ch NCCL_MAX_NCHANNELS=$ch -- --dtype bf16
  run f32_n8_ch$ch 8 NCCL_MIN_NCHANNELS=$ch NCCL_MAX_NCHANNELS=$ch --
done
run bf16_n8_nomsccl 8 RCCL_MSCCL_ENABLE=0 -- --dtype bf16
# ---- 4. wire format and the staged backward, A/B at N = 8 ----
run bf16_n8_wire_f32 8 DPIG_GRAD_EXCHANGE=f32 -- --dtype bf16
run f32_n8_wire_bf16 8 DPIG_GRAD_EXCHANGE=bf16 --
run bf16_n8_nosplit 8 DPIG_SPLIT_BACKWARD=0 -- --dtype bf16
run f32_n8_nosplit 8 DPIG_SPLIT_BACKWARD=0 --
# ---- 5. cross-rank batch-norm statistics (eager islands between captured graphs) at N = 2 and 8 ----
run f32_n2_syncbn 2 DPIG_SYNC_BN=1 --
run f32_n8_syncbn 8 DPIG_SYNC_BN=1 --
python - "$OUT" <<'PY'
import glob, json, os, sys
rows = {}
for p in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        rows[os.path.basename(p)[:-5]] = json.load(open(p))
    except Exception:
        pass
for fam in ("f32", "bf16"):
    base = rows.get("%s_n1" % fam)
    if base:
        print("weak-scaling efficiency (%s): " % fam + ", ".join(
            "N=%d %.3f" % (rows["%s_n%d" % (fam, n)]["n_gpus"], rows["%s_n%d" % (fam, n)]["value"] / (rows["%s_n%d" % (fam, n)]["n_gpus"] * base["value"]))
            for n in (2, 4, 8) if "%s_n%d" % (fam, n) in rows))
bad = [k for k, d in rows.items() if d.get("n_gpus", 1) > 1 and d.get("rccl_ranks_seen") != d.get("n_gpus")]
print("runs: %d with a result%s" % (len(rows), ("; RANK-COUNT MISMATCH in " + ", ".join(bad)) if bad else "; every multi-rank run saw all its ranks"))
PY
