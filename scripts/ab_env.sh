#!/bin/bash
# Same-box interleaved A/B of one environment switch:  ab_env.sh VAR "v1 v2 ..." ROUNDS bench.py-arguments...
# prints img/s and ms/step of every run (bench.py without roofline / info lines / CPU leg)
VAR=$1; VALS=$2; ROUNDS=$3; shift 3
for r in $(seq 1 $ROUNDS); do for v in $VALS; do
  out=$(env $VAR=$v timeout 300 python bench.py "$@" --no-roofline --no-info-lines --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$VAR=$v $* : $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms")' 2>/dev/null || echo FAILED)"
done; done
