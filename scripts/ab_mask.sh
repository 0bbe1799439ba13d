for m in 0 1 0 1; do DPIG_FUSE_INPUT_MASK=$m python bench.py --steps 30 --warmup 5 --no-info-lines --no-roofline --no-cpu-baseline 2>&1 | tail -1 | cut -c60-150; done
for m in 0 1 0 1; do DPIG_FUSE_INPUT_MASK=$m python bench.py --workload df256 --dtype bf16 --steps 20 --warmup 3 --no-info-lines --no-roofline 2>&1 | tail -1 | cut -c90-190; done
