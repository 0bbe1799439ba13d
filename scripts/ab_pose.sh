python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -3
for p in map keypoints map keypoints; do python bench.py --steps 30 --warmup 5 --no-info-lines --no-roofline --no-cpu-baseline --pose $p 2>&1 | tail -1 | cut -c60-150; done
for p in map keypoints; do python bench.py --workload df256 --dtype bf16 --steps 20 --warmup 3 --no-info-lines --no-roofline --pose $p 2>&1 | tail -1 | cut -c90-190; done
