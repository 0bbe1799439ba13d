python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -3
bash scripts/collect_stats.sh r03d_market_f32 >/dev/null 2>&1
grep -E "pose_stem|thin_generic|fewc|class_sum|colsum|emb_class|generic" profiles/r03d_market_f32_kernel_stats.csv | awk -F'",' '{print substr($1,1,90), $2}' | head -20
tail -n 2 profiles/r03d_market_f32_kernel_stats.md
