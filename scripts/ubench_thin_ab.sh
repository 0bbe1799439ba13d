python -m pytest tests/test_conv_bf16_gpu.py -x -q -k "thin" 2>&1 | tail -3
python scripts/ubench_thin.py df 2>&1 | grep -E "fewc fwd|thin3"
DPIG_THIN_MFMA=0 python scripts/ubench_thin.py df 2>&1 | grep -E "fewc fwd"
