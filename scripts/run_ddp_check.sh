python -m pytest tests/test_ddp_gpu.py -x -q -s 2>&1 | grep -v Warning | tail -15
export DPIG_DIST_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 2>&1 | tail -4
