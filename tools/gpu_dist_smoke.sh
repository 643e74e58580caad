#!/bin/bash
# Exercise the torch.distributed.run launch line and the RCCL call path on a 1-GPU box.
mkdir -p gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/dist_n1.log 2>&1; echo "rc=$?" >> gpurun_out/dist_n1.log
SDPA_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --q-batch 8192 > gpurun_out/dist_forced.log 2>&1; echo "rc=$?" >> gpurun_out/dist_forced.log
tail -2 gpurun_out/dist_n1.log | cut -c1-400; tail -4 gpurun_out/dist_forced.log | cut -c1-600
