mkdir -p gpurun_out
{ rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8; python -c "import torch;print(torch.cuda.device_count())"; nproc; grep -m1 "model name" /proc/cpuinfo; ls /root/reference 2>&1 | head -2; ls /opt/conda/bin/mpiexec; } > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest1.log
tail -30 gpurun_out/pytest1.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench1.log
tail -5 gpurun_out/bench1.log
