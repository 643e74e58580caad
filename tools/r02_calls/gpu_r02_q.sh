#!/bin/bash
# round 2, GPU call Q: bf16 duo + wide kernels without row-max tracking (zero reference exponent,
# packed row sums, scalar end-of-shard fix of the K pieces): bf16 suite, head-dim timings,
# cost of single VALU ops beside bf16 MFMAs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_host_pipeline.py tests/test_gpu_baseline_configs.py -q -x 2>&1 | tail -15 | cut -c1-300 > $O/pytest.log
cat $O/pytest.log
timeout 300 python tools/gpu_bf16_bench.py 512 256 128 64 > $O/bf16_dims.log 2>&1
cat $O/bf16_dims.log | cut -c1-200
(cd tools/probes && timeout 120 ./mfma_probe 1 3 > $O/valu_op_costs.log 2>&1)
cat $O/valu_op_costs.log
