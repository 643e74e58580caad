#!/bin/bash
# round 2, GPU call I: Q-prescaled duo kernel + declared M0 clobbers -- parity, then A/B against the
# variants without them (bf16 head dims, config 5, and the fp32 headline kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
export TMPDIR=/tmp
V=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/lib/variants
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_host_pipeline.py -q 2>&1 | tail -6 > $O/pytest.log
for rep in 1 2; do
for tag in default noprescale nom0bf16; do
  if [ $tag = default ]; then unset SDPA_HIP_LIB; else export SDPA_HIP_LIB=$V/libsdpa_hip_$tag.so; fi
  echo "== $tag" >> $O/bf16.log
  timeout 200 python tools/gpu_bf16_bench.py 512 256 128 64 2>&1 | grep shape | head -4 | cut -c1-110 >> $O/bf16.log
done
for tag in default nom0f32; do
  if [ $tag = default ]; then unset SDPA_HIP_LIB; else export SDPA_HIP_LIB=$V/libsdpa_hip_$tag.so; fi
  echo "== $tag" >> $O/f32.log
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-boundary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms_per_step %.4f kernel_ms_avg %.4f TF %.2f' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['achieved']))" >> $O/f32.log
done; done
unset SDPA_HIP_LIB
cat $O/pytest.log | cut -c1-200; cat $O/bf16.log $O/f32.log
