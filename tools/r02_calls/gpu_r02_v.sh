#!/bin/bash
# round 2, GPU call V: progressive pinning of the caller's pageable arrays (boundary timing with
# SDPA_PROGRESSIVE_PIN=1/0 at the metric shape and the other configs), host-pipeline tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02v
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_parity.py -q -x 2>&1 | tail -8 | cut -c1-300 > $O/pytest.log
cat $O/pytest.log
for pp in 1 0 1 0; do
  echo "== SDPA_PROGRESSIVE_PIN=$pp" >> $O/hostlevel.log
  SDPA_PROGRESSIVE_PIN=$pp timeout 300 python tools/gpu_hostlevel.py headline config2 config4 2>&1 | grep shape | cut -c1-420 >> $O/hostlevel.log
done
cat $O/hostlevel.log
