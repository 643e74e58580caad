#!/bin/bash
# round 2, GPU call P: does VALU work overlap MFMAs on one SIMD? (fp32 32x32x2 and bf16 32x32x16,
# one and two waves per SIMD)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02p; mkdir -p $O
cd $R/tools/probes
timeout 120 ./mfma_probe 2 2 > $O/f32_overlap.log 2>&1
timeout 200 ./mfma_probe 2 > $O/bf16_overlap.log 2>&1
cat $O/f32_overlap.log; grep -E "wave|mixed2<op1|mixed2<op2|chain<8>" $O/bf16_overlap.log
