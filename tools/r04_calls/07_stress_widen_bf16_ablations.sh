#!/bin/bash
# round 4, call 7: (a) the host-widening tests; (b) can the suite's rare GPU memory fault be provoked by alternating
# PyTorch's pageable copies with the C host's hipHostRegister of heap arrays -- and does it go away when nothing is
# registered; (c) A/B of the result's widening placement at the boundary; (d) timing-only ablations of the bf16 tandem
# kernel's loop (what bounds config 5) and the ragged-mask-in-every-step variant (rounds 1-3); (e) 2 loopback ranks at config 4 with the new
# default reservation: do the merge tails run under the next batch's fused kernels
O=gpurun_out/r04_07; mkdir -p $O
R=$GRAFT_REPO_ROOT
PKG=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_bf16.py -q -x > $O/pytest_widen.log 2>&1; echo "(a) rc=$? $(grep -aE ' passed| failed' $O/pytest_widen.log | tail -1 | cut -c1-100)"
grep -a "Error\|assert\|FAILED" $O/pytest_widen.log | head -10 | cut -c1-300
# (b)
# (both faults of the round came seconds behind two REFUSED registrations of already page-locked caller arrays:
#  refused = that path, blind as in round 3; probed = the shipped default, which asks hipPointerGetAttributes first)
timeout 900 python tools/gpu_register_stress.py 75 refused probed register > $O/register_stress.log 2>&1; cat $O/register_stress.log | cut -c1-600
# (c)
timeout 300 python tools/gpu_hostlevel.py headline config2 --widen > $O/host_widen_ab.log 2>&1
timeout 300 python tools/gpu_hostlevel.py headline config2 --widen --pinned >> $O/host_widen_ab.log 2>&1
grep '^{' $O/host_widen_ab.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['shape'], 'pinned' if j['pinned'] else 'pageable', j['knobs'], 'total', j['total_ms'], 'tail', j['tail_ms'], 'reg', j['register_ms'], 'widen', j['host_widen'])"
# (d)
for v in base maskall tabl1 tabl2 tabl4 tabl8 tabl16 tabl3 tabl5 tabl31 maskall base; do
  echo -n "$v: " >> $O/bf16_tandem_ablations.log
  SDPA_HIP_LIB=$PKG/lib/variants/libsdpa_hip_$v.so timeout 200 python tools/gpu_bf16_bench.py 512 2>&1 | grep '^{' | head -1 >> $O/bf16_tandem_ablations.log
done
cat $O/bf16_tandem_ablations.log | cut -c1-200
# (d2) the same A/B of the ragged mask for the duo / pipe kernels (d <= 256)
for v in maskall base maskall base; do
  SDPA_HIP_LIB=$PKG/lib/variants/libsdpa_hip_$v.so timeout 200 python tools/gpu_bf16_bench.py 256 128 64 2>&1 | grep '^{' | sed "s/^/$v: /" >> $O/bf16_ragged_mask_hoist.log
done
cut -c1-160 $O/bf16_ragged_mask_hoist.log
# (e)
(cd /tmp && SDPA_VIRTUAL_GPUS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_v2 -o t -- python $R/tools/gpu_hostlevel.py config4 > $R/$O/trace_v2.log 2>&1)
python tools/summarize_overlap.py $O/trace_v2 > $O/config4_2_loopback_ranks_overlap_default_reserve.txt 2>&1
tail -3 $O/config4_2_loopback_ranks_overlap_default_reserve.txt; grep total_ms $O/trace_v2.log | tail -1 | cut -c1-300
rm -rf $O/trace_v2
