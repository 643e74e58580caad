#!/bin/bash
# round 5, call 19: does a device->host copy turn into a shader launch when it is ENQUEUED while the kernel it waits for still runs?
O=gpurun_out/r05_19; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
TL=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
for rt in rocm72 torch; do
  for pat in 4 9 10; do
    pre=""; [ $rt = torch ] && pre="$TL/libamdhip64.so:$TL/libhsa-runtime64.so"
    (cd /tmp && LD_PRELOAD=$pre timeout 100 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/t_${rt}_$pat -o t -- $R/tools/probes/d2h_pattern $pat 16777216 > $R/$O/p_${rt}_$pat.log 2>&1)
    k=$(grep -h copyBuffer $O/t_${rt}_$pat/*kernel_stats.csv 2>/dev/null | cut -d, -f1-4 | tr -d '"')
    c=$(grep -h MEMORY_COPY $O/t_${rt}_$pat/*memory_copy_stats.csv 2>/dev/null | cut -d, -f1-4 | tr -d '"' | tr '\n' ' ')
    echo "$rt pattern $pat | $(grep -h '^pattern' $O/p_${rt}_$pat.log) | shader: ${k:-none} | engine: ${c:-none}" | tee -a $O/pending.log
    rm -rf $O/t_${rt}_$pat
  done
done
