#!/bin/bash
# round 5, call 14: tools/probes/d2h_pattern.hip -- which issue pattern turns a device->host copy into a shader launch
O=gpurun_out/r05_14; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for pat in 0 1 2 3 4 5 6 7 8; do
  for bytes in 4194304 16777216; do
    (cd /tmp && timeout 100 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/t_${pat}_$bytes -o t -- $R/tools/probes/d2h_pattern $pat $bytes > $R/$O/p_${pat}_$bytes.log 2>&1)
    k=$(grep -h copyBuffer $O/t_${pat}_$bytes/*kernel_stats.csv 2>/dev/null | cut -d, -f1-4 | tr -d '"')
    c=$(grep -h MEMORY_COPY $O/t_${pat}_$bytes/*memory_copy_stats.csv 2>/dev/null | cut -d, -f1-4 | tr -d '"' | tr '\n' ' ')
    echo "pattern $pat bytes $bytes | $(grep -h '^pattern' $O/p_${pat}_$bytes.log) | shader: ${k:-none} | engine: ${c:-none}" | tee -a $O/patterns.log
    rm -rf $O/t_${pat}_$bytes
  done
done
