#!/bin/bash
# PMC passes of the config-5 bf16 kernel for the shipped library and the variant libraries named on the command line
# (tags of lib/variants/libsdpa_hip_<tag>.so; "shipped" = lib/libsdpa_hip.so): wave cycles split into waiting / issue-stalled /
# issuing, LDS issue stalls, instruction counts -- per dispatch of the main kernel, mean over the launches.
#   bash tools/gpu_bf16_pmc_ab.sh [--garbage] shipped t2p ...
R=${GRAFT_REPO_ROOT:-/root/repo}
PKG=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
O=$R/gpurun_out/bf16_pmc_ab; mkdir -p $O
G=""; [ "$1" = "--garbage" ] && { G="--garbage"; shift; }
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  lib=$PKG/lib/variants/libsdpa_hip_$tag.so
  [ $tag = shipped ] && lib=$PKG/lib/libsdpa_hip.so
  [ -f $lib ] || { echo "no $lib"; continue; }
  AB_ROUNDS=2 timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/pmc_$tag -o b -- python $R/tools/gpu_bf16_ab.py $G $tag=$lib > $O/pmc_$tag.log 2>&1
  AB_ROUNDS=2 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc2_$tag -o b -- python $R/tools/gpu_bf16_ab.py $G $tag=$lib > $O/pmc2_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections, json
out = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/bf16_pmc_ab'
for d in sorted(glob.glob(out + '/pmc_*')) + sorted(glob.glob(out + '/pmc2_*')):
    if not os.path.isdir(d):
        continue
    kind, tag = os.path.basename(d).split('_', 1)
    acc = collections.defaultdict(list)
    for p in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            if 'fused_bf16_tandem' in r['Kernel_Name'] or 'fused_bf16_tiled' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(tag, kind, json.dumps({c: round(sum(v) / len(v), 1) for c, v in acc.items()}), 'dispatches', max((len(v) for v in acc.values()), default=0))
PY
