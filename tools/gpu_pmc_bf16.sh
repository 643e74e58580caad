#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_bf16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/a -o b -- python $R/tools/gpu_bf16_bench.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/b -o b -- python $R/tools/gpu_bf16_bench.py > $OUT/b.log 2>&1
python - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/pmc_bf16'
for d in sorted(glob.glob(out+'/[ab]')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            if 'fused_bf16' in r['Kernel_Name']:
                key=r['Kernel_Name'][:60]+' grid='+r.get('Grid_Size','?')
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(k, {c:'%.4g'%(sum(x)/len(x)) for c,x in v.items()})
PY
tail -3 $OUT/b.log
