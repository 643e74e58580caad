#!/bin/bash
# PMC A/B of $SDPA_TUNE variants: MFMA busy cycles vs GPU active cycles (effective clock, pipe utilisation)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in ${TUNES:-0 112}; do
  SDPA_TUNE=$t rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/t$t -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/t$t.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/pmc_ab'
for d in sorted(glob.glob(out+'/t*')):
    if not os.path.isdir(d): continue
    acc=collections.defaultdict(list); dur=[]
    for p in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            if 'fused' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
                if 'Start_Timestamp' in r and r['Counter_Name']=='GRBM_GUI_ACTIVE': dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    for p in glob.glob(d+'/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            if 'fused' in r['Kernel_Name']: dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    m={k:sum(v)/len(v) for k,v in acc.items()}
    ns=sum(dur)/len(dur) if dur else float('nan')
    gui=m.get('GRBM_GUI_ACTIVE',float('nan'))/8
    print(os.path.basename(d), 'dur_ms=%.3f'%(ns/1e6), 'gui_cycles/xcd=%.4g'%gui, 'clock_GHz=%.3f'%(gui/ns), 'mfma_util=%.4f'%(m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/gui), {k:'%.4g'%v for k,v in m.items()})
PY
