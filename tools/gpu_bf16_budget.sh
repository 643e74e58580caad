#!/bin/bash
# The per-step cycle budget of the bf16 tandem kernel at BASELINE config 5 ("... or a measured
# per-step cycle budget (MFMA 2048 cyc + what remains) showing the floor of THIS tiling with the DMA cost removed").
# For the shipped kernel and for timing-only ablation builds (-DSDPA_TANDEM_ABL=bits, tools/build_variant.sh; built on the
# build host): kernel ms, then one PMC pass -- GPU cycles (GRBM_GUI_ACTIVE), matrix-pipe busy, wave cycles split into
# waiting (s_waitcnt / barrier), issue-stalled and issuing.  A workgroup walks 2048 K/V tiles: cycles per step =
# GPU cycles / 2048; the MFMA floor is 64 MFMAs x 32 cycles = 2048.
R=${GRAFT_REPO_ROOT:-/root/repo}
PKG=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
O=$R/gpurun_out/bf16_budget; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for tag in base tabl4 tabl2 tabl8 tabl16 tabl6 tabl14 tabl30 tabl31; do
  lib=$PKG/lib/variants/libsdpa_hip_$tag.so
  [ $tag = base ] && lib=$PKG/lib/libsdpa_hip.so
  [ -f $lib ] || continue
  for rep in 1 2; do SDPA_HIP_LIB=$lib timeout 200 python $R/tools/gpu_bf16_bench.py 512 2>/dev/null | head -1 | sed "s/^/$tag timing /" >> $O/budget.log; done
  SDPA_HIP_LIB=$lib timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_$tag -o b -- python $R/tools/gpu_bf16_bench.py 512 > $O/pmc_$tag.log 2>&1
  SDPA_HIP_LIB=$lib timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc2_$tag -o b -- python $R/tools/gpu_bf16_bench.py 512 > $O/pmc2_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections, json
out = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/bf16_budget'
rows = []
for d in sorted(glob.glob(out + '/pmc_*')) + sorted(glob.glob(out + '/pmc2_*')):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d).split('_', 1)[1]
    acc = collections.defaultdict(list)
    for p in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            if 'fused_bf16_tandem' in r['Kernel_Name'] and r.get('Grid_Size', '') not in ('',):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    # (the bench launches the d = 512 kernel ~10 + warm-up times and the 8192 x 8192 d = 128 case on another kernel)
    rows.append((tag, os.path.basename(d).split('_')[0], {c: sum(v) / len(v) for c, v in acc.items()}, {c: len(v) for c, v in acc.items()}))
for tag, kind, avg, cnt in rows:
    print(tag, kind, json.dumps({k: round(v, 1) for k, v in avg.items()}), 'dispatches', max(cnt.values()) if cnt else 0)
PY
cat $O/budget.log
