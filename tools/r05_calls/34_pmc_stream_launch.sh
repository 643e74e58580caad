#!/bin/bash
# round 5, call 34: PMC passes over the boundary call of config 5 in bf16 (the persistent fused_bf16_tandem_stream_kernel<512>): where do the ~1.3 ms between its 4.8 ms and
# the kernel's own 3.3-3.5 ms go?  (counters in their own runs, kernel-trace only; per-dispatch means of the stream kernel)
O=gpurun_out/r05_34; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/gpu_hostlevel.py config5:bf16"
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $R/$O/pmc_sq -o t -- $CMD > $R/$O/pmc_sq.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$O/pmc_tcc -o t -- $CMD > $R/$O/pmc_tcc.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS --output-format csv -d $R/$O/pmc_inst -o t -- $CMD > $R/$O/pmc_inst.log 2>&1
cd $R
python - <<'P'
import csv, glob, collections
for d in ("pmc_sq", "pmc_tcc", "pmc_inst"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/r05_34/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "tandem" in k or "fused_bf16" in k:
                acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(d, k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in sorted(cs.items())})
P
grep -h total_ms $O/pmc_sq.log | tail -1 | cut -c1-300
rm -rf $O/pmc_sq $O/pmc_tcc $O/pmc_inst
