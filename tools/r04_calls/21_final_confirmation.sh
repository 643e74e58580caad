#!/bin/bash
# round 4, call 21: the committed tree as the driver will see it: whole GPU suite, smoke, the bench line
O=gpurun_out/r04_21; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/suite.log 2>&1; rc=$?
echo "suite rc=$rc $(grep -aE ' passed| failed' $O/suite.log | tail -1 | cut -c1-100)"
if [ $rc -ne 0 ]; then grep -an "Memory access fault\|SIGABRT\|Fatal\|^FAILED\|assert" $O/suite.log | head -20 | cut -c1-300; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python -c "
import json; j=json.load(open('$O/bench_n1.json'))
print(j['ms_per_step'], j['value'], j['roofline']['frac'], 'boundary', j['boundary']['ms'], j['boundary']['pinned_caller_arrays']['ms'], 'sc3', j['scaling_config3']['ms_per_step'])"
