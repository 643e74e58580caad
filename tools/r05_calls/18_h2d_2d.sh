#!/bin/bash
# round 5, call 18: is a pitched host->device copy the copy engine's (tools/probes/h2d_2d_probe.hip), on ROCm 7.2 and on the runtime PyTorch bundles
O=gpurun_out/r05_18; mkdir -p $O
timeout 100 tools/probes/h2d_2d_probe > $O/rocm72.log 2>&1; echo "rc=$?"; cat $O/rocm72.log | cut -c1-200
TL=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
LD_PRELOAD=$TL/libamdhip64.so:$TL/libhsa-runtime64.so timeout 100 tools/probes/h2d_2d_probe > $O/torch_runtime.log 2>&1; echo "rc=$?"; cat $O/torch_runtime.log | cut -c1-200
LD_PRELOAD=$TL/libamdhip64.so:$TL/libhsa-runtime64.so timeout 100 tools/probes/d2h_probe 16777216 0 2>&1 | head -12 > $O/d2h_torch_runtime.log; cat $O/d2h_torch_runtime.log | cut -c1-200
