#!/bin/bash
# round 5, call 16: tools/probes/hostcvt_placement_probe.cpp -- what bounds the converter pool at ~134 GB/s of fp64 source on the GPU box's
# host whatever its thread count (16..64): thread placement, the source pages' NUMA node, the item size?  (no GPU work)
O=gpurun_out/r05_16; mkdir -p $O
PKG=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd
timeout 300 tools/probes/hostcvt_placement_probe $PWD/$PKG/lib/libsdpa_hip.so > $O/placement.log 2>&1; echo "rc=$?"; cat $O/placement.log | cut -c1-200
