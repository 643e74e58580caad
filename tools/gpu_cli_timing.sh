#!/bin/bash
# One-shot CLI timing: fresh process, file -> verdict.  `wall` is the whole process (exec to exit),
# the CLI's own stderr line breaks it down; "Elapsed time" is the timed attention() call.
# Reader variants: pinned (default), pageable (SDPA_PINNED_IO=0), and the read->H2D overlap
# (SDPA_CLI_PREFETCH=1, which moves K/V out of the timed region).
R=${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY'
import numpy as np, struct
for name,(m,n,d) in {"headline":(32768,65536,128),"config2":(8192,8192,128),"config4":(131072,65536,128)}.items():
    rng=np.random.default_rng(1)
    with open("/tmp/%s.bin"%name,"wb") as f:
        f.write(struct.pack("<4i",m,n,d,d))
        for shape in ((m,d),(n,d),(n,d)):
            f.write(rng.uniform(-1,1,shape).tobytes())
        f.write(np.zeros((m,d)).tobytes())
PY
CLI=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/bin/attention-hip
MPICLI=$R/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/bin/attention-mpi-hip
run() {   # label, env..., file
  local label=$1; shift
  local t0=$(date +%s.%N)
  env "$@" > /tmp/cli.out 2> /tmp/cli.err
  local t1=$(date +%s.%N)
  echo "== $label: wall $(python -c "print('%.0f ms' % (($t1-$t0)*1e3))") | $(grep Elapsed /tmp/cli.out) | $(grep 'wall clock' /tmp/cli.err | sed 's/attention-hip: //')"
}
for f in headline config2 config4; do
  for rep in 1 2; do
  run "$f pinned reader (default) #$rep" SDPA_VERBOSE=1 $CLI /tmp/$f.bin
  run "$f SDPA_CLI_PREFETCH=1 #$rep" SDPA_VERBOSE=1 SDPA_CLI_PREFETCH=1 $CLI /tmp/$f.bin
  run "$f SDPA_PINNED_IO=0 #$rep" SDPA_VERBOSE=1 SDPA_PINNED_IO=0 $CLI /tmp/$f.bin
  done
done
run "headline MPI flavour, mpiexec -n 4" SDPA_VERBOSE=1 /opt/conda/bin/mpiexec -n 4 $MPICLI /tmp/headline.bin
grep "start -> result" /tmp/cli.err
