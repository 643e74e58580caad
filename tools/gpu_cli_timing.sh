#!/bin/bash
# One-shot CLI timing (fresh process, fresh pages) with and without host page-locking.
python - <<'PY'
import numpy as np, struct
for name,(m,n,d) in {"headline":(32768,65536,128),"config2":(8192,8192,128)}.items():
    rng=np.random.default_rng(1)
    with open("/tmp/%s.bin"%name,"wb") as f:
        f.write(struct.pack("<4i",m,n,d,d))
        for shape in ((m,d),(n,d),(n,d)):
            f.write(rng.uniform(-1,1,shape).tobytes())
        f.write(np.zeros((m,d)).tobytes())
PY
CLI=mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/bin/attention-hip
for f in headline config2; do
  echo "== $f pinned reader (default)"; SDPA_VERBOSE=1 $CLI /tmp/$f.bin 2>&1 | grep -E "total|Elapsed"
  for reg in 1 0; do
  echo "== $f SDPA_PINNED_IO=0 SDPA_HOST_REGISTER=$reg"; SDPA_PINNED_IO=0 SDPA_HOST_REGISTER=$reg SDPA_VERBOSE=1 $CLI /tmp/$f.bin 2>&1 | grep -E "total|Elapsed" ; done; done
