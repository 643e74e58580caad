"""Generate tests/golden/ref_mpi_fp32/<case>_P<p>.f32 with the REFERENCE's own MPI program.

Run in the build container (needs /root/reference, MPICH in /opt/conda):
    make -C oracle ref && python oracle/make_ref_mpi_fp32.py

For every golden case and P in {1, 2, 8} the reference's unmodified fp32 K/V-sharded
`attention()` (attention-mpi.c:191-407) is run under `mpiexec -n P` through the dump harness
oracle/_ref/attention-mpi-dump (oracle/ref_mpi_dump.c) and its raw result array is stored as
float32 (every value IS an fp32 value widened at attention-mpi.c:373/:396).  These files pin the
restatement `oracle_attention_sharded_f32` (tests/test_oracle.py) and are the reference-side
error the GPU tests compare the HIP path's error with (err_gpu / err_ref).
"""
import json
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle as O  # noqa: E402

RANKS = (1, 2, 8)


def main():
    exe = os.path.join(O.HERE, "_ref", "attention-mpi-dump")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/attention-mpi-dump missing: run `make -C oracle ref`")
    gold = os.path.join(os.path.dirname(O.HERE), "tests", "golden")
    out_dir = os.path.join(gold, "ref_mpi_fp32")
    os.makedirs(out_dir, exist_ok=True)
    index = []
    import tempfile
    tmp = tempfile.mkdtemp(dir="/tmp")
    for case in json.load(open(os.path.join(gold, "INDEX.json"))):
        Q, K, V, ans = O.load_golden(case)
        case_path = O.golden_file(case, tmp)
        # (the mid-size cases: one rank count -- the documented no-optimisation build takes its time -- and 8 ranks)
        for p in (RANKS if case.get("file") else (1, 8)):
            name = "%s_P%d.f32" % (case["name"], p)
            path = os.path.join(out_dir, name)
            subprocess.run(["/opt/conda/bin/mpiexec", "-n", str(p), exe, case_path, path],
                           check=True, capture_output=True)
            got = np.fromfile(path, dtype=np.float32).reshape(ans.shape).astype(np.float64)
            err = float(np.abs(got - ans).max())
            index.append(dict(case=case["name"], ranks=p, file=name, max_abs_err_vs_fp64=err,
                              source="reference attention-mpi.c:191-407, mpiexec -n %d (MPICH 3.3.2)" % p))
            print("%-18s P=%d  max|ref_fp32 - fp64 answer| = %.3e" % (case["name"], p, err))
    with open(os.path.join(out_dir, "INDEX.json"), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__":
    main()
