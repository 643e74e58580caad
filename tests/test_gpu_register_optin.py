"""GPU (-m gpu): $SDPA_HOST_REGISTER=1 -- page-locking the caller's arrays for the duration of the call, the default of
rounds 1-3, opt-in since round 4 -- still gives the default path's result bit for bit: progressive and one-go
registration, one rank and loopback ranks, both egresses.

It runs in a PROCESS OF ITS OWN, and that is the point of the round-4 change: a process that has registered and
unregistered heap ranges is no longer safe for plain pageable hipMemcpy from the same addresses (PyTorch's .cuda() of a
numpy array) -- the GPU faults on a host page seconds to minutes later (profiles/r04/gpu_memory_fault_root_cause_*.log;
the -m gpu suite with two such tests in process: 2 faults in 3 runs, call 12).  The child makes no pageable copy of its
own: numpy arrays in, the C host, numpy arrays out."""
import os
import subprocess
import sys

import pytest

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu

CHILD = r'''
import importlib, os, sys
import numpy as np
ROOT, PKG = sys.argv[1:3]
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle as O
pkg = importlib.import_module(PKG)
KNOBS = ("SDPA_HOST_REGISTER", "SDPA_DEBUG", "SDPA_VIRTUAL_GPUS", "SDPA_QBATCH", "SDPA_EGRESS", "SDPA_HOST_CVT", "SDPA_HOST_WIDEN",
         "SDPA_STREAMED")
def engine(**env):
    pkg.shutdown()
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in env.items():
        if k == "SDPA_PROGRESSIVE_PIN":                 # (a test knob: lives in $SDPA_DEBUG since round 6)
            os.environ["SDPA_DEBUG"] = "progressive_pin=%s" % v
        else:
            os.environ[k] = str(v)
    pkg.init(1)
checked = 0
for (m, n, d, prec, env) in [(1500, 9000, 128, None, {}),
                             (16384, 16384, 128, None, {}),                                   # arrays big enough for progressive pinning
                             (700, 9000, 512, "bf16", {}),
                             (1500, 9000, 128, None, {"SDPA_VIRTUAL_GPUS": 3, "SDPA_QBATCH": 512}),
                             (1500, 9000, 128, None, {"SDPA_VIRTUAL_GPUS": 2, "SDPA_QBATCH": 512, "SDPA_EGRESS": "root"})]:
    Q, K, V = O.make_inputs(m, n, d, d, "D2", seed=m + n)
    engine(**env)
    want = pkg.attention(Q, K, V, precision=prec)
    t = pkg.last_timing()
    assert t["register_us"] == 0, t
    ref = O.numpy_attention_f64(Q[:64], K, V)
    tol = (1e-2 if prec == "bf16" else 5e-5) * max(1.0, float(np.abs(V).max()))
    assert np.abs(want[:64] - ref).max() <= tol
    # device converts cannot feed the streamed first batch (round 5: no kernel becomes resident beside that launch), so
    # that variant runs one launch per K/V chunk: its bitwise twin is the default with $SDPA_STREAMED=0 (another split /
    # merge tree than the streamed launch's: equal within the tolerance, not bit for bit)
    engine(**env, SDPA_STREAMED=0)
    want_chunked = pkg.attention(Q, K, V, precision=prec)
    assert pkg.last_timing()["streamed"] == 0
    assert np.abs(want_chunked - want).max() <= tol
    for reg in ({"SDPA_HOST_REGISTER": 1}, {"SDPA_HOST_REGISTER": 1, "SDPA_PROGRESSIVE_PIN": 0},
                {"SDPA_HOST_REGISTER": 1, "SDPA_HOST_CVT": 0, "SDPA_HOST_WIDEN": 0}):
        engine(**env, **reg)
        for rep in range(2):
            got = pkg.attention(Q, K, V, precision=prec)
            assert np.array_equal(got, want_chunked if "SDPA_HOST_CVT" in reg else want), (m, n, d, prec, env, reg, rep)
        t = pkg.last_timing()
        # (with host converts / host widening chosen per problem some or all arrays need no registration; with both
        #  forced off every array of a MiB or more is registered)
        assert t["register_us"] > 0 if "SDPA_HOST_CVT" in reg else t["register_us"] >= 0, t
        checked += 1
print("registered paths agree with the default bit for bit: %d configurations" % checked)
'''


def test_registered_caller_arrays_give_the_default_paths_result_bit_for_bit():
    env = dict(os.environ)
    for k in ("SDPA_HOST_REGISTER", "SDPA_DEBUG", "SDPA_VIRTUAL_GPUS", "SDPA_QBATCH", "SDPA_EGRESS", "SDPA_STREAMED"):
        env.pop(k, None)
    def child():
        return subprocess.run([sys.executable, "-c", CHILD, ROOT, PKG], capture_output=True, text=True, timeout=900, env=env)
    r = child()
    if r.returncode != 0 and "Memory access fault" in r.stderr:
        # The documented hazard of the opt-in itself: not seen in this child so far (it makes no pageable
        # copy of its own).  ONE such fault is the path's known risk and is reported loudly, not silently: the test is
        # repeated in a fresh process, and a second fault FAILS it (an xfail here must not be able to mask a
        # regression in the registered paths).
        sys.stderr.write("\n*** test_gpu_register_optin: GPU memory fault inside the SDPA_HOST_REGISTER=1 child; retrying once ***\n"
                         + r.stderr[-1500:] + "\n")
        r2 = child()
        assert r2.returncode == 0 and "15 configurations" in r2.stdout, (
            "the SDPA_HOST_REGISTER=1 child failed twice in a row", r.stderr[-800:], r2.stdout[-300:], r2.stderr[-1500:])
        pytest.xfail("one GPU memory fault inside the SDPA_HOST_REGISTER=1 child (the hazard the default avoids); the retry passed")
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2500:])
    assert "15 configurations" in r.stdout, r.stdout[-500:]
