"""Same-box A/B of the fp32 fused kernel between differently built copies of the library.
Every library is loaded into ONE process (ctypes, RTLD_LOCAL) and timed through the device-level C ABI
(sdpa_dev_shard_partial_f32 on a side stream, HIP events), interleaved lib after lib, round after round, so that
clock state and box are the same for all of them.
    python tools/gpu_lib_ab.py [name=path ...]     default: shipped library + every lib/variants/*.so
Prints one JSON line per (shape, library) with the per-round launch times and the median, then a summary table."""
import ctypes
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
SHAPES = [("headline", 32768, 65536, 128), ("config2", 8192, 8192, 128), ("d256", 32768, 65536, 256),
          ("rank_share_1of8", 32768, 8192, 128)]
ROUNDS = int(os.environ.get("AB_ROUNDS", "5"))

libs = {}
args = [a for a in sys.argv[1:] if "=" in a]
if args:
    for a in args:
        k, v = a.split("=", 1)
        libs[k] = v
else:
    libs["shipped"] = os.path.join(PKG, "lib", "libsdpa_hip.so")
    for so in sorted(glob.glob(os.path.join(PKG, "lib", "variants", "*.so"))):
        libs[os.path.basename(so)[len("libsdpa_hip_"):-3]] = so

vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
handles = {}
for name, path in libs.items():
    lib = ctypes.CDLL(path)
    lib.sdpa_version.restype = ctypes.c_char_p
    lib.sdpa_dev_workspace_bytes.restype = sz
    lib.sdpa_dev_workspace_bytes.argtypes = [ci] * 4
    lib.sdpa_dev_shard_partial_f32.restype = ci
    lib.sdpa_dev_shard_partial_f32.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, vp, sz, vp]
    handles[name] = lib
    print(json.dumps({"lib": name, "path": os.path.relpath(path, ROOT), "version": lib.sdpa_version().decode()}), flush=True)

dev = torch.device("cuda:0")
st = torch.cuda.Stream(device=dev)
g = torch.Generator(device=dev)
g.manual_seed(7)
summary = {}
for tag, m, n, d in SHAPES:
    ld = 64 if d <= 64 else 128 if d <= 128 else 256
    Q = torch.rand((m, ld), generator=g, device=dev) * 2 - 1
    K = torch.rand((n, ld), generator=g, device=dev) * 2 - 1
    V = torch.rand((n, ld), generator=g, device=dev) * 2 - 1
    contrib = torch.empty((m, ld), device=dev)
    lmax = torch.empty((m,), device=dev)
    lsum = torch.empty((m,), device=dev)
    ws = {k: torch.empty((max(16, lib.sdpa_dev_workspace_bytes(m, n, d, d)),), dtype=torch.uint8, device=dev)
          for k, lib in handles.items()}
    torch.cuda.synchronize()

    def launch(k):
        lib = handles[k]
        rc = lib.sdpa_dev_shard_partial_f32(Q.data_ptr(), ld, K.data_ptr(), ld, V.data_ptr(), ld, contrib.data_ptr(), ld,
                                            lmax.data_ptr(), lsum.data_ptr(), m, n, d, d, ws[k].data_ptr(), ws[k].numel(),
                                            st.cuda_stream)
        assert rc == 0, (k, rc)

    reps = 8 if m * n >= 1 << 30 else 60
    times = {k: [] for k in handles}
    outs = {}
    with torch.cuda.stream(st):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15:          # clock pre-warm
            launch(next(iter(handles)))
            st.synchronize()
        for rnd in range(ROUNDS):
            for k in handles:
                launch(k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(reps):
                    launch(k)
                e1.record(st)
                st.synchronize()
                times[k].append(e0.elapsed_time(e1) / reps)
                if rnd == 0:
                    outs[k] = (contrib.clone(), lmax.clone(), lsum.clone())
    base = next(iter(handles))
    for k in handles:
        med = float(np.median(times[k]))
        same = all(torch.equal(a, b) for a, b in zip(outs[k], outs[base]))
        summary[(tag, k)] = med
        print(json.dumps({"shape": tag, "m": m, "n": n, "d": d, "lib": k, "ms_rounds": [round(x, 4) for x in times[k]],
                          "ms_median": round(med, 4), "tflops": round(4.0 * m * n * d / (med * 1e-3) / 1e12, 2),
                          "frac_of_157.3": round(4.0 * m * n * d / (med * 1e-3) / 1e12 / 157.3, 4),
                          "bitwise_equal_to_%s" % base: same, "includes_split_merge": True}), flush=True)
print("\n%-18s" % "shape" + "".join("%14s" % k for k in handles))
for tag, *_ in SHAPES:
    print("%-18s" % tag + "".join("%14.4f" % summary[(tag, k)] for k in handles))
