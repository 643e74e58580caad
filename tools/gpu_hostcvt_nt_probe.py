"""Streaming (non-temporal) stores in the host converter rows: on or off?  Rates of sdpa_host_cvt_rows from T threads over
config 5's K (65536 x 512 fp64 = 268 MB), destination page-locked (sdpa_host_alloc) when a GPU is there, else numpy;
flags 4 = ordinary stores, 2 = streaming stores.  No GPU work.   python tools/gpu_hostcvt_nt_probe.py [threads...]"""
import ctypes, importlib, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
lib = pkg.load()
rows, cols = 65536, 512
src = np.random.default_rng(0).uniform(-1, 1, (rows, cols))
threads_list = [int(a) for a in sys.argv[1:]] or [8, 32]
ptr = lib.sdpa_host_alloc(rows * cols * 4)
if ptr:
    dst_ptr, where = ptr, "page-locked"
else:
    keep = np.zeros((rows, cols), dtype=np.float32)
    dst_ptr, where = keep.ctypes.data, "numpy"


def run(kind, threads, flags):
    per = rows // threads
    el = 4 if kind == 0 else 2
    def work(i):
        r0 = i * per
        n = per if i + 1 < threads else rows - r0
        lib.sdpa_host_cvt_rows(src.ctypes.data + r0 * cols * 8, dst_ptr + r0 * cols * el, n, cols, cols, kind, 1.0, flags)
    best = None
    for _ in range(5):
        ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return rows * cols * 8 / best / 1e9


for kind in (0, 1):
    for threads in threads_list:
        for rep in range(2):
            a, b = run(kind, threads, 4), run(kind, threads, 2)
            print(json.dumps({"kind": "f32" if kind == 0 else "bf16", "threads": threads, "dst": where,
                              "ordinary_GBps_of_fp64_source": round(a, 1), "streaming_GBps": round(b, 1), "ratio": round(b / a, 3)}), flush=True)
if ptr:
    lib.sdpa_host_free(ptr)
