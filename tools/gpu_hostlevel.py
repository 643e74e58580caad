"""Boundary (host fp64 in / out) timing of sdpa_attention_f64 at named shapes, optionally swept
over the pipeline's environment knobs (read per call):
    python tools/gpu_hostlevel.py headline config2 --sweep
Each line: shape, knobs, then the stage breakdown of the best of 5 warm calls."""
import importlib, os, sys, time, json, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
shapes = {"headline": (32768, 65536, 128), "config2": (8192, 8192, 128), "config1": (512, 512, 64),
          "config3": (32768, 262144, 128), "config4": (131072, 65536, 128), "config5": (32768, 65536, 512)}
args = [a for a in sys.argv[1:] if not a.startswith("--")]
sweep = "--sweep" in sys.argv
pinned = "--pinned" in sys.argv
lib = pkg.load()
pkg.init(1)
# --hostcvt: A/B of the convert placement -- each shape first with the device converts ($SDPA_HOST_CVT=0),
# then with host converts (=1) at several thread counts, then the default (auto: chosen per problem); the
# engine is re-created for each
hostcvt = "--hostcvt" in sys.argv
# --widen: A/B of where the result is widened to fp64 -- the device ($SDPA_HOST_WIDEN=0: fp64 rows cross PCIe) or host
# threads (=1: fp32 rows cross, attention-mpi.c:373/:396), then the default; interleaved twice so that drift shows
widen = "--widen" in sys.argv
# --register: what the boundary costs WITHOUT page-locking the caller's arrays ($SDPA_HOST_REGISTER=0) -- fp64 from
# pageable memory to the device converts, or through the library's own page-locked staging (host converts / host widening)
register = "--register" in sys.argv
# --streamed: A/B of the first batch's form (round 5) -- one launch per K/V chunk ($SDPA_STREAMED=0) against ONE persistent
# launch that follows its inputs (=1, the default), interleaved; then the group size of the streamed form
streamed = "--streamed" in sys.argv
KNOBS = ("SDPA_STREAMED", "SDPA_STREAM_CHUNK_MIN", "SDPA_QBATCH", "SDPA_KV_CHUNK_MIN", "SDPA_KV_CHUNK_MAX", "SDPA_ROW_PIECES", "SDPA_PIECE_MIN_ROWS", "SDPA_HOST_REGISTER",
         "SDPA_PROGRESSIVE_PIN")
BASE_DEBUG = os.environ.get("SDPA_DEBUG", "")      # the caller's own $SDPA_DEBUG stays in force under every row
DEBUG = {"SDPA_STREAM_CHUNK_MIN": "stream_chunk_min", "SDPA_KV_CHUNK_MIN": "kv_chunk_min", "SDPA_KV_CHUNK_MAX": "kv_chunk_max", "SDPA_ROW_PIECES": "row_pieces",
         "SDPA_PIECE_MIN_ROWS": "piece_min_rows", "SDPA_PROGRESSIVE_PIN": "progressive_pin"}      # -> $SDPA_DEBUG="name=value,..."
SWEEP = [{},
         {"SDPA_ROW_PIECES": 1},
         {"SDPA_ROW_PIECES": 2},
         {"SDPA_ROW_PIECES": 8},
         {"SDPA_ROW_PIECES": 8, "SDPA_KV_CHUNK_MIN": 2048},
         {"SDPA_PIECE_MIN_ROWS": 2048},
         {"SDPA_KV_CHUNK_MIN": 2048},
         {"SDPA_KV_CHUNK_MIN": 8192},
         {"SDPA_KV_CHUNK_MAX": 8192},
         {"SDPA_KV_CHUNK_MAX": 32768},
         {"SDPA_KV_CHUNK_MAX": 65536},
         {"SDPA_QBATCH": 16384},
         {"SDPA_QBATCH": 8192},
         {"SDPA_KV_CHUNK_MIN": 1 << 20, "SDPA_ROW_PIECES": 1},     # round-1 structure: nothing streamed
         ]


def hostbuf(a):
    if not pinned:
        return a, None
    ptr = lib.sdpa_host_alloc(a.nbytes)
    v = np.ctypeslib.as_array((ctypes.c_double * a.size).from_address(ptr)).reshape(a.shape)
    v[...] = a
    return v, ptr


for name in args or ["headline", "config2", "config1"]:
    prec = "bf16" if name.endswith(":bf16") else None     # e.g. config5:bf16
    m, n, d = shapes[name.split(":")[0]]
    rng = np.random.default_rng(0)
    Q, K, V = (rng.uniform(-1, 1, s) for s in ((m, d), (n, d), (n, d)))
    (Q, pq), (K, pk), (V, pv) = hostbuf(Q), hostbuf(K), hostbuf(V)
    R, pr = hostbuf(np.zeros((m, d)))
    flags = 2 if prec else 0
    CVT = ([{"SDPA_HOST_CVT": 0}] + [{"SDPA_HOST_CVT": 1, "SDPA_HOST_CVT_THREADS": t} for t in (16, 32, 48, 64, 96)] + [{}]) if hostcvt else [{}]
    WID = [{"SDPA_HOST_WIDEN": 0}, {"SDPA_HOST_WIDEN": 1}, {"SDPA_HOST_WIDEN": 0}, {"SDPA_HOST_WIDEN": 1}, {}]
    REG = [{"SDPA_HOST_REGISTER": 1}, {"SDPA_HOST_REGISTER": 0},
           {"SDPA_HOST_REGISTER": 0, "SDPA_HOST_CVT": 0, "SDPA_HOST_WIDEN": 0},
           {"SDPA_HOST_REGISTER": 0, "SDPA_HOST_CVT": 1, "SDPA_HOST_WIDEN": 0},
           {"SDPA_HOST_REGISTER": 0, "SDPA_HOST_CVT": 1, "SDPA_HOST_WIDEN": 1},
           {"SDPA_HOST_REGISTER": 1}, {}]
    STR = [{"SDPA_STREAMED": 0}, {"SDPA_STREAMED": 1}, {"SDPA_STREAMED": 0}, {"SDPA_STREAMED": 1},
           {"SDPA_STREAM_CHUNK_MIN": 2048}, {"SDPA_STREAM_CHUNK_MIN": 8192}, {"SDPA_ROW_PIECES": 8}, {"SDPA_ROW_PIECES": 2}, {}]
    for knobs in (SWEEP if sweep else WID if widen else REG if register else STR if streamed else CVT):
        for k in KNOBS + (("SDPA_HOST_CVT", "SDPA_HOST_CVT_THREADS", "SDPA_HOST_WIDEN") if (hostcvt or widen or register) else ()):
            os.environ.pop(k, None)           # (outside those modes a caller's $SDPA_HOST_CVT_THREADS etc. stay in force)
        os.environ.pop("SDPA_DEBUG", None)
        dbg = ([BASE_DEBUG] if BASE_DEBUG else []) + ["%s=%s" % (DEBUG[k], v) for k, v in knobs.items() if k in DEBUG]     # the tuning knobs live in ONE variable
        if dbg:
            os.environ["SDPA_DEBUG"] = ",".join(dbg)
        for k, v in knobs.items():
            if k not in DEBUG:
                os.environ[k] = str(v)
        if hostcvt or widen or register or streamed:
            pkg.shutdown()
            pkg.init(1)
            if lib.sdpa_prepare(m, n, d, d, flags) != 0:
                raise SystemExit("sdpa_prepare failed")
        best = None
        for it in range(6):
            t0 = time.perf_counter()
            rc = lib.sdpa_attention_f64(Q.ctypes.data, K.ctypes.data, V.ctypes.data, R.ctypes.data, m, n, d, d, flags)
            dt = time.perf_counter() - t0
            assert rc == 0, rc
            t = pkg.last_timing(); t["wall_ms"] = dt * 1e3
            if it and (best is None or t["total_us"] < best["total_us"]): best = t
        row = {"shape": name, "pinned": pinned, "knobs": knobs}
        for k in ("total_us", "head_us", "tail_us", "register_us", "kv_stage_us", "pipeline_us", "kernel_us"):
            row[k.replace("_us", "_ms")] = round(best[k] / 1e3, 3)
        for k in ("q_batches", "kv_chunks", "fused_launches", "kv_splits", "host_convert_threads", "host_widen", "streamed", "last_kernel"):
            row[k] = best[k]
        row["kernel_tflops"] = round(4.0 * m * n * d / (best["kernel_us"] * 1e-6) / 1e12, 1)
        print(json.dumps(row), flush=True)
    for p in (pq, pk, pv, pr):
        if p: lib.sdpa_host_free(p)
