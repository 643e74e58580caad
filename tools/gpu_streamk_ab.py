"""Stream-K work distribution of the fp32 pipelined kernels against the classic equal-split grid (round 4).
Per shape and stream (whole chip / 8 or 16 CUs reserved by grid size / 8 reserved by CU mask, round 3's way): kernel + split-merge ms per launch with
$SDPA_STREAMK=0 (classic), =1 (stream-K forced), unset (the cost model's choice), the slab counts, whether the
triples equal the classic ones bit for bit, and the error of 48 rows against the fp64 restatement.
    python tools/gpu_streamk_ab.py [quick]"""
import ctypes, importlib, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
lib = pkg.load()
be = pkg.HipBackend("cuda:0")
dev = torch.device("cuda:0")
quick = "quick" in sys.argv
shapes = [(32768, 65536, 128), (33000, 65536, 128), (40000, 65536, 128), (32768, 8192, 128), (8192, 8192, 128),
          (8320, 8192, 128), (32768, 65536, 64), (32768, 32768, 256), (131072, 16384, 128), (4096, 262144, 128)]
if quick:
    shapes = [shapes[0], shapes[1], shapes[3], shapes[4]]


def reserving_stream(reserve, by_mask=False):
    """sdpa_dev_stream_create(reserve): the reservation by grid size (round 4) or, for A/B, by CU mask (round 3)"""
    if reserve == 0:
        return torch.cuda.Stream(device=dev)
    os.environ["SDPA_RESERVE_BY_MASK"] = "1" if by_mask else "0"
    sp = ctypes.c_void_p()
    pkg._lib.check(lib.sdpa_dev_stream_create(reserve, ctypes.byref(sp)), "sdpa_dev_stream_create")
    os.environ.pop("SDPA_RESERVE_BY_MASK", None)
    return torch.cuda.ExternalStream(sp.value, device=dev)


streams = {"0": reserving_stream(0), "8": reserving_stream(8), "16": reserving_stream(16), "8 (CU mask)": reserving_stream(8, True)}
g = torch.Generator(device="cuda"); g.manual_seed(1)
rng = np.random.default_rng(3)
for m, n, d in shapes:
    K = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    V = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    Q = torch.rand((m, d), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    rows = np.sort(rng.choice(m, 48, replace=False))
    want = O.numpy_attention_f64(Q.cpu().numpy(), K.cpu().numpy(), V.cpu().numpy(), rows)
    sa = pkg.ShardedAttention(be)
    sa.load_kv_shard_f64(K, V, n, d, d)
    qf = sa.convert_q(Q)
    torch.cuda.synchronize()
    base = None
    for reserve in streams:
        st = streams[reserve]
        for knob in ("0", "1", None):
            if knob is None:
                os.environ.pop("SDPA_STREAMK", None)
            else:
                os.environ["SDPA_STREAMK"] = knob
            pkg.reload_env()
            with torch.cuda.stream(st):
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.06:      # clock pre-warm
                    out = sa.batch_partial(qf)
                    st.synchronize()
                reps = 6 if m * n >= 1 << 30 else 40
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(reps):
                    out = sa.batch_partial(qf)
                e1.record(st)
                st.synchronize()
                ms = e0.elapsed_time(e1) / reps
                res = be.finish_f64(out[0], out[2], d)
                st.synchronize()
            trip = tuple(t.clone() for t in out)
            if base is None:
                base = trip
            same = all(torch.equal(a[:, :d] if a.dim() == 2 else a, b[:, :d] if b.dim() == 2 else b) for a, b in zip(trip, base))
            err = float(np.abs(res.cpu().numpy()[rows] - want).max())
            print(json.dumps({"shape": [m, n, d], "reserve_cus": reserve, "SDPA_STREAMK": knob or "auto",
                              "ms": round(ms, 4), "tflops": round(4.0 * m * n * d / (ms * 1e-3) / 1e12, 1),
                              "bitwise_equal_to_classic_whole_chip": same, "err48": err}), flush=True)
os.environ.pop("SDPA_STREAMK", None)
