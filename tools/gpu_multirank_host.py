"""Host side of the P > 1 schedule of sdpa_attention_f64 on loopback ranks (SDPA_VIRTUAL_GPUS=P, one GPU):
when is each rank's FIRST fused launch enqueued, when is everything enqueued -- host clock, so the figures
hold on an 8-GPU node -- for the round-3 schedule (one enqueue thread per rank, collectives on the comm
streams, reduce-scatter egress) and the round-2 one (one thread, reduce to the root).
    python tools/gpu_multirank_host.py [config3 headline config4] [--ranks 8]
The device-side totals of P ranks that share ONE GPU are not a multi-GPU measurement; they are printed to
show that nothing regressed on the loopback path."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
shapes = {"headline": (32768, 65536, 128), "config2": (8192, 8192, 128), "config3": (32768, 262144, 128),
          "config4": (131072, 65536, 128)}
args = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]
P = int(sys.argv[sys.argv.index("--ranks") + 1]) if "--ranks" in sys.argv else 8
MODES = [("round 3: threads + comm streams + reduce-scatter", {}),
         ("threads, reduce to root", {"SDPA_EGRESS": "root"}),
         ("round 2: one enqueue thread, reduce to root", {"SDPA_EGRESS": "root", "SDPA_ENQUEUE_THREADS": "0"}),
         ("one enqueue thread, reduce-scatter", {"SDPA_ENQUEUE_THREADS": "0"})]
for name in args or ["config3", "headline", "config4"]:
    m, n, d = shapes[name]
    rng = np.random.default_rng(0)
    Q, K, V = (rng.uniform(-1, 1, s) for s in ((m, d), (n, d), (n, d)))
    ref = None
    for label, env in MODES:
        pkg.shutdown()
        for k in ("SDPA_EGRESS", "SDPA_ENQUEUE_THREADS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        os.environ["SDPA_VIRTUAL_GPUS"] = str(P)
        pkg.init(1)
        best = None
        for it in range(5):
            out = pkg.attention(Q, K, V)
            t = pkg.last_timing()
            if it and (best is None or max(t["enqueue_first_kernel_us"]) < max(best["enqueue_first_kernel_us"])):
                best = t
        ref = out if ref is None else ref
        f = best["enqueue_first_kernel_us"]
        print(json.dumps({"shape": name, "ranks": P, "schedule": label,
                          "first_kernel_enqueued_us": [round(x) for x in f],
                          "rank_spread_us": round(max(f) - min(f)), "enqueue_total_us": round(best["enqueue_total_us"]),
                          "register_us": round(best["register_us"]), "q_batches": best["q_batches"],
                          "total_ms_on_one_gpu": round(best["total_us"] / 1e3, 2),
                          "tail_ms_on_one_gpu": round(best["tail_us"] / 1e3, 2),
                          "bit_identical_to_first_schedule": bool(np.array_equal(out, ref))}), flush=True)
pkg.shutdown()
