"""Same-box A/B of the bf16 fused kernel (BASELINE config 5) between differently built copies of the library.
Every library is loaded into ONE process (ctypes) and timed through the device-level C ABI
(sdpa_dev_shard_partial_bf16 on a side stream, HIP events), interleaved lib after lib, round after round, so that
clock state and box are the same for all of them (tools/gpu_lib_ab.py is the fp32 twin).
    python tools/gpu_bf16_ab.py [--garbage] [name=path ...]    default: shipped library + every lib/variants/*.so
Each library converts the SAME fp64 operands with ITS OWN converters (the image layout belongs to the library);
--garbage: one set of random bf16 images for everybody (timing-only builds whose kernels expect another layout).
Prints one JSON line per (shape, library): per-round launch times, median, fraction of 2.5 PF, max |delta| of the
normalised rows against the first library's."""
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
SHAPES = [("config5", 32768, 65536, 512, 512)]
if os.environ.get("AB_SHAPES"):
    SHAPES = [tuple([s.split(":")[0]] + [int(x) for x in s.split(":")[1:]]) for s in os.environ["AB_SHAPES"].split(",")]
ROUNDS = int(os.environ.get("AB_ROUNDS", "5"))
garbage = "--garbage" in sys.argv

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

vp, ci, cl, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t
handles = {}
for name, path in libs.items():
    lib = ctypes.CDLL(path)
    lib.sdpa_version.restype = ctypes.c_char_p
    lib.sdpa_dev_workspace_bytes_bf16.restype = sz
    lib.sdpa_dev_workspace_bytes_bf16.argtypes = [ci] * 4
    lib.sdpa_dev_bf16_ld.restype = ci
    lib.sdpa_dev_bf16_dvp.restype = ci
    lib.sdpa_dev_bf16_ldn.restype = cl
    lib.sdpa_dev_bf16_ldn.argtypes = [cl]
    lib.sdpa_dev_shard_partial_bf16.restype = ci
    lib.sdpa_dev_shard_partial_bf16.argtypes = [vp, ci, vp, ci, vp, cl, vp, ci, vp, vp, ci, ci, ci, ci, vp, sz, vp]
    lib.sdpa_dev_cvt_d2bf_q.argtypes = [vp, vp, cl, ci, ci, vp]
    if hasattr(lib, "sdpa_dev_cvt_d2bf_k"):                   # round 6: converters that know the shape's image layout
        lib.sdpa_dev_cvt_d2bf_k.argtypes = [vp, vp, cl, ci, ci, vp]
        lib.sdpa_dev_cvt_d2bf_v.argtypes = [vp, vp, cl, ci, vp]
    lib.sdpa_dev_cvt_d2bf.argtypes = [vp, vp, cl, ci, ci, vp]
    lib.sdpa_dev_cvt_d2bf_t.argtypes = [vp, vp, cl, ci, ci, cl, vp]
    handles[name] = lib
    print(json.dumps({"lib": name, "path": os.path.relpath(path, ROOT), "version": lib.sdpa_version().decode()}), flush=True)

dev = torch.device("cuda:0")
st = torch.cuda.Stream(device=dev)
g = torch.Generator(device=dev)
g.manual_seed(7)
summary = {}
for tag, m, n, dk, dv in SHAPES:
    Q = torch.rand((m, dk), generator=g, device=dev, dtype=torch.float64) * 2 - 1
    K = torch.rand((n, dk), generator=g, device=dev, dtype=torch.float64) * 2 - 1
    V = torch.rand((n, dv), generator=g, device=dev, dtype=torch.float64) * 2 - 1
    first = next(iter(handles.values()))
    ldk, dvp, ldn = first.sdpa_dev_bf16_ld(dk), first.sdpa_dev_bf16_dvp(dv), first.sdpa_dev_bf16_ldn(n)
    ldo = (dv + 3) // 4 * 4
    img = {}
    with torch.cuda.stream(st):
        for k, lib in handles.items():
            Qb = torch.zeros((m, ldk), dtype=torch.int16, device=dev)
            Kb = torch.zeros((ldn, ldk), dtype=torch.int16, device=dev)
            Vt = torch.zeros((dvp * ldn,), dtype=torch.int16, device=dev)
            assert lib.sdpa_dev_cvt_d2bf_q(Q.data_ptr(), Qb.data_ptr(), m, dk, ldk, st.cuda_stream) == 0
            if garbage or not hasattr(lib, "sdpa_dev_cvt_d2bf_k"):
                assert lib.sdpa_dev_cvt_d2bf(K.data_ptr(), Kb.data_ptr(), n, dk, ldk, st.cuda_stream) == 0
                assert lib.sdpa_dev_cvt_d2bf_t(V.data_ptr(), Vt.data_ptr(), n, dv, dvp, ldn, st.cuda_stream) == 0
            else:
                assert lib.sdpa_dev_cvt_d2bf_k(K.data_ptr(), Kb.data_ptr(), n, dk, dv, st.cuda_stream) == 0
                assert lib.sdpa_dev_cvt_d2bf_v(V.data_ptr(), Vt.data_ptr(), n, dv, st.cuda_stream) == 0
            img[k] = (Qb, Kb, Vt)
        st.synchronize()
    del Q, K, V
    contrib = torch.empty((m, ldo), device=dev)
    lmax = torch.empty((m,), device=dev)
    lsum = torch.empty((m,), device=dev)
    ws = {k: torch.empty((max(16, lib.sdpa_dev_workspace_bytes_bf16(m, n, dk, dv)),), dtype=torch.uint8, device=dev)
          for k, lib in handles.items()}
    torch.cuda.synchronize()

    def launch(k):
        lib = handles[k]
        Qb, Kb, Vt = img[k]
        rc = lib.sdpa_dev_shard_partial_bf16(Qb.data_ptr(), ldk, Kb.data_ptr(), ldk, Vt.data_ptr(), ldn, contrib.data_ptr(),
                                             ldo, lmax.data_ptr(), lsum.data_ptr(), m, n, dk, dv, ws[k].data_ptr(),
                                             ws[k].numel(), st.cuda_stream)
        assert rc == 0, (k, rc)

    reps = 8
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
                    outs[k] = (contrib[:, :dv] / lsum[:, None]).clone()
                    if os.environ.get("AB_STAMP"):     # -DSDPA_TANDEM_STAMP builds: cycles per step of [A], [B], the fence, and the step count
                        ls = lsum.cpu().numpy()
                        rows = [[[round(float(x), 1) for x in ls[qb * 128 + w * 32: qb * 128 + w * 32 + 4]] for w in range(4)] for qb in (0, 1, 100, 255)]
                        print(json.dumps({"lib": k, "stamps_A_B_fence_steps": rows}), flush=True)
    base = next(iter(handles))
    for k in handles:
        med = float(np.median(times[k]))
        err = float((outs[k] - outs[base]).abs().max())
        summary[(tag, k)] = med
        flop = 2.0 * m * n * (dk + dv)
        print(json.dumps({"shape": tag, "m": m, "n": n, "dk": dk, "dv": dv, "lib": k, "ms_rounds": [round(x, 4) for x in times[k]],
                          "ms_median": round(med, 4), "tflops": round(flop / (med * 1e-3) / 1e12, 1),
                          "frac_of_2.5PF": round(flop / (med * 1e-3) / 2.5e15, 4),
                          "max_abs_delta_vs_%s" % base: err, "finite": bool(torch.isfinite(outs[k]).all()),
                          "includes_redo_pass_launch": True}), flush=True)
    del img, outs
print("\n%-18s" % "shape" + "".join("%14s" % k for k in handles))
for tag, *_ in SHAPES:
    print("%-18s" % tag + "".join("%14.4f" % summary[(tag, k)] for k in handles))
