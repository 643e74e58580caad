"""Where do the host-side converts' ~100 GB/s come from?  (no GPU work; run on the GPU box for its host)
Rates of sdpa_host_cvt_rows from T threads, source = a numpy array first-touched on a chosen NUMA node,
destination = numpy (pageable) or sdpa_host_alloc (page-locked), threads unpinned / pinned to the source's
node / pinned to the other node.   python tools/gpu_hostcvt_probe.py"""
import ctypes, importlib, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
lib = pkg.load()


def node_cpus(n):
    out = []
    for part in open("/sys/devices/system/node/node%d/cpulist" % n).read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
allcpus = sorted(os.sched_getaffinity(0))
rows, cols = 65536, 512                                   # 268 MB of fp64, config 5's K
print(json.dumps({"numa_nodes": nodes, "cpus": len(allcpus), "rows": rows, "cols": cols}))


def run(src, dst_ptr, kind, threads, cpus):
    per = rows // threads
    el = 4 if kind == 0 else 2
    def work(i):
        if cpus is not None:
            os.sched_setaffinity(0, cpus)
        r0 = i * per
        n = per if i + 1 < threads else rows - r0
        lib.sdpa_host_cvt_rows(src.ctypes.data + r0 * cols * 8, dst_ptr + r0 * cols * el, n, cols, cols, kind, 1.0, 0)
    best = None
    for _ in range(4):
        ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return rows * cols * 8 / best / 1e9


libc = ctypes.CDLL(None, use_errno=True)


def page_node(addr):
    """NUMA node of the page at addr: get_mempolicy(MPOL_F_NODE | MPOL_F_ADDR); None when the call is not allowed"""
    node = ctypes.c_int(-1)
    rc = libc.syscall(239, ctypes.byref(node), None, 0, ctypes.c_void_p(addr), 1 | 2)
    return node.value if rc == 0 else None


for src_node in nodes[:2]:
    os.sched_setaffinity(0, node_cpus(src_node))           # first touch on this node
    src = np.random.default_rng(0).uniform(-1, 1, (rows, cols))
    dst_np = np.zeros((rows, cols), dtype=np.float32)      # also first-touched here
    pin = lib.sdpa_host_alloc(rows * cols * 4)
    ctypes.memset(pin, 0, rows * cols * 4)
    os.sched_setaffinity(0, allcpus)
    print(json.dumps({"first_touch_node": src_node, "get_mempolicy_says": {"src": page_node(src.ctypes.data), "numpy_dst": page_node(dst_np.ctypes.data),
                      "pinned_dst": page_node(pin)}}))
    other = [n for n in nodes if n != src_node][:1]
    for kind in (0, 1):
        for threads in (8, 32, 64):
            row = {"src_node": src_node, "kind": "f32" if kind == 0 else "bf16", "threads": threads}
            for name, cpus in (("unpinned", None), ("on_src_node", node_cpus(src_node)),
                               ("on_other_node", node_cpus(other[0]) if other else None)):
                row["numpy_dst_GBps_" + name] = round(run(src, dst_np.ctypes.data, kind, threads, cpus), 1)
                row["pinned_dst_GBps_" + name] = round(run(src, pin, kind, threads, cpus), 1)
            print(json.dumps(row), flush=True)
    lib.sdpa_host_free(pin)
