"""Timeline of the LAST host-level call in a rocprofv3 --kernel-trace --memory-copy-trace run of tools/gpu_hostlevel.py:
every fused launch with the idle time of the compute stream in front of it, and the copies that were in flight then.
    python tools/summarize_timeline.py <dir with *_kernel_trace.csv [and *_memory_copy_trace.csv]> [n_fused_launches_per_call]
Answers "where does the boundary call lose the time its kernels do not account for"."""
import csv, glob, os, re, sys
d = sys.argv[1]
per_call = int(sys.argv[2]) if len(sys.argv) > 2 else 0
K, C = [], []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        C.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?"))))
K.sort(); C.sort()
fused = [k for k in K if "fused_" in k[2]]
if not fused:
    raise SystemExit("no fused launches in the trace")
if not per_call:                      # the calls are separated by the host's think time: split at gaps > 2 ms with no kernel at all
    per_call = 1
    for a, b in zip(fused[:-1][::-1], fused[1:][::-1]):
        if b[0] - a[1] > 2_000_000:
            break
        per_call += 1
last = fused[-per_call:]
t0 = last[0][0]
short = lambda n: (re.search(r"(fused_\w+|cvt_\w+|merge_\w+|split_merge\w*|finish_\w+|loop_\w+|ncclDevKernel\w*|copyBuffer\w*)", n) or re.match(r"(.{0,28})", n)).group(1)
# everything from 3 ms before the call's first fused launch to 2 ms behind its last one
lo, hi = t0 - 3_000_000, last[-1][1] + 2_000_000
ev = [(a, b, "K " + short(n) + " s" + str(s)) for a, b, n, s in K if a >= lo and a <= hi]
ev += [(a, b, "C %s %s B" % (dr, by)) for a, b, dr, by in C if a >= lo and a <= hi]
ev.sort()
print("last call: %d fused launches, first at t = 0; kernels (K) and copies (C), us relative to it" % per_call)
prev_fused_end = None
busy = 0
for a, b, what in ev:
    gap = ""
    if what.startswith("K fused_"):
        if prev_fused_end is not None:
            gap = "   <- compute idle %.1f us before it" % ((a - prev_fused_end) / 1e3) if a - prev_fused_end > 20_000 else ""
        prev_fused_end = b
        busy += b - a
    if what.startswith("K fused_") or (b - a) > 50_000 or what.startswith("C "):
        print("%10.1f .. %10.1f (%8.1f us)  %s%s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, what, gap))
span = last[-1][1] - last[0][0]
print("fused launches: %.3f ms busy over a span of %.3f ms (idle %.3f ms)" % (busy / 1e6, span / 1e6, (span - busy) / 1e6))
