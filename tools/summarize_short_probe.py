"""Cut the kernel trace of tools/gpu_short_kernel_probe.py by fused-launch index: per shape and mode
the fused kernel's own mean duration, the idle gap in front of it and the span of one iteration."""
import csv, glob, sys
out = sys.argv[1]
paths = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fused_idx = [i for i, r in enumerate(rows) if "fused_pipelined_kernel" in r["Kernel_Name"]]
N, WARM = 30, 5
per_shape = WARM + 4 * N
print("fused launches in trace: %d (expected %d per shape)" % (len(fused_idx), per_shape))
for s in range(len(fused_idx) // per_shape):
    base = s * per_shape + WARM
    for mode in range(4):
        idx = fused_idx[base + mode * N: base + (mode + 1) * N]
        dur = [(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3 for i in idx]
        gap = [(int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 for i in idx[1:]]
        span = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 for a, b in zip(idx[1:-1], idx[2:])]
        grid = rows[idx[0]]["Grid_Size_X"]
        print("shape %d mode %d grid %s: fused own %.1f us (min %.1f max %.1f), gap in front %.1f us, iteration span %.1f us"
              % (s, mode, grid, sum(dur[1:]) / len(dur[1:]), min(dur), max(dur), sum(gap) / len(gap), sum(span) / len(span)))
        if mode == 3 and s == 0:   # the kernels of one iteration, in order
            a, b = idx[5], idx[6]
            for i in range(a, b):
                print("      %-60s dur %.1f us, starts %.1f us after the previous kernel ended"
                      % (rows[i]["Kernel_Name"][:60], (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3,
                         (int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3))
