"""From a rocprofv3 --kernel-trace of a multi-batch P > 1 run on loopback ranks: for every collective
kernel (loop_reduce* / loop_gather*) the fused kernels that were running on the device at the same time.
    python tools/summarize_overlap.py <dir with *_kernel_trace.csv>
Shows that batch b's merge collectives run UNDER batch b+1's fused kernels (attention-mpi.c:364-380)."""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", "?")))
rows.sort(key=lambda r: r[1])
fused = [r for r in rows if "fused_" in r[0]]
# the merge tail of a batch: loopback collectives (SDPA_VIRTUAL_GPUS), RCCL's kernels (ncclDevKernel_*, forced one-rank
# communicator or real ranks), the merge kernels and the fp32 -> fp64 widening that follow them on the comm stream
coll = [r for r in rows if any(k in r[0] for k in ("loop_reduce", "loop_gather", "nccl", "merge_gathered", "merge_rescale",
                                                    "merge_normalise", "cvt_f2d"))]
t0 = rows[0][1] if rows else 0
print("%d dispatches, %d fused launches, %d collective kernels" % (len(rows), len(fused), len(coll)))
n_over = n_hideable = n_slow = 0
tail = coll[-12:] if len(coll) > 48 else coll
for name, a, b, st in coll:
    over = [(fa, fb, fs) for _, fa, fb, fs in fused if fa < b and fb > a]
    n_over += 1 if over else 0
    # a tail kernel CAN hide only if fused work of a later batch exists when it becomes ready: a fused launch that is
    # running at its start or starts within 2 ms of it (the last batch of a call has none: nothing left to run under)
    hideable = any(fa < a + 2_000_000 and fb > a for _, fa, fb, fs in fused)
    n_hideable += 1 if hideable else 0
    # ... and it has only really run UNDER the fused kernel if it did not simply wait for that kernel's end
    if over and any(abs(b - fb) < 100_000 and (b - a) > 1_000_000 for fa, fb, fs in over):
        n_slow += 1
    if (name, a, b, st) in tail:
        import re
        mm = re.search(r"(loop_\w+|ncclDevKernel\w*|merge_\w+|cvt_f2d\w*)", name)
        short = mm.group(1) if mm else name[:28]
        print("%-28s stream %-4s %9.1f .. %9.1f us (%6.1f us)  concurrent fused launches: %s" % (
            short, st, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3,
            ", ".join("stream %s %.0f..%.0f us" % (fs, (fa - t0) / 1e3, (fb - t0) / 1e3) for fa, fb, fs in over) or "none"))
print("collective kernels that ran while a fused kernel was running: %d of %d" % (n_over, len(coll)))
print("  of the %d that had a later batch's fused kernel to run under: %d did (%.0f %%); %d of them took > 1 ms and ended "
      "with the fused kernel (started beside it, waited for its slots)" % (
          n_hideable, n_over, 100.0 * n_over / max(1, n_hideable), n_slow))
