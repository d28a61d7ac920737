"""GPU busy time (union of the kernel intervals) over the last steps of a rocprofv3 kernel trace in which several handles
run at once: what fraction of the span the GPU had at least one kernel running.
Usage: python tools/busy_report.py <kernel_trace.csv> [steps_to_take]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
take = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if "k_partition" in e[2]]
if len(starts) < take + 1:
    sys.exit("need more steps in the trace")
win = ev[starts[-take - 1]:starts[-1]]                 # `take` whole steps' worth of kernels (interleaved across handles)
t0, t1 = win[0][0], max(e[1] for e in win)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
summed = sum(e - s for s, e, _ in win)
span = t1 - t0
print("%d steps: span %.3f ms (%.3f per step), GPU busy (union of kernels) %.3f ms = %.1f %% of the span, idle %.3f ms per step; "
      "sum of kernel durations %.3f ms per step (kernels of two handles overlap where small ones leave CUs free)"
      % (take, span / 1e6, span / 1e6 / take, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6 / take, summed / 1e6 / take))
