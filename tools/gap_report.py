"""GPU idle time between consecutive kernels of the last bench step, from a rocprofv3 kernel trace.
Usage: python tools/gap_report.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# steps start at k_partition
starts = [i for i, e in enumerate(ev) if "k_partition" in e[2]]
if len(starts) < 2:
    sys.exit("need at least two steps")
a, b = starts[-2], starts[-1]
step = ev[a:b]
t0 = step[0][0]
busy = sum(e[1] - e[0] for e in step)
span = ev[b][0] - t0
print("step span %.3f ms, kernels busy %.3f ms, idle %.3f ms, %d kernels" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(step)))
gaps = []
for i in range(len(step)):
    nxt = step[i + 1][0] if i + 1 < len(step) else ev[b][0]
    gaps.append((nxt - step[i][1], step[i][2].split("(")[0][-40:], (step[i + 1][2] if i + 1 < len(step) else ev[b][2]).split("(")[0][-40:]))
for g, x, y in sorted(gaps, reverse=True)[:14]:
    print("  idle %7.1f us after %-40s before %s" % (g / 1e3, x, y))
