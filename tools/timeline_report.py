"""Kernels and memory copies of the last milliseconds of a rocprofv3 trace (--kernel-trace --memory-copy-trace) as phases
per stream: consecutive events of one stream less than `gap_us` apart are merged into a phase (first .. last event name).
Usage: python tools/timeline_report.py <dir> [last_ms] [gap_us]"""
import csv, glob, sys
d = sys.argv[1]
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
gap = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 15e3
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
mt = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(kt[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r.get("Stream_Id", r.get("Queue_Id", "?")), r["Kernel_Name"].split("(")[0].replace("void shk::", "").replace("shk::", "")[:28]))
if mt:
    for r in csv.DictReader(open(mt[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Stream_Id", "?"), r["Direction"].replace("MEMORY_COPY_", "")))
t_end = max(e[1] for e in ev); t0 = t_end - int(last_ms * 1e6)
ev = sorted(e for e in ev if e[1] >= t0)
def union(iv):
    iv = sorted(iv); tot = 0; cs = ce = None
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None: tot += ce - cs
            cs, ce = s, e
        else: ce = max(ce, e)
    if ce is not None: tot += ce - cs
    return tot
ku = union([(s, e) for s, e, k, _, _ in ev if k == "K"]); cu = union([(s, e) for s, e, k, _, _ in ev if k == "C"]); bu = union([(s, e) for s, e, _, _, _ in ev])
print("last %.1f ms: kernels busy %.2f ms, copies busy %.2f ms, either %.2f ms (idle %.2f), both at once %.2f ms" % (last_ms, ku / 1e6, cu / 1e6, bu / 1e6, last_ms - bu / 1e6, (ku + cu - bu) / 1e6))
streams = sorted(set((k, st) for _, _, k, st, _ in ev))
for k, st in streams:
    mine = [e for e in ev if e[2] == k and e[3] == st]
    phases = []
    for s, e, _, _, name in mine:
        if phases and s - phases[-1][1] < gap:
            phases[-1][1] = max(phases[-1][1], e); phases[-1][3] = name; phases[-1][4] += e - s; phases[-1][5] += 1
        else:
            phases.append([s, e, name, name, e - s, 1])
    print("%s stream %s:" % ("kernels" if k == "K" else "copies ", st))
    for s, e, a, b, busy, n in phases:
        print("   %+8.3f .. %+8.3f ms  (%6.3f ms, busy %6.3f, %3d events)  %s .. %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, busy / 1e6, n, a, b))
