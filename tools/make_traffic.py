"""profiles/traffic.json from a pmc_summary.txt: HBM-side bytes per launch of every shk kernel.
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3), collected in separate passes; FETCH_SIZE is doubled
as MI355X_MICROARCH.md (HBM section) prescribes for gfx950 (128-byte requests tallied at 64 B).
Usage: python tools/make_traffic.py profiles/<tag>/pmc_summary.txt profiles/<tag>"""
import json, re, sys
cur, d = None, {}
for line in open(sys.argv[1]):
    if line.startswith("shk::"):
        cur = line.strip().split("<")[0].replace("shk::", "")
        d.setdefault(cur, {})
    else:
        m = re.match(r"\s+(\w+)\s+mean\s+(\S+)", line)
        if m and cur:
            d[cur][m.group(1)] = float(m.group(2))
out = {"source": f"{sys.argv[2]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, KiB; FETCH doubled per "
                 "MI355X_MICROARCH.md HBM note)",
       "bytes_per_launch": {k: (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 for k, v in sorted(d.items())}}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out["bytes_per_launch"], indent=1))
