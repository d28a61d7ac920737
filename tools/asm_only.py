"""Timing experiment: full step, prints the assemble-stage timings (results are NOT checked)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from sparrowhawk_amd import AssemblyHelper
dev = torch.device("cuda", 0)
d_bases, d_seg, n_reads, n_bases, genome = bench.make_reads_on_device(torch, dev, 5_000_000, 100, 150, 0xEC02)
acc = {}
for it in range(6):
    h = AssemblyHelper.new(31, False, 5, 20, 0, False, False, False, False)
    h.preprocess_packed_device(d_bases.data_ptr(), d_seg.data_ptr(), n_reads, n_bases, n_reads)
    try:
        h.assemble()
    except Exception as e:
        print("assemble:", e)
    t = h.timings()
    if it >= 2:
        for k, v in t.items(): acc[k] = acc.get(k, 0) + v / 4
    h.free()
print({k: round(v, 3) for k, v in acc.items() if "graph" in k or "adj" in k})
