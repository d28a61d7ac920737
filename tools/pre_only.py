"""Timing experiment: preprocess only (count step), prints the stage timings."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from sparrowhawk_amd import AssemblyHelper
dev = torch.device("cuda", 0)
ERR = float(os.environ.get("ERR", "0")); K = int(os.environ.get("K", "31"))
if ERR > 0:
    from sparrowhawk_amd import synth
    dr = synth.device_reads(torch, dev, 5_000_000, 3_333_334, 150, K, 0xEC03, err=ERR, mask_errors=os.environ.get("MASK", "0") == "1")
    d_bases, d_seg, n_reads, n_bases = dr.words, dr.seg_off, dr.n_seg, dr.n_bases
else:
    d_bases, d_seg, n_reads, n_bases, genome = bench.make_reads_on_device(torch, dev, 5_000_000, 100, 150, 0xEC02)
acc = {}
for it in range(6):
    h = AssemblyHelper.new(K, False, 5, 20, 0, False, False, False, False)
    try:
        h.preprocess_packed_device(d_bases.data_ptr(), d_seg.data_ptr(), n_reads, n_bases, n_reads)
    except Exception as e:
        print("preprocess:", e)
    t = h.timings()
    if it >= 2:
        for k, v in t.items(): acc[k] = acc.get(k, 0) + v / 4
    h.free()
print({k: round(v, 3) for k, v in acc.items()})
