"""Stage times of a circular 5 Mbp isolate + 50 kbp plasmid against the same replicons sampled as linear ones."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sparrowhawk_amd import AssemblyHelper, synth
dev = torch.device("cuda", 0)
k, L, COV = 31, 150, 100
lens_ = np.array([5_000_000, 50_000], dtype=np.int64)
genomes, goff = synth.device_genomes(torch, dev, lens_, 0xC1C)
w = lens_ / lens_.sum()
n_reads = int(lens_.sum()) * COV // L
def run(d):
    h = AssemblyHelper.new(k, True, 5, 20, 0, False, False, False, False)
    h.preprocess_packed_device(d.words.data_ptr(), d.seg_off.data_ptr(), d.n_seg, d.n_bases, d.n_reads)
    h.assemble()
    t = h.timings(); h.free(); return t
for circ in (True, False):
    d = synth.device_sample_reads(torch, dev, genomes, goff, w, n_reads, L, k, 0xC1C, circular=circ)
    ts = [run(d) for _ in range(8)][3:]
    keys = sorted(ts[0])
    print("circular" if circ else "linear")
    for kk in keys:
        if kk.startswith("outputs") or kk.startswith("collapse") or kk.startswith("assemble") or kk.startswith("correct") or kk.startswith("adj") or kk.startswith("graph"):
            print("   %-40s %.4f" % (kk, min(t[kk] for t in ts)))
