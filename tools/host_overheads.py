"""host-side cost of the calls bench.py makes around a step (new, timings, free), microseconds"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparrowhawk_amd import AssemblyHelper, synth
from bench import make_reads_on_device
dev = torch.device("cuda:0")
d_bases, d_seg, n_reads, n_bases, genome = make_reads_on_device(torch, dev, 5_000_000, 100, 150, 0xEC02)
def step(timings=True):
    t = [time.perf_counter()]
    h = AssemblyHelper.new(31, False, 5, 20, 0, False, False, False, False); t.append(time.perf_counter())
    h.preprocess_packed_device(d_bases.data_ptr(), d_seg.data_ptr(), n_reads, n_bases, n_reads); t.append(time.perf_counter())
    h.assemble(); t.append(time.perf_counter())
    if timings: h.timings()
    t.append(time.perf_counter())
    info = (h.n_solid, h.n_distinct); t.append(time.perf_counter())
    h.free(); t.append(time.perf_counter())
    return [1e6 * (b - a) for a, b in zip(t, t[1:])]
for _ in range(3): step()
import numpy as np
rows = np.array([step() for _ in range(30)])
print("new %.0f  preprocess %.0f  assemble %.0f  timings %.0f  counters %.0f  free %.0f  | total %.0f us" % (*rows.mean(0), rows.sum(1).mean()))
