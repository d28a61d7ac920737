"""Debug: configs[4] share through the plain path WITHOUT correction (the graph the sharded path contracts)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparrowhawk_amd import AssemblyHelper, synth
dev = torch.device("cuda", 0)
k, err, seed, L = 31, 0.005, 0xEC05, 150
lengths, weights = synth.metagenome_spec(2000, 3_000_000, 1.0, seed)
genomes, goff = synth.device_genomes(torch, dev, lengths, seed)
for n_share in [int(x) for x in sys.argv[1:]] or [25_000_000]:
    d = synth.device_sample_reads(torch, dev, genomes, goff, weights, n_share, L, k, seed, err=err, read_index0=3 * 25_000_000)
    for nb, nd in ((True, True), (False, False)):
        h = AssemblyHelper.new(k, False, 2, 20, 0, False, False, nb, nd)
        try:
            h.preprocess_packed_device(d.words.data_ptr(), d.seg_off.data_ptr(), d.n_seg, d.n_bases, d.n_reads)
            h.assemble()
            print(n_share, "no correction" if nb else "corrected", "ok, n_solid", h.n_solid, {kk: round(v, 1) for kk, v in h.timings().items() if "splitters" in kk or "regroup" in kk}, flush=True)
        except Exception as e:
            print(n_share, "no correction" if nb else "corrected", "FAILED:", e, flush=True)
        h.free()
    del d
