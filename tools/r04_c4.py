"""configs[4]-like reads (metagenome, 0.5 % errors) at several sizes: time of pass 1 and pass 2 (preprocess only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparrowhawk_amd import AssemblyHelper, synth
dev = torch.device("cuda", 0)
k, err, seed, L = 31, 0.005, 0xEC05, 150
lengths, weights = synth.metagenome_spec(2000, 3_000_000, 1.0, seed)
genomes, goff = synth.device_genomes(torch, dev, lengths, seed)
for n in [int(x) for x in sys.argv[1:]] or [3_000_000, 6_000_000, 12_000_000, 25_000_000]:
    d = synth.device_sample_reads(torch, dev, genomes, goff, weights, n, L, k, seed, err=err, read_index0=3 * 25_000_000)
    for it in range(2):
        h = AssemblyHelper.new(k, False, 2, 20, 0, False, False, False, False)
        h.preprocess_packed_device(d.words.data_ptr(), d.seg_off.data_ptr(), d.n_seg, d.n_bases, d.n_reads)
        t = h.timings()
        h.free()
    print(n, {kk: round(v, 2) for kk, v in t.items() if kk in ("partition_kernel", "count_kernel", "partition_retry", "preprocess_device_total_host_clock")}, "env", {e: os.environ[e] for e in os.environ if e.startswith("SHK_")}, flush=True)
    del d
