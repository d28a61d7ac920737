"""Debug: plain vs sharded (one rank) on a 5 M-read slice of the metagenome: solid sets, adjacency bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sparrowhawk_amd import AssemblyHelper, synth
from sparrowhawk_amd.dist import LibComm, sharded_preprocess_rccl
dev = torch.device("cuda", 0)
k, err, seed, L = 31, 0.005, 0xEC05, 150
n_share = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
lengths, weights = synth.metagenome_spec(2000, 3_000_000, 1.0, seed)
genomes, goff = synth.device_genomes(torch, dev, lengths, seed)
d = synth.device_sample_reads(torch, dev, genomes, goff, weights, n_share, L, k, seed, err=err, read_index0=3 * 25_000_000)
comm = LibComm(0, 1)
res = {}
for sharded in (False, True):
    h = AssemblyHelper.new(k, False, 2, 20, 0, False, False, True, True)       # no correction: the adjacency bytes stay comparable
    if sharded:
        sharded_preprocess_rccl(h, d.words.data_ptr(), d.seg_off.data_ptr(), d.n_seg, d.n_bases, d.n_reads, comm, n_partitions=int(os.environ.get("N_PART", "0")))
    else:
        h.preprocess_packed_device(d.words.data_ptr(), d.seg_off.data_ptr(), d.n_seg, d.n_bases, d.n_reads)
    keys, cnt = h.solid()
    o = np.lexsort(tuple(keys[:, j] for j in range(keys.shape[1])))
    print("sharded" if sharded else "plain", "n_solid", h.n_solid, "n_distinct", h.n_distinct, "instances", h.total_instances,
          "unique keys", len(np.unique(keys[:, 0])), flush=True)
    try:
        if os.environ.get("NO_ASM"): raise RuntimeError("skipped")
        h.assemble()
        ok = True
    except Exception as e:
        print("assemble:", e, flush=True); ok = False
    try:
        a0, a1, al = h.adjacency()
        keys2, _ = h.solid()
        o2 = np.lexsort(tuple(keys2[:, j] for j in range(keys2.shape[1])))
        res[sharded] = (keys2[o2], a0[o2])
    except Exception as e:
        print("adjacency:", e, flush=True)
        res[sharded] = (keys[o], None)
    print({kk: round(v, 2) for kk, v in h.timings().items() if "adjacency" in kk or "graph" in kk}, flush=True)
    h.free()
print("same solid set:", np.array_equal(res[False][0], res[True][0]))
if res[False][1] is not None and res[True][1] is not None:
    diff = np.flatnonzero(res[False][1] != res[True][1])
    print("adjacency bytes that differ:", len(diff), "of", len(res[False][1]))
    for i in diff[:10]:
        print("  key %016x plain %02x sharded %02x" % (int(res[False][0][i, 0]), int(res[False][1][i]), int(res[True][1][i])))
comm.free()
