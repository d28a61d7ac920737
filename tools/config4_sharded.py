"""configs[4] share of one GPU (25 M reads of the 2 000-genome metagenome) through the SHARDED path with a one-rank RCCL
communicator, beside the plain path: what the unitig graph on the host, the stitching and the writer cost at 4 M contigs.
Usage: python tools/config4_sharded.py [n_reads ...]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparrowhawk_amd import AssemblyHelper, synth
from sparrowhawk_amd.dist import LibComm, sharded_preprocess_rccl
dev = torch.device("cuda", 0)
k, err, seed, L = 31, 0.005, 0xEC05, 150
lengths, weights = synth.metagenome_spec(2000, 3_000_000, 1.0, seed)
genomes, goff = synth.device_genomes(torch, dev, lengths, seed)
comm = LibComm(0, 1)
for n_share in [int(x) for x in sys.argv[1:]] or [25_000_000]:
    d = synth.device_sample_reads(torch, dev, genomes, goff, weights, n_share, L, k, seed, err=err, read_index0=3 * 25_000_000)
    sha = {}
    for sharded in (False, True):
        t0 = time.time()
        h = AssemblyHelper.new(k, False, 2, 20, 0, False, False, False, False)
        try:
            if sharded:
                sharded_preprocess_rccl(h, d.words.data_ptr(), d.seg_off.data_ptr(), d.n_seg, d.n_bases, d.n_reads, comm)
            else:
                h.preprocess_packed_device(d.words.data_ptr(), d.seg_off.data_ptr(), d.n_seg, d.n_bases, d.n_reads)
            t1 = time.time()
            h.assemble()
            a = h.get_assembly()
            sha[sharded] = hashlib.sha256(a.encode()).hexdigest()
            print(n_share, "sharded" if sharded else "plain", "preprocess %.2f s, assemble + text %.2f s" % (t1 - t0, time.time() - t1), len(a),
                  {kk: round(v, 1) for kk, v in h.timings().items() if v >= 1.0}, flush=True)
            del a
        except Exception as e:
            print(n_share, "sharded" if sharded else "plain", "FAILED:", e, {kk: round(v, 1) for kk, v in h.timings().items() if v >= 1.0}, flush=True)
        h.free()
    print(n_share, "identical JSON:", len(sha) == 2 and sha[False] == sha[True], flush=True)
    del d
comm.free()
