"""Does anything behind the end of the packed read stream influence the result?  The packed reads of small random cases are
uploaded with RANDOM words behind their last word (same n_bases / segment table) and counted through the device entry point;
the distinct (k-mer, count) table must equal the oracle's.  Usage: python tools/pad_garbage_check.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from util import make_dataset, run_oracle, sorted_table
from sparrowhawk_amd import AssemblyHelper, pack_fastq
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
bad = 0
for case in range(n_cases):
    k = int(rng.choice([15, 17, 21, 25, 31, 33, 41]))
    rl = int(rng.choice([60, 100, 150]))
    g, fq = make_dataset(int(rng.integers(300, 3000)), int(rng.choice([6, 10, 16, 30])), read_len=rl, err=float(rng.choice([0.0, 0.003, 0.01])), seed=int(rng.integers(1 << 30)))
    recs = fq.decode().split("@r")[1:]
    part = ("@r" + "@r".join(recs[int(rng.integers(0, 4))::4])).encode()          # a quarter of the reads, like one of four ranks
    bases, seg, nb, nr = pack_fastq(part, k, 0)
    used = (nb + 15) // 16
    padded = np.concatenate([bases[:used].copy(), rng.integers(0, 1 << 32, 256, dtype=np.uint64).astype(np.uint32)])
    if nb % 16:                                                                   # garbage in the unused bits of the last word too
        padded[used - 1] |= np.uint32((int(rng.integers(0, 1 << 32)) << (2 * (nb % 16))) & 0xFFFFFFFF)
    d_bases = torch.from_numpy(padded.view(np.int32)).to(dev); d_seg = torch.from_numpy(seg.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    h = AssemblyHelper.new(k, False, 0, 0, 0, False, False, False, False)
    h.preprocess_packed_device(d_bases.data_ptr(), d_seg.data_ptr(), len(seg) - 1, nb, nr)
    hk, hc, _ = sorted_table(*h.distinct())
    o = run_oracle([part], k=k, min_count=0, min_qual=0)
    ok_, oc_ = o.distinct()
    if not (np.array_equal(hk, ok_) and np.array_equal(hc, oc_)):
        bad += 1
        print("case", case, "k", k, "rl", rl, "reads", nr, "bases", nb, "rows", len(hc), "oracle rows", len(oc_), "DIFFER", flush=True)
    h.free()
print("%d of %d cases differ" % (bad, n_cases))
