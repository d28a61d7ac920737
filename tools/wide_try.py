import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import make_dataset, run_oracle, compare_all
from sparrowhawk_amd import AssemblyHelper
k, circ, mc = int(sys.argv[1]), sys.argv[2] == "1", int(sys.argv[3])
g, fq = make_dataset(30000, 40, read_len=400, err=0.002, seed=300 + k, circular=circ)
h = AssemblyHelper.new(k, True, mc, 0, 0, False, False, False, False)
h.preprocess(fq); print("preprocess ok", flush=True)
h.assemble(); print("assemble ok", flush=True)
o = run_oracle([fq], k=k, min_count=mc, min_qual=0)
compare_all(h, o); print("parity ok", flush=True)
