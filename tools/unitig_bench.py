"""Host unitig graph (csrc/unitig_graph.cpp) at metagenome scale: N isolated unitigs + a share of linked / forked ones, both
strands.  Usage: SHK_UG_DEBUG=1 python tools/unitig_bench.py [n_unitigs]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sparrowhawk_amd import _lib
L = _lib.load()
k = 31
n_u = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(3)
MASK = (1 << 62) - 1
def revcomp(x):                                         # 31-mers in 62 bits, first base most significant
    x = (~x) & np.uint64(MASK)
    out = np.zeros_like(x)
    for i in range(31):
        out |= ((x >> np.uint64(2 * i)) & np.uint64(3)) << np.uint64(2 * (30 - i))
    return out
f = rng.integers(0, MASK, n_u, dtype=np.uint64); l = rng.integers(0, MASK, n_u, dtype=np.uint64)
# a tenth of the unitigs: i -> i+1 joined by a simple link (last of i overlaps first of i+1 by k-1)
j = np.arange(0, n_u // 10 * 2, 2)
l[j] = (l[j] & np.uint64(3 << 60)) | (f[j + 1] >> np.uint64(2))
n = 2 * n_u
first = np.zeros(n, dtype=np.uint64); last = np.zeros(n, dtype=np.uint64)
first[0::2] = f; last[0::2] = l; first[1::2] = revcomp(l); last[1::2] = revcomp(f)
ln = np.full(n, 40, dtype=np.uint64); kc = np.full(n, 400, dtype=np.uint64); circ = np.zeros(n, dtype=np.uint8)
L.shk_host_unitig_assemble.restype = C.c_void_p
t0 = time.time()
ptr = L.shk_host_unitig_assemble(k, n, first.ctypes.data, last.ctypes.data, ln.ctypes.data, kc.ctypes.data, circ.ctypes.data, None, None, None, 1, 1)
dt = time.time() - t0
text = C.string_at(ptr)[:200].decode()
print("%d records: %.2f s (including the text of the result)" % (n, dt), text.split("\n")[0])
L.shk_host_free(C.c_void_p(ptr))
