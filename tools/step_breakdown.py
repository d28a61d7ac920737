"""Wall-clock of each C-ABI call of one bench step (host clock, mean over steps)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sparrowhawk_amd import AssemblyHelper, _lib
dev = torch.device("cuda", 0)
d_bases, d_seg, n_reads, n_bases, genome = bench.make_reads_on_device(torch, dev, 5_000_000, 100, 150, 0xEC02)
L = _lib.load()
raw_get = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p)(("shk_get_assembly", L))
acc = {}
def T(name, f):
    t0 = time.perf_counter(); r = f(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0); return r
N = 20
for it in range(N + 3):
    if it == 3: acc.clear()
    h = T("new", lambda: AssemblyHelper.new(31, False, 5, 20, 0, False, False, False, False))
    T("preprocess", lambda: h.preprocess_packed_device(d_bases.data_ptr(), d_seg.data_ptr(), n_reads, n_bases, n_reads))
    T("assemble", lambda: h.assemble())
    T("get_assembly", lambda: raw_get(h._h))
    t = T("timings", lambda: h.timings())
    T("free", lambda: h.free())
for k, v in acc.items(): print("%-14s %.3f ms" % (k, v / N * 1e3))
print("sum %.3f ms" % (sum(acc.values()) / N * 1e3))
for k in sorted(t): print("   %-45s %.3f" % (k, t[k]))
