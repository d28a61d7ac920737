"""Packed reads in host PINNED memory -> contigs (the SURVEY 8(d) clock), one or two handles in flight, alone in a process:
under `rocprofv3 --kernel-trace --memory-copy-trace` the trace shows whether the uploads of one handle run under the kernels
of the other (tools/timeline_report.py).  Usage: python tools/host_pinned_timeline.py [steps] [threads]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ctypes
from sparrowhawk_amd import AssemblyHelper, _lib
raw_get_assembly = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p)(("shk_get_assembly", _lib.load()))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n_thr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
d_bases, d_seg, n_reads, n_bases, genome = bench.make_reads_on_device(torch, dev, 5_000_000, 100, 150, 0xEC02)
hw = torch.empty(d_bases.numel(), dtype=torch.int32).pin_memory(); hs = torch.empty(d_seg.numel(), dtype=torch.int32).pin_memory()
hw.copy_(d_bases); hs.copy_(d_seg); torch.cuda.synchronize()


def step():
    h = AssemblyHelper.new(31, False, 5, 20, 0, False, False, False, False)
    h.preprocess_packed_host(hw.data_ptr(), hs.data_ptr(), n_reads, n_bases, n_reads)
    h.assemble()
    assert raw_get_assembly(h._h)        # (the pointer: no 15 MB Python copy inside the timed loop)
    h.free()


lock, todo = threading.Lock(), [0]


def worker(n):
    while True:
        with lock:
            if todo[0] >= n:
                return
            todo[0] += 1
        step()


for n in (4, steps):
    todo[0] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(n,)) for _ in range(n_thr)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    torch.cuda.synchronize()
    print("%d steps, %d in flight, host pinned: %.3f ms per step" % (n, n_thr, (time.perf_counter() - t0) / n * 1e3), flush=True)
