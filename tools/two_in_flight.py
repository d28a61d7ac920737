"""Two handles in flight on one GPU (two host threads, two streams): the steps bench.py times as `value_two_in_flight`,
alone in a process so that a rocprofv3 kernel trace of it can be read by tools/busy_report.py.
Usage: python tools/two_in_flight.py [steps] [threads]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sparrowhawk_amd import AssemblyHelper
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n_thr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
d_bases, d_seg, n_reads, n_bases, genome = bench.make_reads_on_device(torch, dev, 5_000_000, 100, 150, 0xEC02)


def step():
    h = AssemblyHelper.new(31, False, 5, 20, 0, False, False, False, False)
    h.preprocess_packed_device(d_bases.data_ptr(), d_seg.data_ptr(), n_reads, n_bases, n_reads)
    h.assemble()
    assert h.get_assembly()
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
    print("%d steps, %d in flight: %.3f ms per step" % (n, n_thr, (time.perf_counter() - t0) / n * 1e3), flush=True)
