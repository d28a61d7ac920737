"""Rate of the gzip reader (shk_host_gunzip = the reader of shk_preprocess) on a single-member .fastq.gz of `MB` megabytes
of text, by thread count; the bytes are compared with zlib's.  VERDICT r2 item 3: >= 2 GB/s of output on the GPU box's cores."""
import ctypes as C, gzip, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sparrowhawk_amd import _lib, synth
L = _lib.load()
MB = int(os.environ.get("MB", 256))


def gunzip(b):
    out, n, mt, sec = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_double()
    rc = L.shk_host_gunzip(b, len(b), C.byref(out), C.byref(n), C.byref(mt), C.byref(sec))
    dt = sec.value                                        # the reader itself (the hook's copy of the result is not part of shk_preprocess)
    assert rc == 0, rc
    r = C.string_at(out.value, n.value)
    L.shk_host_free(out)
    return r, mt.value, dt


n_reads = MB * 1000000 // 316
g = synth.random_genome(5_000_000, 1)
codes, quals = synth.sample_reads(g, n_reads, 150, 2, err=0.005)
fq = bytes(synth.to_fastq_fixed(codes, quals))
del codes, quals
t0 = time.perf_counter()
z = gzip.compress(fq, compresslevel=int(os.environ.get("LEVEL", 6)))
print("text %.0f MB -> gzip %.0f MB (%.1f s to compress)" % (len(fq) / 1e6, len(z) / 1e6, time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); ref = zlib.decompress(z, 31); tz = time.perf_counter() - t0
assert ref == fq
print("python zlib.decompress: %.2f s = %.2f GB/s" % (tz, len(fq) / tz / 1e9), flush=True)
before = 0
for T in [int(x) for x in os.environ.get("THREADS", "1,8,16,32,64,128").split(",")]:
    os.environ["SHK_GUNZIP_THREADS"] = str(T)
    for rep in range(2):
        r, mt, dt = gunzip(z)
        print("threads %3d run %d: %.3f s = %.2f GB/s of text  (multi-threaded members so far: %d)  equal to zlib: %s" % (T, rep, dt, len(fq) / dt / 1e9, mt, r == fq), flush=True)
        assert r == fq
