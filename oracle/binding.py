"""ctypes binding of oracle/libshk_oracle.so (test infrastructure; never imported by the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_oracle(force=False):
    so = os.path.join(_HERE, "libshk_oracle.so")
    src = os.path.join(_HERE, "shk_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libshk_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build_oracle())
        u64, u32, vp, cp = C.c_uint64, C.c_uint32, C.c_void_p, C.c_char_p
        sig = {
            "shko_new": (vp, [u32, u32, u32, C.c_int, C.c_int, C.c_int]),
            "shko_free": (None, [vp]),
            "shko_last_error": (cp, [vp]),
            "shko_add_read": (C.c_int, [vp, cp, cp, u64]),
            "shko_add_fastq": (C.c_int, [vp, cp, u64]),
            "shko_n_reads": (u64, [vp]), "shko_n_bases": (u64, [vp]),
            "shko_count": (C.c_int, [vp, C.c_int]),
            "shko_total_instances": (u64, [vp]), "shko_n_distinct": (u64, [vp]),
            "shko_get_distinct": (None, [vp, vp, vp]),
            "shko_get_histo": (None, [vp, vp]),
            "shko_used_min_count": (u32, [vp]), "shko_fit_ok": (C.c_int, [vp]),
            "shko_fit": (C.c_int, [vp, vp]),
            "shko_n_solid": (u64, [vp]), "shko_get_solid": (None, [vp, vp, vp]),
            "shko_get_adjacency": (None, [vp, vp]), "shko_get_alive": (None, [vp, vp]),
            "shko_correct": (C.c_int, [vp]), "shko_collapse": (C.c_int, [vp]),
            "shko_assemble": (C.c_int, [vp]),
            "shko_tips_removed": (u64, [vp]), "shko_bubbles_removed": (u64, [vp]),
            "shko_n_contigs": (u64, [vp]), "shko_contig_len": (u64, [vp, u64]),
            "shko_contig_kc": (u64, [vp, u64]), "shko_contig_seq": (cp, [vp, u64]),
            "shko_fasta": (cp, [vp]), "shko_gfa1": (cp, [vp]), "shko_gfa2": (cp, [vp]),
            "shko_dot": (cp, [vp]), "shko_assembly_json": (cp, [vp]),
            "shko_preprocessing_json": (cp, [vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _LIB = L
    return _LIB


def oracle_fit(histo):
    h = np.ascontiguousarray(histo, dtype=np.uint64)
    assert h.shape == (500,)
    out = C.c_uint32(0)
    ok = _lib().shko_fit(h.ctypes.data, C.addressof(out))
    return bool(ok), int(out.value)


class Oracle:
    """Stage-by-stage access to the oracle (SPEC.md S1-S11)."""

    def __init__(self, k=31, min_count=5, min_qual=20, do_fit=False,
                 no_bubble_collapse=False, no_dead_end_removal=False):
        self.L = _lib()
        self.k = k
        self.W = (2 * k + 63) // 64
        self.h = self.L.shko_new(k, min_count, min_qual, int(do_fit), int(no_bubble_collapse),
                                 int(no_dead_end_removal))
        if not self.h:
            raise ValueError("oracle: bad parameters")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.shko_free(self.h)
            self.h = None

    def add_fastq(self, data: bytes):
        if self.L.shko_add_fastq(self.h, data, len(data)) != 0:
            raise ValueError(self.L.shko_last_error(self.h).decode())

    def add_read(self, seq: bytes, qual: bytes = None):
        self.L.shko_add_read(self.h, seq, qual, len(seq))

    def count(self, naive=False):
        if self.L.shko_count(self.h, int(naive)) != 0:
            raise RuntimeError(self.L.shko_last_error(self.h).decode())

    @property
    def n_reads(self): return self.L.shko_n_reads(self.h)
    @property
    def n_bases(self): return self.L.shko_n_bases(self.h)
    @property
    def total_instances(self): return self.L.shko_total_instances(self.h)
    @property
    def used_min_count(self): return self.L.shko_used_min_count(self.h)
    @property
    def fit_ok(self): return bool(self.L.shko_fit_ok(self.h))

    def distinct(self):
        n = self.L.shko_n_distinct(self.h)
        keys = np.zeros((n, self.W), dtype=np.uint64)
        cnt = np.zeros(n, dtype=np.uint32)
        if n:
            self.L.shko_get_distinct(self.h, keys.ctypes.data, cnt.ctypes.data)
        return keys, cnt

    def solid(self):
        n = self.L.shko_n_solid(self.h)
        keys = np.zeros((n, self.W), dtype=np.uint64)
        cnt = np.zeros(n, dtype=np.uint32)
        if n:
            self.L.shko_get_solid(self.h, keys.ctypes.data, cnt.ctypes.data)
        return keys, cnt

    def histo(self):
        h = np.zeros(500, dtype=np.uint64)
        self.L.shko_get_histo(self.h, h.ctypes.data)
        return h

    def adjacency(self):
        n = self.L.shko_n_solid(self.h)
        a = np.zeros(n, dtype=np.uint8)
        if n:
            self.L.shko_get_adjacency(self.h, a.ctypes.data)
        return a

    def alive(self):
        n = self.L.shko_n_solid(self.h)
        a = np.zeros(n, dtype=np.uint8)
        if n:
            self.L.shko_get_alive(self.h, a.ctypes.data)
        return a

    def correct(self): self.L.shko_correct(self.h)
    def collapse(self): self.L.shko_collapse(self.h)

    def assemble(self):
        if self.L.shko_assemble(self.h) != 0:
            raise RuntimeError(self.L.shko_last_error(self.h).decode())

    @property
    def tips_removed(self): return self.L.shko_tips_removed(self.h)
    @property
    def bubbles_removed(self): return self.L.shko_bubbles_removed(self.h)

    def contigs(self):
        n = self.L.shko_n_contigs(self.h)
        return [self.L.shko_contig_seq(self.h, i).decode() for i in range(n)]

    def contig_kc(self):
        n = self.L.shko_n_contigs(self.h)
        return [self.L.shko_contig_kc(self.h, i) for i in range(n)]

    def fasta(self): return self.L.shko_fasta(self.h).decode()
    def gfa1(self): return self.L.shko_gfa1(self.h).decode()
    def gfa2(self): return self.L.shko_gfa2(self.h).decode()
    def dot(self): return self.L.shko_dot(self.h).decode()
    def assembly_json(self): return self.L.shko_assembly_json(self.h).decode()
    def preprocessing_json(self): return self.L.shko_preprocessing_json(self.h).decode()


# ---- multi-threaded CPU baseline (oracle/cpu_mt.cpp): bench.py's cpu_baseline leg and tests/test_cpu_mt.py ----
_MT = None


def _mt():
    global _MT
    if _MT is None:
        so = os.path.join(_HERE, "libshk_cpu_mt.so")
        src = os.path.join(_HERE, "cpu_mt.cpp")
        if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-C", _HERE, "libshk_cpu_mt.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        u64, u32, vp, cp = C.c_uint64, C.c_uint32, C.c_void_p, C.c_char_p
        for name, (res, args) in {
            "cpumt_hardware_threads": (C.c_uint, []), "cpumt_new": (vp, [u32, u32]), "cpumt_free": (None, [vp]),
            "cpumt_threads": (u32, [vp]), "cpumt_count": (None, [vp, vp, vp, u64, u32]),
            "cpumt_total_instances": (u64, [vp]), "cpumt_histo": (None, [vp, vp]), "cpumt_filter": (C.c_int, [vp, u32]),
            "cpumt_count_times": (None, [vp, vp, vp]), "cpumt_set_threads": (None, [vp, u32]),
            "cpumt_n_solid": (u64, [vp]), "cpumt_get_solid": (None, [vp, vp, vp]),
            "cpumt_assemble": (None, [vp, C.c_int, C.c_int]), "cpumt_n_contigs": (u64, [vp]), "cpumt_fasta": (cp, [vp]),
        }.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _MT = L
    return _MT


class CpuMt:
    """The multi-threaded CPU restatement (k <= 63): packed reads in, histogram / solid set / FASTA out."""

    def __init__(self, k, threads=0):
        self.L = _mt()
        self.k, self.W = k, (2 * k + 63) // 64
        self.h = self.L.cpumt_new(k, threads)
        if not self.h:
            raise ValueError("cpu_mt: k must be odd and within [15, 63]")
        self.threads = self.L.cpumt_threads(self.h)

    def __del__(self):
        try:
            if self.h:
                self.L.cpumt_free(self.h)
                self.h = None
        except Exception:
            pass

    @staticmethod
    def hardware_threads():
        return _mt().cpumt_hardware_threads()

    def count(self, bases, seg_off, emit_threshold=0):
        """bases: uint32 words (2-bit packed, the device layout), seg_off: uint32[n_seg + 1]"""
        b = np.ascontiguousarray(bases, dtype=np.uint32)
        s = np.ascontiguousarray(seg_off, dtype=np.uint32)
        self._keep = (b, s)
        self.L.cpumt_count(self.h, b.ctypes.data, s.ctypes.data, len(s) - 1, emit_threshold)

    @property
    def total_instances(self): return self.L.cpumt_total_instances(self.h)

    def set_threads(self, threads):
        self.L.cpumt_set_threads(self.h, threads)
        self.threads = self.L.cpumt_threads(self.h)

    def count_times(self):
        """seconds of the last count(): (scatter into partitions, per-partition tables)"""
        a, b = C.c_double(0), C.c_double(0)
        self.L.cpumt_count_times(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def histo(self):
        h = np.zeros(500, dtype=np.uint64)
        self.L.cpumt_histo(self.h, h.ctypes.data)
        return h

    def filter(self, threshold):
        if self.L.cpumt_filter(self.h, threshold) != 0:
            raise ValueError("threshold below the emit threshold")

    def solid(self):
        n = self.L.cpumt_n_solid(self.h)
        keys = np.zeros((n, self.W), dtype=np.uint64)
        cnt = np.zeros(n, dtype=np.uint32)
        if n:
            self.L.cpumt_get_solid(self.h, keys.ctypes.data, cnt.ctypes.data)
        return keys, cnt

    def assemble(self, no_bubble_collapse=False, no_dead_end_removal=False):
        self.L.cpumt_assemble(self.h, int(no_bubble_collapse), int(no_dead_end_removal))

    @property
    def n_contigs(self): return self.L.cpumt_n_contigs(self.h)

    def fasta(self): return self.L.cpumt_fasta(self.h).decode()
