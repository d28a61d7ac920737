"""AssemblyHelper — same names, argument order and call sequence as the reference's
wasm-bindgen struct (www/src/workers/Assembler.ts:15-39; driver sequence :73-139):

    helper = AssemblyHelper.new(k, verbose, min_count, min_qual, csize, do_bloom, do_fit,
                                no_bubble_collapse, no_dead_end_removal)
    helper.preprocess(file1, file2_or_None)
    info = json.loads(helper.get_preprocessing_info())   # {nkmers, histo[500], used_min_count}
    helper.assemble()
    out = json.loads(helper.get_assembly())               # {outfasta, ncontigs, outdot, outgfa, outgfav2}

Where the Rust crate panics (surfacing as a JS exception, Assembler.ts:93-106) this raises
ShkError carrying the C ABI error code.
"""
import ctypes as C
import json
import os

import numpy as np

from . import _lib

ERR_NAMES = {-1: "SHK_E_PARAM", -2: "SHK_E_STATE", -3: "SHK_E_PARSE", -4: "SHK_E_OOM",
             -5: "SHK_E_DEVICE", -6: "SHK_E_INTERNAL"}


class ShkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


def _as_bytes(f):
    if f is None:
        return None
    if isinstance(f, (bytes, bytearray, memoryview)):
        return bytes(f)
    if isinstance(f, (str, os.PathLike)):
        with open(f, "rb") as fh:
            return fh.read()
    if hasattr(f, "read"):
        return f.read()
    raise TypeError("file must be bytes, a path or a file object")


class AssemblyHelper:
    def __init__(self, handle, k):
        self._L = _lib.load()
        self._h = handle
        self.k = k
        self.states = []          # every progress string posted, in order (AssemblyPage.vue:458-609)
        self._cb = _lib.PROGRESS_CB(lambda s, _u: self.states.append(s.decode()))
        self._L.shk_set_progress_cb(self._h, self._cb, None)

    # static factory, not a constructor — exactly like the reference (Assembler.ts:94)
    @staticmethod
    def new(k, verbose, min_count, min_qual, csize, do_bloom, do_fit, no_bubble_collapse,
            no_dead_end_removal):
        L = _lib.load()
        h = L.shk_new(int(k), int(bool(verbose)), int(min_count), int(min_qual), int(csize),
                      int(bool(do_bloom)), int(bool(do_fit)), int(bool(no_bubble_collapse)),
                      int(bool(no_dead_end_removal)))
        if not h:
            raise ShkError(L.shk_new_error(), L.shk_new_error_message().decode())
        return AssemblyHelper(h, int(k))

    def free(self):
        if self._h:
            self._L.shk_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise ShkError(rc, self._L.shk_last_error(self._h).decode())

    # ---- the reference surface ------------------------------------------------------------
    def preprocess(self, file1, file2=None):
        b1, b2 = _as_bytes(file1), _as_bytes(file2)
        self._check(self._L.shk_preprocess(self._h, b1, len(b1), b2, len(b2) if b2 is not None else 0))

    def get_preprocessing_info(self):
        s = self._L.shk_get_preprocessing_info(self._h)
        if s is None:
            raise ShkError(-2, self._L.shk_last_error(self._h).decode())
        return s.decode()

    def assemble(self):
        self._check(self._L.shk_assemble(self._h))

    def get_assembly(self):
        s = self._L.shk_get_assembly(self._h)
        if s is None:
            raise ShkError(-2, self._L.shk_last_error(self._h).decode())
        return s.decode()

    # ---- streaming / device-resident forms (include/shk.h) ---------------------------------
    def push_reads(self, chunk):
        self._check(self._L.shk_push_reads(self._h, chunk, len(chunk)))

    def finish_reads(self):
        self._check(self._L.shk_finish_reads(self._h))

    def preprocess_packed_device(self, d_bases_ptr, d_seg_off_ptr, n_seg, n_bases, n_reads=0):
        self._check(self._L.shk_preprocess_packed_device(self._h, d_bases_ptr, d_seg_off_ptr, n_seg,
                                                         n_bases, n_reads))

    def preprocess_packed_host(self, bases_ptr, seg_off_ptr, n_seg, n_bases, n_reads=0):
        self._check(self._L.shk_preprocess_packed_host(self._h, bases_ptr, seg_off_ptr, n_seg, n_bases, n_reads))

    # ---- stage inspection -------------------------------------------------------------------
    @property
    def key_words(self): return self._L.shk_key_words(self._h)
    @property
    def total_instances(self): return self._L.shk_total_instances(self._h)
    @property
    def n_distinct(self): return self._L.shk_n_distinct(self._h)
    @property
    def n_solid(self): return self._L.shk_n_solid(self._h)
    @property
    def used_min_count(self): return self._L.shk_used_min_count(self._h)

    def _table(self, n, fn):
        W = self.key_words
        keys = np.zeros((n, W), dtype=np.uint64)
        cnt = np.zeros(n, dtype=np.uint32)
        self._check(fn(self._h, keys.ctypes.data, cnt.ctypes.data, n))
        return keys, cnt

    def distinct(self): return self._table(self.n_distinct, self._L.shk_get_distinct)
    def solid(self): return self._table(self.n_solid, self._L.shk_get_solid)

    def histo(self):
        h = np.zeros(500, dtype=np.uint64)
        self._check(self._L.shk_get_histo(self._h, h.ctypes.data))
        return h

    def adjacency(self):
        n = self.n_solid
        a0, a1, al = (np.zeros(n, dtype=np.uint8) for _ in range(3))
        self._check(self._L.shk_get_adjacency(self._h, a0.ctypes.data, a1.ctypes.data, al.ctypes.data, n))
        return a0, a1, al

    def timings(self):
        return json.loads(self._L.shk_get_timings(self._h).decode())

    @property
    def peak_device_bytes(self):
        """most device memory this handle held at once (the reference reports peak wasm memory: Assembler.ts:69-71,137)"""
        return int(self._L.shk_peak_device_bytes(self._h))


def pack_fastq(data, k, min_qual):
    """Host packer (SPEC S1-S2): returns (bases u32[], seg_off u32[], n_bases, n_reads)."""
    L = _lib.load()
    out = _lib.ShkPacked()
    err = C.c_char_p()
    rc = L.shk_pack_fastq(data, len(data), k, min_qual, C.byref(out), C.byref(err))
    if rc != 0:
        raise ShkError(rc, (err.value or b"").decode())
    try:
        nw = (out.n_bases >> 4) + 2
        bases = np.ctypeslib.as_array(out.bases, shape=(nw,)).copy()
        seg = np.ctypeslib.as_array(out.seg_off, shape=(out.n_seg + 1,)).copy()
        return bases, seg, int(out.n_bases), int(out.n_reads)
    finally:
        L.shk_packed_free(C.byref(out))
