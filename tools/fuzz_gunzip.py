"""Randomised differential test of the gzip reader (csrc/inflate_mt.cpp + the zlib path behind shk_host_gunzip) against
Python's zlib, on the CPU: members large enough for the multi-threaded inflater (>= 1 MiB of deflate data), made with random
levels, strategies, window sizes, flush points and member counts, from FASTQ-like text, text with long runs, binary data and
mixtures; every stream also truncated and with one byte flipped (must be an error or zlib's bytes, never a crash or other
bytes).  Usage: python tools/fuzz_gunzip.py [cases] [seed]"""
import ctypes as C
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sparrowhawk_amd import _lib, synth

L = _lib.load()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0)


def gunzip(z):
    out, n, mt = C.c_void_p(), C.c_size_t(), C.c_uint64()
    rc = L.shk_host_gunzip(z, len(z), C.byref(out), C.byref(n), C.byref(mt), None)
    if rc != 0:
        return rc, None, mt.value
    got = C.string_at(out.value, n.value) if n.value else b""
    L.shk_host_free(out)
    return 0, got, mt.value


def zlib_gunzip(z):
    """every member of the stream, as `zcat` would"""
    out, rest = [], z
    while rest:
        d = zlib.decompressobj(31)
        out.append(d.decompress(rest))
        if not d.eof:
            raise zlib.error("truncated")
        rest = d.unused_data
    return b"".join(out)


def fastq_text(nbytes):
    g = synth.random_genome(int(rng.integers(2000, 200000)), int(rng.integers(1 << 30)))
    rl = int(rng.choice([75, 100, 150, 251]))
    n = max(1, nbytes // (2 * rl + 12))
    codes, quals = synth.sample_reads(g, n, rl, int(rng.integers(1 << 30)), err=float(rng.choice([0, 0.01, 0.05])))
    return bytes(synth.to_fastq_fixed(codes, quals))


def make_text(nbytes):
    kind = rng.choice(["fastq", "fastq", "runs", "binary", "mixed", "lines"])
    if kind == "fastq":
        return fastq_text(nbytes)
    if kind == "runs":                                       # long matches at distance 1 .. 32768, length 258 runs
        parts, left = [], nbytes
        while left > 0:
            m = int(rng.integers(1, 100000))
            parts.append(bytes([int(rng.integers(32, 127))]) * m if rng.random() < 0.5 else rng.integers(65, 70, m, dtype=np.uint8).tobytes())
            left -= m
        return b"".join(parts)
    if kind == "binary":
        return rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes()
    if kind == "lines":
        words = [rng.integers(65, 91, int(rng.integers(1, 400)), dtype=np.uint8).tobytes() for _ in range(200)]
        idx = rng.integers(0, 200, nbytes // 100 + 1)
        return b"\n".join(words[i] for i in idx)
    return fastq_text(nbytes // 2) + rng.integers(0, 256, nbytes // 4, dtype=np.uint8).tobytes() + fastq_text(nbytes // 4)


def compress(text):
    level = int(rng.choice([0, 1, 2, 4, 6, 6, 9]))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    wbits = int(rng.choice([31, 31, 31, 25, 28]))            # gzip wrapper, windows of 32 KiB / 512 B / 4 KiB
    memlevel = int(rng.choice([8, 8, 1, 9]))
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    n_flush = int(rng.choice([0, 0, 1, 5, 40]))
    cuts = sorted(int(x) for x in rng.integers(0, len(text) + 1, n_flush))
    out, at = [], 0
    for c in cuts:
        out.append(co.compress(text[at:c]))
        out.append(co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))))
        at = c
    out.append(co.compress(text[at:]))
    out.append(co.flush())
    return b"".join(out), dict(level=level, strategy=strategy, wbits=wbits, memlevel=memlevel, flushes=n_flush)


if __name__ == "__main__":
    t0 = time.time()
    n_mt = 0
    for case in range(n_cases):
        n_members = int(rng.choice([1, 1, 1, 2, 3]))
        size = int(rng.choice([1, 100, 70000, 3_000_000, 8_000_000, 20_000_000]))
        members, texts, descs = [], [], []
        for _ in range(n_members):
            t = make_text(size) if size > 1 else (b"" if rng.random() < 0.5 else b"A")
            z, d = compress(t)
            members.append(z); texts.append(t); descs.append(d)
        z, want = b"".join(members), b"".join(texts)
        os.environ["SHK_GUNZIP_THREADS"] = str(int(rng.choice([1, 2, 3, 8])))
        desc = dict(case=case, members=n_members, size=size, zlen=len(z), threads=os.environ["SHK_GUNZIP_THREADS"], how=descs)
        try:
            assert zlib_gunzip(z) == want
            rc, got, mt_a = gunzip(z)
            assert rc == 0 and got == want, ("intact stream", rc, None if got is None else len(got), len(want))
            # truncated: an error (never a crash, never silently short)
            for cut in (int(rng.integers(1, len(z))), len(z) - 1, len(z) - 8, 10, 3):
                if 0 < cut < len(z):
                    rc, got, _ = gunzip(z[:cut])
                    if n_members > 1 and rc == 0:                # (a cut exactly between two members is a valid, shorter stream)
                        assert got == zlib_gunzip(z[:cut])
                    else:
                        assert rc != 0, ("truncated at", cut, "accepted")
            # one byte changed: an error, or — when the change is harmless (header fields zlib ignores too) — zlib's bytes
            for _ in range(3):
                pos = int(rng.integers(2, len(z)))              # (not the two magic bytes: without them the input is taken as plain text, by contract)
                bad = bytearray(z); bad[pos] ^= 1 << int(rng.integers(0, 8)); bad = bytes(bad)
                try:
                    ref = zlib_gunzip(bad)
                except zlib.error:
                    ref = None
                rc, got, _ = gunzip(bad)
                if ref is None:
                    assert rc != 0, ("corrupt byte at", pos, "accepted")
                else:
                    assert rc == 0 and got == ref, ("corrupt byte at", pos, "harmless for zlib", rc)
            rc, got, mt_b = gunzip(z)
            n_mt += mt_b - mt_a
        except Exception as e:
            print("FAIL", desc, repr(e), flush=True)
            raise
        if case % 10 == 0:
            print("case", case, "ok  %.0f s" % (time.time() - t0), desc, flush=True)
    print("all", n_cases, "cases: zlib's bytes or an error as zlib gives; the multi-threaded inflater took", n_mt, "members; %.0f s" % (time.time() - t0))
