"""The DEVICE inflater (csrc/inflate_gpu.hip, through shk_device_gunzip) against Python's zlib on the streams of
tools/fuzz_gunzip.py: members of 70 kB ... 20 MB made with random levels, strategies, windows, flush points and member counts
from FASTQ-like text, runs, binary data and mixtures, each also truncated and with single bits flipped.  The device inflater
either hands back exactly zlib's bytes or declines (the product then reads the member on the host, which owns the error
messages): it must NEVER return other bytes.  Needs a GPU.  Usage: python tools/fuzz_gunzip_device.py [cases] [seed]"""
import ctypes as C
import os
import sys
import time
import zlib
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SHK_GUNZIP_DEVICE_MIN", "32768")
import torch  # noqa: F401  (one HIP runtime)
import fuzz_gunzip as fg

L = fg.L
rng = fg.rng
n_cases = fg.n_cases


def device_gunzip(z):
    out, n, why, ms = C.c_void_p(), C.c_size_t(), C.c_char_p(), C.c_double()
    rc = L.shk_device_gunzip(z, len(z), C.byref(out), C.byref(n), C.byref(why), C.byref(ms))
    if rc == 0:
        got = C.string_at(out.value, n.value) if n.value else b""
        L.shk_host_free(out)
        return 0, got, "", ms.value
    return rc, None, (why.value or b"").decode(), ms.value


t0 = time.time()
taken, reasons, rates = 0, Counter(), []
for case in range(n_cases):
    n_members = int(rng.choice([1, 1, 1, 1, 2]))
    size = int(rng.choice([70000, 300_000, 3_000_000, 8_000_000, 20_000_000]))
    os.environ["SHK_GUNZIP_DEVICE_CHUNK"] = str(int(rng.choice([4096, 16384, 49152, 49152, 200000])))
    members, texts, descs = [], [], []
    for _ in range(n_members):
        t = fg.make_text(size)
        z, d = fg.compress(t)
        members.append(z); texts.append(t); descs.append(d)
    z, want = b"".join(members), b"".join(texts)
    desc = dict(case=case, members=n_members, size=size, zlen=len(z), chunk=os.environ["SHK_GUNZIP_DEVICE_CHUNK"], how=descs)
    try:
        rc, got, why, ms = device_gunzip(z)
        assert rc in (0, 1), ("error code", rc, why)
        if rc == 0:
            assert n_members == 1, "several members taken as one"
            assert got == want, ("intact stream: OTHER BYTES", len(got), len(want))
            taken += 1
            if len(want) >= 3_000_000:
                rates.append(len(want) / 1e9 / (ms * 1e-3))
        else:
            reasons[why] += 1
        # damaged streams: declined, or (harmless damage) zlib's bytes
        for cut in (int(rng.integers(1, len(z))), len(z) - 1, len(z) - 8):
            if 0 < cut < len(z):
                rc2, got2, why2, _ = device_gunzip(z[:cut])
                assert rc2 == 1 or (rc2 == 0 and got2 == fg.zlib_gunzip(z[:cut])), ("truncated at", cut, rc2)
        for _ in range(3):
            pos = int(rng.integers(2, len(z)))
            bad = bytearray(z); bad[pos] ^= 1 << int(rng.integers(0, 8)); bad = bytes(bad)
            try:
                ref = fg.zlib_gunzip(bad)
            except zlib.error:
                ref = None
            rc2, got2, why2, _ = device_gunzip(bad)
            assert rc2 == 1 or (rc2 == 0 and ref is not None and got2 == ref), ("corrupt byte at", pos, rc2, "zlib error" if ref is None else "zlib fine")
    except Exception as e:
        print("FAIL", desc, repr(e), flush=True)
        raise
    if case % 10 == 0:
        print("case", case, "ok  %.0f s" % (time.time() - t0), "taken so far", taken, flush=True)
print("all", n_cases, "cases: zlib's bytes, or declined; the device inflater took", taken, "intact members; declined:", dict(reasons),
      "; GB/s of text on the members >= 3 MB it took: min %.2f median %.2f max %.2f" % (min(rates), sorted(rates)[len(rates) // 2], max(rates)) if rates else "",
      "; %.0f s" % (time.time() - t0))
