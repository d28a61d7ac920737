"""The device inflater on the bench isolate's .fastq.gz (1.05 GB of text in one member): stage times.  Needs a GPU."""
import ctypes as C, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparrowhawk_amd import _lib, synth
L = _lib.load()
dev = torch.device("cuda", 0)
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 3_333_334
g = torch.Generator(device=dev); g.manual_seed(0xEC02)
genome = torch.randint(0, 4, (5_000_000,), generator=g, device=dev, dtype=torch.int32)
parts, ar = [], torch.arange(150, device=dev)
for r0 in range(0, n_reads, 1 << 19):
    r1 = min(n_reads, r0 + (1 << 19))
    starts = torch.randint(0, 5_000_000 - 150 + 1, (r1 - r0,), generator=g, device=dev)
    codes = genome[starts[:, None] + ar[None, :]]
    parts.append(synth.device_fastq_fixed(torch, codes).cpu())
fq = torch.cat(parts).numpy().tobytes()
for level in (int(x) for x in os.environ.get('LEVELS', '1,6').split(',')):
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    gz = co.compress(fq) + co.flush()
    for chunk in (int(x) for x in os.environ.get('CHUNKS', '0,65536,131072').split(',')):
        if chunk:
            os.environ["SHK_GUNZIP_DEVICE_CHUNK"] = str(chunk)
        else:
            os.environ.pop("SHK_GUNZIP_DEVICE_CHUNK", None)
        for rep in range(2):
            out, n, why, ms = C.c_void_p(), C.c_size_t(), C.c_char_p(), C.c_double()
            t0 = time.perf_counter()
            rc = L.shk_device_gunzip(gz, len(gz), C.byref(out), C.byref(n), C.byref(why), C.byref(ms))
            dt = time.perf_counter() - t0
            ok = rc == 0 and C.string_at(out.value, n.value) == fq
            if rc == 0:
                L.shk_host_free(out)
        print("level %d, %.3f GB -> %.3f GB, chunk %d: rc %d (%s) equal %s, inflate %.1f ms = %.1f GB/s of text (call incl. download %.0f ms)"
              % (level, len(gz) / 1e9, len(fq) / 1e9, chunk, rc, (why.value or b"").decode(), ok, ms.value, len(fq) / 1e9 / (ms.value * 1e-3) if ms.value else 0, dt * 1e3), flush=True)
