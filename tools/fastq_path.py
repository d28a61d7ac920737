"""PCIe-inclusive rate of the drop-in entry point: shk_preprocess on FASTQ text held in host memory
(parse -> mask -> segment -> pack -> count -> filter), device parser vs host parser."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from sparrowhawk_amd import AssemblyHelper, synth
G, COV, L = int(os.environ.get("G", 5_000_000)), 100, 150
g = synth.random_genome(G, 0xEC02)
codes, quals = synth.sample_reads(g, G * COV // L, L, 0xEC02 + 1, err=0.005)
fq = synth.to_fastq_fixed(codes, quals)
nb = codes.size
print("FASTQ bytes %.2f GB, bases %.0f M" % (len(fq) / 1e9, nb / 1e6), flush=True)
del codes, quals
for name, env, bb in (("device, pipelined", "0", None), ("device, single shot", "0", "single"), ("device, batches", "0", str(1 << 27)), ("host parser", "1", None)):
    if os.environ.get("SKIP_HOST") == "1" and env == "1":
        continue
    if os.environ.get("ONLY_DEVICE") == "1" and bb:
        continue
    os.environ["SHK_HOST_PARSER"] = env
    os.environ.pop("SHK_BATCH_BASES", None); os.environ.pop("SHK_FASTQ_PIPELINE_MIN", None)
    if bb == "single":
        os.environ["SHK_FASTQ_PIPELINE_MIN"] = str(1 << 60); bb = None   # the whole text uploaded, then parsed
    if bb:
        os.environ["SHK_BATCH_BASES"] = bb               # 256 MB of text per piece
    for it in range(2 if env == "0" else 1):
        h = AssemblyHelper.new(31, False, 5, 20, 0, False, False, False, False)
        t0 = time.perf_counter()
        try:
            h.preprocess(fq)
        except Exception as e:
            print('preprocess:', e)
        dt = time.perf_counter() - t0
        t = h.timings()
        print("%-14s run %d: preprocess %.1f ms = %.2f Gbases/s   %s" % (name, it, dt * 1e3, nb / dt / 1e9,
              {k: round(v, 1) for k, v in t.items() if "host_clock" in k or "fastq" in k}), flush=True)
        n = h.n_solid
        if not os.environ.get("SHK_DEBUG_FQ"): h.assemble()
        h.free()
    print("   n_solid", n)
