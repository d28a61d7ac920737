"""Randomised differential test: the HIP path against the oracle on many small random datasets
(k, coverage, error rate, repeats, min_count, min_qual, flags, one or two files, gzip, chunking drawn at
random).  Not part of the pytest suite: run on a GPU box, e.g.  python tools/fuzz_parity.py 300 1"""
import gzip
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SHK_KEEP_STAGES", "1")               # quiet handles keep the initial adjacency too (compare_all reads it)
import numpy as np
import torch  # noqa: F401
from sparrowhawk_amd import AssemblyHelper, synth
from util import compare_all, run_oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
GLEN_LO, GLEN_HI = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (300, 30000)   # genome sizes
t0 = time.time()
for case in range(n_cases):
    k = int(rng.choice([15, 21, 27, 31, 33, 41, 51, 63, 65, 77, 89, 95, 101, 127, 129, 161, 191, 193, 225, 255]))   # one- to eight-word keys
    glen = int(rng.integers(GLEN_LO, GLEN_HI))
    g = synth.random_genome(glen, int(rng.integers(1 << 30)))
    if rng.random() < 0.4:                                   # planted repeats -> branching graph
        L = min(int(rng.integers(k + 5, 4 * k)), glen - 1)           # (a genome shorter than the repeat: the repeat shrinks)
        src = int(rng.integers(0, glen - L)); dst = int(rng.integers(0, glen - L))
        g[dst:dst + L] = g[src:src + L]
    if rng.random() < 0.2:                                   # inverted repeat / hairpin material
        L = min(int(rng.integers(k, 3 * k)), glen - 1); src = int(rng.integers(0, glen - L)); dst = int(rng.integers(0, glen - L))
        g[dst:dst + L] = (3 - g[src:src + L])[::-1]
    rl = int(rng.choice([max(k + 3, 60), 100, 150, 250, k + 120, 300, 700, 2300, 5500]))   # (round 4, >= 300: several tiles per round of pass 1, segments walked in pieces)
    rl = max(min(rl, glen - 1), k + 1)
    cov = float(rng.choice([3, 8, 20, 40]))
    err = float(rng.choice([0.0, 0.002, 0.01, 0.03]))
    circular = bool(rng.random() < 0.3)
    n_reads = max(1, int(glen * cov / rl))
    codes, quals = synth.sample_reads(g, n_reads, rl, int(rng.integers(1 << 30)), err=err, circular=circular)
    if rng.random() < 0.3:                                   # N bases
        m = rng.random(codes.shape) < 0.003
        fq = synth.to_fastq(codes, quals).decode()
        # put Ns into the text form
        lines = fq.split("\n")
        for i in range(1, len(lines), 4):
            row = m[(i - 1) // 4]
            if row.any():
                s = list(lines[i])
                for j in np.flatnonzero(row): s[j] = "N"
                lines[i] = "".join(s)
        fq = "\n".join(lines).encode()
    else:
        fq = synth.to_fastq(codes, quals)
    min_count = int(rng.choice([0, 1, 2, 3, 5]))
    min_qual = int(rng.choice([0, 11, 20, 33]))
    do_fit = bool(rng.random() < 0.25)
    do_bloom = bool(rng.random() < 0.1 and min_count >= 3)
    csize = int(rng.choice([0, 0, 500, 150000]))
    nb, nd = bool(rng.random() < 0.15), bool(rng.random() < 0.15)
    files = [fq]
    if rng.random() < 0.3:
        recs = fq.split(b"\n@r")
        half = max(1, len(recs) // 2)
        f1 = b"\n@r".join(recs[:half]) + b"\n"
        f2 = b"@r" + b"\n@r".join(recs[half:]) if len(recs) > half else None
        files = [f1] + ([f2] if f2 else [])
    sent = [gzip.compress(f) if rng.random() < 0.3 else f for f in files]
    env = {}
    if rng.random() < 0.3: env["SHK_HOST_PARSER"] = "1"
    if rng.random() < 0.2: env["SHK_BATCH_BASES"] = str(int(rng.integers(2000, 200000)))
    if rng.random() < 0.3: env["SHK_PART_P"] = str(int(rng.choice([2, 8, 64, 256, 16384])))   # few partitions: LDS tables overflow
    if rng.random() < 0.15: env["SHK_PROBE_PARTS"] = str(int(rng.choice([0, 1, 4, 8, 16])))     # (8, 16: the verdict "error-rich" after 1 or 2 partitions handed over)
    if rng.random() < 0.1: env["SHK_OVF_CAP_PCT"] = "60"            # bucket regions overflow: re-scatter / residue classes
    if rng.random() < 0.1: env["SHK_NO_REPARTITION"] = "1"
    if rng.random() < 0.35: env["SHK_SPLIT_LOG"] = str(int(rng.choice([0, 1, 3, 7, 10, 14])))   # rings with many / one / no splitter
    if rng.random() < 0.2: env["SHK_WRITER_PAR_MIN"] = "1"          # the writer's parallel paths
    if rng.random() < 0.3: env.update({"SHK_FASTQ_PIPELINE_MIN": "1", "SHK_FASTQ_PIECES": str(int(rng.choice([2, 3, 9])))})   # one batch parsed in pieces
    if rng.random() < 0.25: env["SHK_GP_ROWS"] = str(int(rng.choice([16, 64, 4096, 100000])))   # tiny graph partitions / partitions beyond the LDS table
    if rng.random() < 0.25: env["SHK_REGROUP_ROWS"] = str(int(rng.choice([0, 1])))   # rows moved into graph-partition order (or never)
    if rng.random() < 0.35: env["SHK_DEVICE_WRITER_MIN"] = "1"     # the get_assembly() text made on the device (csrc/writer_gpu.h)
    if rng.random() < 0.6: env["SHK_SHARD_DEDUPE"] = str(int(rng.choice([0, 1, 1])))   # sharded cases: records deduplicated by the sender (weights) or raw
    if rng.random() < 0.3: env["SHK_COUNT_SPLIT"] = "0"            # pass 2 fused (default: dedupe + weighted count as two kernels)
    if rng.random() < 0.3: env["SHK_COUNT_MERGE"] = str(int(rng.choice([1, 4])))   # partitions per table of k_count_weighted
    if rng.random() < 0.5: env["SHK_TILE_ROWS"] = str(int(rng.choice([1, 3, 17, 64, 300, 1000, 4096])))   # collapse: many small LDS tiles
    if rng.random() < 0.25: env["SHK_SEG_CAP"] = str(int(rng.choice([1, 8, 64])))   # round 4: the splitter list outgrows its room -> the ranking is called off and repeated
    if rng.random() < 0.3: env["SHK_DEVICE_PLAN"] = "1"            # round 4: emission planned on the device for <= 512 chain records (default: on the host)
    if rng.random() < 0.25: env["SHK_ARRIVAL_MIN"] = "1"           # round 4: the writer starts on contig text that is still arriving (slab-copy kernel + host flags)
    if rng.random() < 0.4: env["SHK_PART_G"] = str(int(rng.choice([1, 2, 5])))   # round 4: few workgroups -> every wave of pass 1 walks many tiles (prefetch of the next round)
    if rng.random() < 0.3: env["SHK_PART_WIN"] = str(int(rng.choice([16, 18, 20])))   # round 4: minimiser window of pass 1 for k >= 31 (one block of 16, two of 9, two of 10)
    if rng.random() < 0.3: env["SHK_GUNZIP_DEVICE_MIN"] = "2048"   # round 4: gzip members go to the device inflater first (it declines most of these tiny ones)
    old = {e: os.environ.get(e) for e in env}
    os.environ.update(env)
    desc = dict(case=case, k=k, glen=glen, rl=rl, cov=cov, err=err, circ=circular, mc=min_count, mq=min_qual, fit=do_fit,
                bloom=do_bloom, csize=csize, nb=nb, nd=nd, nfiles=len(files), env=env)
    sharded_case = False
    try:
        h = AssemblyHelper.new(k, True, min_count, min_qual, csize, do_bloom, do_fit, nb, nd)
        if len(files) == 1 and not do_bloom and csize == 0 and rng.random() < 0.3:
            # the sharded path with a one-rank RCCL communicator: shk_shard_preprocess, then the COLLECTIVE shk_assemble
            # (graph kept sharded, local chains stitched, tips / bubbles on the unitig graph: csrc/shard_graph.h)
            from sparrowhawk_amd import pack_fastq
            from sparrowhawk_amd.dist import LibComm, sharded_preprocess_rccl
            if "comm" not in globals():
                comm = LibComm(0, 1)
            bases, seg, nbases, nreads = pack_fastq(files[0], k, min_qual)
            dev = torch.device("cuda", 0)
            d_bases = torch.from_numpy(bases.view(np.int32)).to(dev); d_seg = torch.from_numpy(seg.view(np.int32)).to(dev)
            torch.cuda.synchronize()
            sharded_preprocess_rccl(h, d_bases.data_ptr(), d_seg.data_ptr(), len(seg) - 1, nbases, nreads, comm)
            sharded_case = True
            desc["sharded"] = True
        elif len(files) == 1 and rng.random() < 0.2:         # the packed-reads-in-host-memory entry point
            from sparrowhawk_amd import pack_fastq
            bases, seg, nbases, nreads = pack_fastq(files[0], k, min_qual)
            h.preprocess_packed_host(bases.ctypes.data, seg.ctypes.data, len(seg) - 1, nbases, nreads)
        elif len(files) == 1 and rng.random() < 0.2:         # packed reads already in HBM (bench.py's entry point), on a QUIET handle half of the time
            from sparrowhawk_amd import pack_fastq
            bases, seg, nbases, nreads = pack_fastq(files[0], k, min_qual)
            dev = torch.device("cuda", 0)
            d_bases = torch.from_numpy(bases.view(np.int32)).to(dev); d_seg = torch.from_numpy(seg.view(np.int32)).to(dev)
            torch.cuda.synchronize()
            if rng.random() < 0.5:
                h.free(); h = AssemblyHelper.new(k, False, min_count, min_qual, csize, do_bloom, do_fit, nb, nd)
            h.preprocess_packed_device(d_bases.data_ptr(), d_seg.data_ptr(), len(seg) - 1, nbases, nreads)
        else:
            h.preprocess(sent[0], sent[1] if len(sent) > 1 else None)
        h.assemble()
        o = run_oracle(files, k=k, min_count=min_count, min_qual=min_qual, do_fit=do_fit, no_bubble_collapse=nb,
                       no_dead_end_removal=nd)
        if do_bloom:
            # approximate by contract: every stored count is the true one or one more, nothing above the threshold is lost
            oe = run_oracle(files, k=k, min_count=0, min_qual=min_qual)
            true = {tuple(r): int(c) for r, c in zip(*[x.tolist() for x in oe.distinct()])}
            got = {tuple(r): int(c) for r, c in zip(*[x.tolist() for x in h.solid()])}
            thr = h.used_min_count
            assert all(key in true and c in (true[key], true[key] + 1) for key, c in got.items())
            assert {key for key, c in true.items() if c > thr} <= set(got)
            assert h.total_instances == o.total_instances
        else:
            compare_all(h, o, check_graph=not sharded_case)   # (sharded: tips / bubbles are settled on the unitig graph, per-node flags stay)
        h.free()
    except Exception as e:                                   # report and stop: a failing case is a bug
        print("FAIL", desc, repr(e), flush=True)
        raise
    finally:
        for e, v in old.items():
            if v is None: os.environ.pop(e, None)
            else: os.environ[e] = v
    if case % (20 if GLEN_HI <= 30000 else 2) == 0:
        print("case", case, "ok  %.0f s" % (time.time() - t0), desc, flush=True)
print("all", n_cases, "cases identical to the oracle in %.0f s" % (time.time() - t0))
