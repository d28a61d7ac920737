#!/usr/bin/env python3
"""bench.py — Gbases/s assembled, k=31, 150 bp reads, on N MI355X (BASELINE.json metric).

A "step" is one pass of the whole hot path over one batch of synthetic reads that are already
2-bit packed and resident in HBM: shk_new -> shk_preprocess_packed_device (k-mer count ->
histogram -> filter) -> shk_assemble (graph -> correct -> collapse -> contigs + FASTA/GFA on the
host) -> shk_get_assembly.  Workload at N=1: BASELINE.json configs[1] — one 5 Mbp isolate,
100x coverage of 150 bp reads (3 333 334 reads, 500 Mbases), k=31, min_count=5.

Launch: python bench.py --gpus N --steps K --warmup W           (any N: for N > 1 without a launcher the script
                                                                 starts its own N ranks as child processes)
        python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N>1, under a launcher)
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_reads_on_device(torch, dev, genome_len, coverage, read_len, seed, read_seed=None, n_reads=None):
    """Synthetic isolate + error-free reads, generated and 2-bit packed on the GPU (SURVEY §8d cfg 2).
    Returns (d_bases int32[words], d_seg_off int32[n_reads+1], n_reads, n_bases, genome codes)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if n_reads is None:
        n_reads = (genome_len * coverage + read_len - 1) // read_len
    genome = torch.randint(0, 4, (genome_len,), generator=g, device=dev, dtype=torch.int32)
    if read_seed is not None:
        g.manual_seed(read_seed)
    n_bases = n_reads * read_len
    n_words = (n_bases + 15) // 16 + 1
    words = torch.zeros(n_words, dtype=torch.int32, device=dev)
    shifts = (2 * torch.arange(16, device=dev, dtype=torch.int32))
    chunk = 1 << 18                                     # reads per chunk (chunk*read_len % 16 == 0)
    ar = torch.arange(read_len, device=dev)
    for r0 in range(0, n_reads, chunk):
        r1 = min(n_reads, r0 + chunk)
        starts = torch.randint(0, genome_len - read_len + 1, (r1 - r0,), generator=g, device=dev)
        strand = torch.randint(0, 2, (r1 - r0,), generator=g, device=dev).bool()
        codes = genome[starts[:, None] + ar[None, :]]
        rc = (3 - codes).flip(1)
        codes = torch.where(strand[:, None], rc, codes).reshape(-1)
        pad = (-codes.numel()) % 16
        if pad:
            codes = torch.cat([codes, torch.zeros(pad, dtype=torch.int32, device=dev)])
        w = (codes.reshape(-1, 16) << shifts[None, :]).sum(dim=1, dtype=torch.int32)
        w0 = (r0 * read_len) // 16
        assert (r0 * read_len) % 16 == 0
        words[w0:w0 + w.numel()] = w
    seg_off = (torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * read_len).to(torch.int32)
    torch.cuda.synchronize()
    return words, seg_off, n_reads, n_bases, genome


def physical_cores():
    """(sockets x cores per socket) from /proc/cpuinfo; None when it cannot be read"""
    try:
        phys = set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
        return len(phys) or None
    except Exception:
        return None


def cpu_baseline(k, min_count, host_words, host_seg, n_bases, gpu_fasta):
    """The multi-threaded CPU restatement (oracle/cpu_mt.cpp: std::thread, two-pass radix scatter of the canonical
    k-mers into one arena, one open-addressing table per partition) timed on the FULL workload of this bench line —
    the same packed reads the GPU got — and cross-checked against the GPU's contigs.  The count is timed at several
    thread counts (SMT siblings do not always pay); the fastest one is the reported baseline.  Build's CPU
    restatement, not upstream sparrowhawk-asm (its source is absent from the reference)."""
    from oracle import CpuMt
    hw, phys = int(CpuMt.hardware_threads()), physical_cores()
    cand = sorted({t for t in (32, 64, 128, 256, phys or 0, hw) if 0 < t <= hw}) or [hw]
    m = CpuMt(k, cand[0])
    sweep, best = {}, None
    for t in cand:
        m.set_threads(t)
        t0 = time.perf_counter()
        m.count(host_words, host_seg, emit_threshold=min_count)
        dt = time.perf_counter() - t0
        sc, tb = m.count_times()
        sweep[str(t)] = {"count_s": round(dt, 3), "scatter_s": round(sc, 3), "tables_s": round(tb, 3),
                         "Mkmers_per_s": round(m.total_instances / dt / 1e6, 1)}
        if best is None or dt < best[1]:
            best = (t, dt)
    if m.threads != best[0]:                             # the rest runs on the rows of the fastest setting
        m.set_threads(best[0])
        m.count(host_words, host_seg, emit_threshold=min_count)
    t1 = time.perf_counter()
    m.filter(min_count)
    m.assemble()
    rest = time.perf_counter() - t1
    fa = m.fasta()
    total = best[1] + rest
    return {"value": n_bases / total / 1e9, "unit": "Gbases/s", "cores": int(best[0]), "kind": "port",
            "hardware_concurrency": hw, "physical_cores": phys,
            "contigs_equal_gpu": bool(fa == gpu_fasta),
            "count_Gkmers_per_s": m.total_instances / best[1] / 1e9,
            "threads_sweep": sweep,
            "sample": f"the full workload ({n_bases / 1e6:.0f} Mbases, the same packed reads), count {best[1]:.2f} s + "
                      f"filter/graph/correct/collapse {rest:.2f} s on {best[0]} threads (the fastest of {cand}); multi-threaded "
                      f"restatement (oracle/cpu_mt.cpp), checked against the single-threaded oracle in tests/test_cpu_mt.py; "
                      f"build's CPU restatement, not upstream sparrowhawk-asm"}


def run_legs(torch, dev, args, raw_get_assembly):
    """BASELINE configs[2] and the FASTQ entry point, each timed like the headline (warm-up, then `--leg-steps` steps
    bracketed by device synchronisation).  Gbases/s counts the bases of the READS (3 333 334 x 150) in every leg."""
    import ctypes
    from sparrowhawk_amd import AssemblyHelper, synth
    legs = {}
    n_reads = (args.genome * args.coverage + args.read_len - 1) // args.read_len
    input_bases = n_reads * args.read_len

    def timed(step, steps):
        step()                                            # warm-up (allocations of this shape come from the pool afterwards)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ts = [step() for _ in range(steps)]
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, ts

    def packed_leg(name, k, dr, bloom, note):
        W = (2 * k + 63) // 64
        state = {}

        def step():
            h = AssemblyHelper.new(k, False, args.min_count, 20, 0, bloom, False, False, False)
            h.preprocess_packed_device(dr.words.data_ptr(), dr.seg_off.data_ptr(), dr.n_seg, dr.n_bases, dr.n_reads)
            h.assemble()
            ptr = raw_get_assembly(h._h)
            assert ptr
            if "ncontigs" not in state:
                state["ncontigs"] = json.loads(ctypes.string_at(ptr))["ncontigs"]
                state["n_solid"], state["n_distinct"] = h.n_solid, h.n_distinct
            t = h.timings()
            state["peak_device_bytes"] = int(t.get("peak_device_bytes", 0))
            h.free()
            return t
        dt, ts = timed(step, args.leg_steps)
        c_ms = sum(t.get("count_kernel", 0.0) for t in ts) / len(ts)
        p_ms = sum(t.get("partition_kernel", 0.0) for t in ts) / len(ts)
        alg = dr.n_bases * 0.25 + dr.n_seg * 4 + state["n_distinct"] * (8 * W + 4)
        dom, k_ms = ("k_partition", p_ms) if p_ms > c_ms else ("count (k_count_partitions + k_ovf_scatter + k_count_buckets)", c_ms)
        legs[name] = {
            "workload": note, "value": input_bases / dt / 1e9, "unit": "Gbases/s", "clock": "device-resident", "ms_per_step": dt * 1e3, "steps": args.leg_steps,
            "peak_device_bytes": state.get("peak_device_bytes"),
            "k": k, "segments": int(dr.n_seg), "bases_on_device": int(dr.n_bases), "n_distinct_kmers": state["n_distinct"],
            "n_solid_kmers": state["n_solid"], "ncontigs": state["ncontigs"],
            "roofline": {"bound": "hbm", "kernel": dom, "kernel_ms": k_ms, "count_step_ms": c_ms + p_ms,
                         "algorithmic_bytes_per_launch": alg, "bytes_per_base": alg / max(1, dr.n_bases),
                         "achieved": alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms > 0 else 0.0,
                         "count_step_frac": (alg / ((c_ms + p_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS) if c_ms + p_ms > 0 else 0.0},
            "stage_ms": {kk: sum(t.get(kk, 0.0) for t in ts) / len(ts) for kk in sorted(ts[-1]) if not kk.endswith("_x1")},
        }

    base = f"{args.genome} bp isolate, {args.coverage}x {args.read_len} bp reads ({n_reads} reads), "
    dr = synth.device_reads(torch, dev, args.genome, n_reads, args.read_len, 51, 0xEC03, err=0.01, mask_errors=True)
    packed_leg("config2_masked", 51, dr, False, base + "k=51, 1 %% substitution errors masked by quality (min_qual cuts the reads into "
               "error-free segments before the device entry point), min_count=%d" % args.min_count)
    del dr
    dr = synth.device_reads(torch, dev, args.genome, n_reads, args.read_len, 51, 0xEC03, err=0.01, mask_errors=False)
    packed_leg("config2_errors_left_in", 51, dr, False, base + "k=51, 1 %% substitution errors LEFT IN (every partition goes through the "
               "k-mer-level repartition), min_count=%d" % args.min_count)
    packed_leg("config2_errors_left_in_bloom", 51, dr, True, base + "k=51, 1 %% errors left in, do_bloom (Bloom pre-filter in front of the "
               "repartition: singletons never stored), min_count=%d" % args.min_count)
    del dr
    torch.cuda.empty_cache()
    # FASTQ text in host memory -> shk_preprocess (device parser, text uploaded in pieces under the parse) -> contigs
    g = torch.Generator(device=dev); g.manual_seed(0xEC02)
    genome = torch.randint(0, 4, (args.genome,), generator=g, device=dev, dtype=torch.int32)
    parts = []
    ar = torch.arange(args.read_len, device=dev)
    for r0 in range(0, n_reads, 1 << 19):
        r1 = min(n_reads, r0 + (1 << 19))
        starts = torch.randint(0, args.genome - args.read_len + 1, (r1 - r0,), generator=g, device=dev)
        strand = torch.randint(0, 2, (r1 - r0,), generator=g, device=dev).bool()
        codes = genome[starts[:, None] + ar[None, :]]
        codes = torch.where(strand[:, None], (3 - codes).flip(1), codes)
        parts.append(synth.device_fastq_fixed(torch, codes).cpu())
    fq = torch.cat(parts).numpy().tobytes()
    del parts, genome
    state = {}

    def fq_step():
        h = AssemblyHelper.new(args.k, False, args.min_count, 20, 0, False, False, False, False)
        h.preprocess(fq)
        h.assemble()
        ptr = raw_get_assembly(h._h)
        assert ptr
        if "ncontigs" not in state:
            state["ncontigs"] = json.loads(ctypes.string_at(ptr))["ncontigs"]
        t = h.timings()
        h.free()
        return t
    dt, ts = timed(fq_step, args.leg_steps)
    # the reference's real input: the same text as ONE gzip member (level 1 here: the compressor is Python's, outside the clock)
    import zlib
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    gz = co.compress(fq) + co.flush()
    fq_plain, fq = fq, gz
    state.clear()
    dtz, tsz = timed(fq_step, args.leg_steps)
    fq = fq_plain
    legs["fastq_gz"] = {
        "workload": base + f"k={args.k}, error-free, as a single-member .fastq.gz of {len(gz) / 1e9:.2f} GB ({len(fq) / 1e9:.2f} GB of text) in host "
                    "memory -> shk_preprocess (the compressed bytes cross PCIe, inflated on the device: csrc/inflate_gpu.hip, then the device parser) -> contigs; PCIe-inclusive",
        "value": input_bases / dtz / 1e9, "unit": "Gbases/s", "clock": "host text (PCIe-inclusive)", "ms_per_step": dtz * 1e3, "steps": args.leg_steps, "ncontigs": state["ncontigs"],
        # (since round 4 the member is inflated on the device: csrc/inflate_gpu.hip; "host" only when it declined)
        "gunzip_where": "device" if all(t.get("gunzip_device_members_x1", 0) for t in tsz) else "host",
        "gunzip_GB_per_s_of_text": len(fq) / 1e9 / max(1e-9, (sum(t.get("gunzip_device_host_clock", 0.0) or t.get("gunzip_host_clock", 0.0) for t in tsz) / len(tsz) * 1e-3)),
        "stage_ms": {kk: sum(t.get(kk, 0.0) for t in tsz) / len(tsz) for kk in sorted(tsz[-1]) if not kk.endswith("_x1")},
    }
    legs["fastq_text"] = {
        "workload": base + f"k={args.k}, error-free, as {len(fq) / 1e9:.2f} GB of FASTQ text in host (pageable) memory -> shk_preprocess "
                    "(device parser, upload in pieces under the parse) -> shk_assemble -> contigs on the host; PCIe-inclusive",
        "value": input_bases / dt / 1e9, "unit": "Gbases/s", "clock": "host text (PCIe-inclusive)", "ms_per_step": dt * 1e3, "steps": args.leg_steps, "ncontigs": state["ncontigs"],
        "stage_ms": {kk: sum(t.get(kk, 0.0) for t in ts) / len(ts) for kk in sorted(ts[-1]) if not kk.endswith("_x1")},
    }
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--coverage", type=int, default=100)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--min-count", type=int, default=5)
    ap.add_argument("--err", type=float, default=0.0, help="substitution error rate of the reads (configs[2]: 0.01 with --k 51)")
    ap.add_argument("--mask-errors", action="store_true", help="erroneous bases carry a low quality and are masked "
                    "(min_qual): reads are cut into error-free segments before they reach the device entry point")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the second timed leg (packed reads in host pinned memory)")
    ap.add_argument("--no-inflight-leg", action="store_true", help="skip the leg with two handles in flight")
    ap.add_argument("--no-legs", action="store_true", help="skip the configs[2] and FASTQ-text legs")
    ap.add_argument("--leg-steps", type=int, default=3)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the N>1 code path on a one-GPU box together with --one-gpu)")
    ap.add_argument("--one-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--mode", choices=["isolates", "sharded"], default="sharded",
                    help="N>1: 'sharded' (default, the north_star path) = ONE pooled sample of N isolates, every rank "
                         "holds an equal share of its reads, k-mer space partitioned across ranks with one RCCL "
                         "pairwise exchange inside libshk_hip.so (shk_shard_preprocess); 'isolates' = every rank "
                         "assembles its own isolate (independent objects, no data-path collective) — the comparison point")
    ap.add_argument("--force-sharded", action="store_true", help="N=1: run the sharded path with a one-rank RCCL communicator "
                    "(what the exchange machinery costs when nothing has to leave the GPU)")
    ap.add_argument("--collectives", choices=["lib", "torch"], default="lib",
                    help="sharded mode: 'lib' = RCCL inside the library (production); 'torch' = the same shk_shard_* "
                         "pieces driven by torch.distributed collectives (rehearsal with --backend gloo --one-gpu)")
    args = ap.parse_args()
    # ---- N > 1 without a launcher: `python bench.py --gpus N` starts its own N ranks — as FRESH child processes of
    # torch.distributed.run, before this process has imported torch or touched the GPU (a process that has initialised
    # the GPU must never exec or be replaced) — hands their stdout through (rank 0's ONE JSON line) and exits with
    # their code.  Under a launcher (WORLD_SIZE set by torch.distributed.run) this is skipped.
    if args.gpus > 1 and "RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes (see the task's environment notes)
        env.setdefault("OMP_NUM_THREADS", "4")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    # Everything that anything prints to stdout from here on (RCCL announces its version there when a communicator is
    # created) goes to stderr: the contract is ONE JSON line on stdout, written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch                                         # before libshk_hip.so: one HIP runtime
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if args.one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from sparrowhawk_amd import AssemblyHelper, _lib
    import ctypes
    L = _lib.load()
    raw_get_assembly = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p)(("shk_get_assembly", L))

    sharded = (world > 1 or args.force_sharded) and args.mode == "sharded"
    lib_error = None
    if sharded:
        # one pooled sample: N isolates, every rank holds an equal share of reads drawn from all of them
        from sparrowhawk_amd.dist import Comm, LibComm, sharded_preprocess, sharded_preprocess_rccl
        comm, lib_error = None, None
        if args.collectives == "lib":
            try:
                comm = LibComm(rank, world)
            except Exception as e:                       # (e.g. an RCCL that refuses the topology)
                lib_error = repr(e)
            if world > 1:
                # every rank must take the same path: agree on whether ALL communicators came up
                flag = torch.tensor([0 if comm is not None else 1], device=dev if args.backend == "nccl" else "cpu")
                dist.all_reduce(flag)
                if int(flag.item()) and comm is not None:
                    comm.free(); comm = None
                    lib_error = "another rank could not create the library's communicator"
            if comm is None:
                # LOUD second path, recorded in the JSON line: the same shk_shard_* pieces, collectives by torch.distributed
                print(f"[bench rank {rank}] libshk's RCCL communicator failed ({lib_error}); collectives by torch.distributed",
                      file=sys.stderr, flush=True)
                if world == 1:
                    raise SystemExit("the library's RCCL communicator could not be created: " + str(lib_error))
                args.collectives = "torch"
        if comm is None:
            comm = Comm(device=dev)
        d_bases, d_seg, n_reads, n_bases, genome = make_reads_on_device(
            torch, dev, args.genome * world, args.coverage, args.read_len, 0xEC02, read_seed=0x5EED + rank,
            n_reads=(args.genome * args.coverage + args.read_len - 1) // args.read_len)
    elif args.err > 0:
        from sparrowhawk_amd import synth
        dr = synth.device_reads(torch, dev, args.genome, (args.genome * args.coverage + args.read_len - 1) // args.read_len,
                                args.read_len, args.k, 0xEC03 + rank, err=args.err, mask_errors=args.mask_errors)
        d_bases, d_seg, n_reads, n_bases, genome = dr.words, dr.seg_off, dr.n_seg, dr.n_bases, dr.genome
    else:
        # every rank owns one isolate of the batch (independent objects: no data-path collective)
        d_bases, d_seg, n_reads, n_bases, genome = make_reads_on_device(
            torch, dev, args.genome, args.coverage, args.read_len, 0xEC02 + rank)

    call_times = {} if os.environ.get("BENCH_CALL_TIMES") else None     # host time per C call, printed to stderr at the end

    def one_step(keep=False):
        c0 = time.perf_counter()
        h = AssemblyHelper.new(args.k, False, args.min_count, 20, 0, False, False, False, False)
        c1 = time.perf_counter()
        if sharded and args.collectives == "torch":
            sharded_preprocess(h, d_bases, d_seg, n_reads, n_bases, n_reads, comm)
        elif sharded:
            sharded_preprocess_rccl(h, d_bases.data_ptr(), d_seg.data_ptr(), n_reads, n_bases, n_reads, comm)
        else:
            h.preprocess_packed_device(d_bases.data_ptr(), d_seg.data_ptr(), n_reads, n_bases, n_reads)
        c2 = time.perf_counter()
        h.assemble()
        c3 = time.perf_counter()
        # the JSON (contigs as FASTA/GFA/DOT) is on the host now; take the pointer without making
        # a Python copy of ~15 MB inside the timed region (copied once, after timing, for checking)
        ptr = raw_get_assembly(h._h)
        assert ptr
        out = ctypes.string_at(ptr) if keep else None
        t = h._L.shk_get_timings(h._h)                   # (the JSON text; parsed behind the timed region: json.loads is 15-20 us of host time per step)
        info = (h.n_solid, h.n_distinct)
        c4 = time.perf_counter()
        h.free()
        c5 = time.perf_counter()
        if call_times is not None:
            for name, dt_ in (("new", c1 - c0), ("preprocess", c2 - c1), ("assemble", c3 - c2), ("get_assembly+timings", c4 - c3), ("free", c5 - c4)):
                call_times.setdefault(name, []).append(dt_ * 1e3)
        return out, t, info

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- N > 1: one untimed trial step decides the path EVERY rank measures.  The library's collectives report a failure on
    # every rank (api.cpp: agree), so all ranks see the exception; the flag is summed anyway, so that a rank-local failure moves
    # everybody too.  Order: RCCL inside the library -> the same shk_shard_* pieces with torch.distributed's collectives -> one
    # isolate per rank (no data-path collective).  Every step down is LOUD (stderr) and recorded in the JSON line ("fallbacks").
    fallbacks = []
    stepped_down_to_replicas = False
    inject = os.environ.get("BENCH_TEST_FAIL_PATHS", "").split(",")     # (tests/test_dist.py: the steps down, rehearsed)
    while sharded and world > 1:
        trial_error = None
        try:
            if args.collectives in inject:
                raise RuntimeError("injected failure (BENCH_TEST_FAIL_PATHS)")
            one_step()
        except Exception as e:
            trial_error = repr(e)
        flag = torch.tensor([1 if trial_error else 0], device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag)
        if int(flag.item()) == 0:
            break
        path = "lib" if args.collectives == "lib" else "torch"
        fallbacks.append({"path": "sharded, collectives=" + path, "error": trial_error or "another rank failed"})
        print(f"[bench rank {rank}] the sharded step failed with collectives={path}: {trial_error or 'on another rank'}",
              file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        if args.collectives == "lib":
            comm.free()
            comm = Comm(device=dev)
            args.collectives = "torch"
        else:
            sharded = False                              # every rank assembles an isolate of its own
            stepped_down_to_replicas = True
            d_bases, d_seg, n_reads, n_bases, genome = make_reads_on_device(
                torch, dev, args.genome, args.coverage, args.read_len, 0xEC02 + rank)
    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    kern_ms, all_t = [], []
    raw_t = []
    for _ in range(args.steps):
        _, t, info = one_step()
        raw_t.append(t)
    barrier()
    dt = time.perf_counter() - t0
    for t in raw_t:
        t = json.loads(t.decode())
        kern_ms.append(t.get("count_kernel", 0.0))
        all_t.append(t)
    if call_times is not None:
        print("[bench] host ms per call (median of the timed steps):",
              {k_: round(sorted(v[args.warmup:args.warmup + args.steps])[len(v[args.warmup:args.warmup + args.steps]) // 2], 3) for k_, v in call_times.items()},
              file=sys.stderr, flush=True)
    out, _, _ = one_step(keep=True)                      # untimed: fetch the result for checking
    # ---- comparison point (N > 1, sharded): the same ranks, every one assembling an isolate of its own — independent
    # objects, no data-path collective.  Timed the same way, reported beside `value`, never as it.
    iso_leg = None
    if sharded and world > 1 and args.err == 0:
        ib, iseg, inr, inb, _g = make_reads_on_device(torch, dev, args.genome, args.coverage, args.read_len, 0xEC02 + rank)

        def iso_step():
            h = AssemblyHelper.new(args.k, False, args.min_count, 20, 0, False, False, False, False)
            h.preprocess_packed_device(ib.data_ptr(), iseg.data_ptr(), inr, inb, inr)
            h.assemble()
            assert raw_get_assembly(h._h)
            h.free()
        for _ in range(max(1, args.warmup)):
            iso_step()
        barrier()
        ti = time.perf_counter()
        for _ in range(args.steps):
            iso_step()
        barrier()
        iso_dt = time.perf_counter() - ti
        tt = torch.tensor([iso_dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        iso_leg = float(tt.item()) / args.steps
        del ib, iseg
    # ---- two handles in flight (N = 1): a batch of isolates keeps the GPU busy while a handle's host side reads counters
    # back and writes the FASTA / GFA text — each handle on its own host thread and stream (sparrowhawk_amd/batch.py does
    # the same).  Reported beside `value` (which stays one handle at a time: its kernel timings are undisturbed).
    inflight_leg = None
    if world == 1 and not sharded and not args.no_inflight_leg:
        import threading
        n_fl, todo_lock, todo = 2, threading.Lock(), [0]

        def fl_worker(n_total):
            while True:
                with todo_lock:
                    if todo[0] >= n_total:
                        return
                    todo[0] += 1
                one_step()
        for n_total in (max(2, args.warmup), args.steps):          # warm-up (second stream, second set of pooled blocks), then timed
            todo[0] = 0
            torch.cuda.synchronize()
            tf = time.perf_counter()
            ths = [threading.Thread(target=fl_worker, args=(n_total,)) for _ in range(n_fl)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize()
            inflight_leg = (time.perf_counter() - tf) / n_total
    # ---- the other driver-timed legs (N = 1): BASELINE configs[2] (k = 51, 1 % errors: masked by quality, left in, and
    # left in with the Bloom pre-filter) and the drop-in entry point on FASTQ text in host memory
    legs = None
    if world == 1 and not sharded and args.err == 0 and not args.no_legs and args.genome == 5_000_000:
        legs = run_legs(torch, dev, args, raw_get_assembly)
    # ---- second leg (N = 1): the clock of SURVEY.md 8(d) — packed reads resident in host PINNED memory -> contig
    # strings on the host; the upload rides in front of pass 1 on the library's stream
    host_leg = None
    if world == 1 and args.err == 0 and not args.no_host_leg and not sharded:
        hw = torch.empty(d_bases.numel(), dtype=torch.int32).pin_memory()
        hs = torch.empty(d_seg.numel(), dtype=torch.int32).pin_memory()
        hw.copy_(d_bases); hs.copy_(d_seg)
        torch.cuda.synchronize()

        def host_step():
            h = AssemblyHelper.new(args.k, False, args.min_count, 20, 0, False, False, False, False)
            h.preprocess_packed_host(hw.data_ptr(), hs.data_ptr(), n_reads, n_bases, n_reads)
            h.assemble()
            assert raw_get_assembly(h._h)
            h.free()
        for _ in range(max(1, args.warmup)):
            host_step()
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(args.steps):
            host_step()
        torch.cuda.synchronize()
        host_leg = (time.perf_counter() - th) / args.steps
    # ---- the SURVEY 8(d) clock with two handles in flight: both fed from host pinned memory, so the upload of isolate i+1
    # rides under the kernels of isolate i (a batch of isolates on one GPU)
    host_inflight_leg = None
    if host_leg is not None and not args.no_inflight_leg:
        import threading
        todo_lock2, todo2 = threading.Lock(), [0]

        def hfl_worker(n_total):
            while True:
                with todo_lock2:
                    if todo2[0] >= n_total:
                        return
                    todo2[0] += 1
                host_step()
        for n_total in (max(2, args.warmup), args.steps):
            todo2[0] = 0
            torch.cuda.synchronize()
            tf = time.perf_counter()
            ths = [threading.Thread(target=hfl_worker, args=(n_total,)) for _ in range(2)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize()
            host_inflight_leg = (time.perf_counter() - tf) / n_total
    # ---- the shard layer with a one-rank communicator (N = 1): what the exchange machinery and the collective shk_assemble
    # cost when nothing has to leave the GPU — driver-timed beside the plain path
    sharded_one_rank = None
    if world == 1 and not sharded and args.err == 0 and not args.no_legs:
        try:
            from sparrowhawk_amd.dist import LibComm, sharded_preprocess_rccl
            comm1 = LibComm(0, 1)
            st1 = {}

            def sh_step():
                h = AssemblyHelper.new(args.k, False, args.min_count, 20, 0, False, False, False, False)
                sharded_preprocess_rccl(h, d_bases.data_ptr(), d_seg.data_ptr(), n_reads, n_bases, n_reads, comm1)
                h.assemble()
                assert raw_get_assembly(h._h)
                st1["t"] = h.timings()
                h.free()
            for _ in range(max(1, args.warmup)):
                sh_step()
            torch.cuda.synchronize()
            ts1 = time.perf_counter()
            for _ in range(args.leg_steps):
                sh_step()
            torch.cuda.synchronize()
            dts = (time.perf_counter() - ts1) / args.leg_steps
            sharded_one_rank = {"value": n_bases / dts / 1e9, "unit": "Gbases/s", "clock": "device-resident", "ms_per_step": dts * 1e3,
                                "steps": args.leg_steps, "host_waits_per_assemble": st1["t"].get("shard_host_waits_x1"),
                                "workload": "the headline isolate through shk_shard_preprocess + the collective shk_assemble on a one-rank RCCL "
                                            "communicator (records deduplicated, packed, exchanged with itself; graph phases on the sharded path)",
                                "stage_ms": {kk: vv for kk, vv in sorted(st1["t"].items()) if kk.startswith("shard_")}}
            comm1.free()
        except Exception as e:
            sharded_one_rank = {"error": repr(e)}
    if world > 1:
        tt = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # sanity on the result of the last step: one contig that is a substring of the genome
    res = json.loads(out)
    ncontigs = res["ncontigs"]
    # closed-form check of the last result: error-free reads of a repeat-free isolate
    # give one contig that is a substring of the genome (up to strand), ends trimmed by the filter
    if args.coverage >= 30 and ncontigs == 1 and not sharded and args.err == 0:
        contig = res["outfasta"].split("\n")[1]
        gs = "".join("ACGT"[int(c)] for c in genome.cpu().tolist()) if args.genome <= 20_000_000 else None
        if gs is not None:
            rc = contig[::-1].translate(str.maketrans("ACGT", "TGCA"))
            assert contig in gs or rc in gs, "contig is not a substring of the genome"
            assert len(contig) > args.genome - 400

    total_bases = n_bases * world
    value = total_bases * args.steps / dt / 1e9
    W = (2 * args.k + 63) // 64
    n_solid, n_distinct = info
    alg_bytes = n_bases * 0.25 + n_reads * 4 + n_distinct * (8 * W + 4)
    c_ms = sum(kern_ms) / max(1, len(kern_ms))
    p_ms = sum(t.get("partition_kernel", 0.0) for t in all_t) / max(1, len(all_t))
    # the dominant kernel of the k-mer-count step (two kernels: partition, count)
    # (pass 2 of one- and two-word keys is three kernels since round 3 — a sample of partitions through k_count_partitions, then
    # k_dedupe_partitions + k_count_weighted — timed together as `count_kernel`; traffic.json knows kernels, so the measured
    # fraction is only given when the dominant one is a single kernel)
    split_pass2 = any("count_dedupe_kernel" in t for t in all_t)
    dom, k_ms = ("k_partition", p_ms) if p_ms > c_ms else (("pass 2 (k_count_partitions sample + k_dedupe_partitions + k_count_weighted)" if split_pass2 else "k_count_partitions"), c_ms)
    traffic, traffic_src, traffic_all = None, None, {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        traffic_all = tj["bytes_per_launch"]
        traffic, traffic_src = traffic_all.get(dom), tj["source"]
    except Exception:
        pass
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    # the second denominator SURVEY.md 8(d) asks for: what a pure streaming read of 2 GiB reaches on this box
    stream_gbs = None
    if rank == 0:
        g = ctypes.c_double(0.0)
        if L.shk_measure_stream_read(2 << 30, 5, ctypes.byref(g)) == 0:
            stream_gbs = g.value
    line = {
        "metric": "Gbases/s assembled, k=31 150bp reads",
        "value": value, "unit": "Gbases/s", "clock": "device-resident", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        # ("replicas": the sharded paths failed on this node and every rank assembled an isolate of its own — see `fallbacks`)
        "scaling": "replicas" if stepped_down_to_replicas else "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": (f"one pooled sample of {args.genome * world} bp, " if sharded else "") +
                               f"{args.genome} bp isolate per GPU, {args.coverage}x {args.read_len} bp reads "
                               f"({n_reads} reads, {n_bases} bases per GPU), k={args.k}, min_count={args.min_count}, "
                               f"{'error-free' if args.err == 0 else ('%g substitution errors%s' % (args.err, ', masked by quality' if args.mask_errors else ''))}, packed 2-bit in HBM",
                   "parallelism": ("single GPU" if world == 1 else
                                   ("one pooled sample, k-mer space sharded by minimiser partition, one RCCL pairwise exchange of the "
                                    "records inside the library (shk_shard_preprocess), " +
                                    ("solid set gathered, graph phases replicated on every rank (SHK_SHARD_GRAPH=0)" if os.environ.get("SHK_SHARD_GRAPH") == "0"
                                     else "graph phases sharded too (collective shk_assemble: neighbour queries, half links, stitched chains, "
                                          "unitig-level correction, per-rank emission)") if args.collectives == "lib"
                                    else "one pooled sample, sharded, collectives by torch.distributed (" + args.backend + ")" +
                                         ("" if not lib_error else " — SECOND PATH: the library's RCCL communicator failed: " + str(lib_error)) +
                                         ("" if not fallbacks else " — SECOND PATH: " + fallbacks[0]["error"]))
                                   if sharded else "one isolate per rank (batch of isolates), no data-path collective" +
                                   ("" if not fallbacks else " — FALLBACK: the sharded path failed on this node, see fallbacks")),
                   "ncontigs": ncontigs, "n_distinct_kmers": n_distinct, "n_solid_kmers": n_solid,
                   "peak_device_bytes": int(max(t.get("peak_device_bytes", 0) for t in all_t)),
                   "clock": "`value` starts with the packed reads resident in HBM (the bench contract); the SURVEY 8(d) clock — packed reads "
                            "in host pinned memory -> contig strings on the host — is `host_pinned.value` / `value_host_pinned`"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     # what SURVEY.md 8(d) asks to report beside the algorithmic fraction: the kernel's MEASURED fabric
                     # traffic (rocprofv3 PMC, profiles/traffic.json) over its live duration, against the same peak
                     "measured_frac": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and k_ms > 0) else None,
                     "measured_frac_other_count_kernel": (lambda o, ms: (traffic_all[o] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                                          if (o in traffic_all and ms > 0) else None)(
                         "k_partition" if dom != "k_partition" else "k_count_partitions", p_ms if dom != "k_partition" else c_ms) if not split_pass2 else None,
                     "achievable_peak": stream_gbs, "frac_of_achievable": (achieved / stream_gbs) if stream_gbs else None,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": k_ms,
                     "count_step_ms": c_ms + p_ms,
                     "count_step_frac": (alg_bytes / ((c_ms + p_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS) if c_ms + p_ms > 0 else 0.0,
                     "note": "integer/hash path bound by instruction issue and LDS round trips, not by HBM: see DESIGN.md section 4"},
        "stage_ms": {k: sum(t.get(k, 0.0) for t in all_t) / len(all_t) for k in sorted(all_t[-1]) if k not in ("peak_device_bytes", "device_bytes_now")},
    }
    if fallbacks:
        line["fallbacks"] = fallbacks                     # the paths that failed before the one measured (see config.parallelism)
    if iso_leg is not None:
        line["value_isolates_mode"] = n_bases * world / iso_leg / 1e9
        line["ms_per_step_isolates_mode"] = iso_leg * 1e3
        line["isolates_mode_note"] = ("comparison point: every rank assembles an isolate of its own (no data-path collective); "
                                      "`value` is the sharded path (DESIGN.md section 6)")
    if legs is not None:
        line["legs"] = legs
    if inflight_leg is not None:
        line["value_two_in_flight"] = n_bases / inflight_leg / 1e9
        line["ms_per_step_two_in_flight"] = inflight_leg * 1e3
        line["two_in_flight_note"] = ("the same steps with two handles in flight (two host threads, two streams): the kernels of one "
                                      "handle fill the host gaps of the other — a batch of isolates on one GPU (sparrowhawk_amd/batch.py)")
    if inflight_leg is not None:
        line["two_in_flight"] = {"value": n_bases / inflight_leg / 1e9, "unit": "Gbases/s", "clock": "device-resident", "ms_per_step": inflight_leg * 1e3}
    if sharded_one_rank is not None:
        line["sharded_one_rank"] = sharded_one_rank
    if host_leg is not None:
        line["host_pinned"] = {"value": n_bases / host_leg / 1e9, "unit": "Gbases/s", "clock": "host-pinned", "ms_per_step": host_leg * 1e3,
                               "note": "SURVEY.md 8(d): packed reads in host pinned memory -> contig strings on the host, one handle at a time"}
        if host_inflight_leg is not None:
            line["host_pinned_two_in_flight"] = {"value": n_bases / host_inflight_leg / 1e9, "unit": "Gbases/s", "clock": "host-pinned",
                                                 "ms_per_step": host_inflight_leg * 1e3,
                                                 "note": "the same clock with two handles in flight: the upload of isolate i+1 under the kernels of isolate i"}
        line["value_host_pinned"] = n_bases / host_leg / 1e9
        line["ms_per_step_host_pinned"] = host_leg * 1e3
        line["host_pinned_note"] = ("packed reads in host pinned memory -> contigs on host (SURVEY.md 8d clock): %.0f MB uploaded in "
                                    "pieces on a copy stream, pass 1 of a piece under the upload of the next (shk_preprocess_packed_host)"
                                    % ((d_bases.numel() + d_seg.numel()) * 4 / 1e6))
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.k <= 63:
            line["cpu_baseline"] = cpu_baseline(args.k, args.min_count, d_bases.cpu().numpy().view("uint32"),
                                                d_seg.cpu().numpy().view("uint32"), n_bases, res["outfasta"])
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())          # the ONE line on the real stdout
    if sharded and hasattr(comm, "free"):
        comm.free()                                      # the library's RCCL communicator goes before torch's group
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
