"""Worker for the multi-process tests (launched with torch.distributed.run).

mode "plan": CPU only — exercises plan_exchange + the record exchange with tagged fake records.
mode "gpu" : every rank drives the real HIP path on cuda:0 (ranks share the one GPU of the test
             box; collectives go through gloo with host staging) and rank 0 writes the result.
mode "rccl": the production path — the collectives run inside libshk_hip.so over RCCL
             (shk_shard_preprocess); torch.distributed (gloo) only carries the ncclUniqueId.  Rank r
             uses cuda:(r % device_count).  RCCL refuses two ranks on one GPU: the worker then
             reports {"skipped": reason} instead of a result.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def set_piece(cfg):
    """cfg["piece"]: SHK_COMM_PIECE_BYTES — every exchange and large collective of the library cut into pieces of that size"""
    if cfg.get("piece"):
        os.environ["SHK_COMM_PIECE_BYTES"] = str(cfg["piece"])


def set_dedupe(cfg, rank):
    """cfg["dedupe"]: "0" / "1" (SHK_SHARD_DEDUPE: records deduplicated by the sender never / always), absent = the library decides;
    a list gives every rank its own setting (the ranks must still agree on what travels)."""
    v = cfg.get("dedupe")
    if isinstance(v, list):
        v = v[rank % len(v)]
    if v is None or v == "auto":
        os.environ.pop("SHK_SHARD_DEDUPE", None)
    else:
        os.environ["SHK_SHARD_DEDUPE"] = str(v)


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from sparrowhawk_amd.dist import Comm, plan_exchange, choose_partitions

    if mode == "plan":
        comm = Comm(device=torch.device("cpu"))
        P = 64
        rng = np.random.default_rng(100 + rank)
        part = rng.integers(0, 40, P).astype(np.uint64)
        part[rng.integers(0, P, 5)] = 0
        allp = comm.all_gather_u64(part)
        plan = plan_exchange(allp, rank)
        # fake 16-byte records tagged (src, partition, index)
        recs = np.zeros((int(part.sum()), 2), dtype=np.uint64)
        for p in range(P):
            b = int(plan["base"][p])
            for i in range(int(part[p])):
                recs[b + i] = (rank << 32 | p, i)
        send = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy())
        recv = comm.all_to_all_bytes(send, plan["send_counts"] * 16, plan["recv_counts"] * 16)
        got = recv.numpy().view(np.uint64).reshape(-1, 2)
        ok = True
        for j, p in enumerate(plan["owned"]):
            for s in range(world):
                o, c = int(plan["run_off"][j, s]), int(plan["run_cnt"][j, s])
                assert c == int(allp[s, p])
                blk = got[o:o + c]
                ok &= bool((blk[:, 0] == (s << 32 | int(p))).all() and (blk[:, 1] == np.arange(c)).all())
        ok &= int(plan["recv_counts"].sum()) == got.shape[0]
        tot = comm.all_reduce_u64(np.array([got.shape[0]], dtype=np.uint64))[0]
        ok &= int(tot) == int(allp.sum())
        res = {"ok": bool(ok), "rank": rank, "P": choose_partitions(10 ** 9, world)}
        with open(f"{out_path}.{rank}", "w") as f:
            json.dump(res, f)
    elif mode == "batch":
        # configs[3] rounds path: every isolate sharded over the ranks (both on cuda:0; collectives over gloo)
        import hashlib
        from sparrowhawk_amd import synth
        from sparrowhawk_amd.batch import assemble_batch
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        comm = Comm(device=dev)
        cfg = json.load(open(sys.argv[3]))
        lengths, k, seed0, L = cfg["lengths"], cfg["k"], cfg["seed0"], 150
        cache = {}

        def reads_for(i, share_rank, share_world):
            n_reads = int(lengths[i]) * cfg["coverage"] // L
            if i not in cache:
                cache.clear()
                cache[i] = synth.device_genomes(torch, dev, [int(lengths[i])], seed0 + i)
            g, off = cache[i]
            lo, hi = n_reads * share_rank // share_world, n_reads * (share_rank + 1) // share_world
            return synth.device_sample_reads(torch, dev, g, off, np.array([1.0]), hi - lo, L, k, seed0 + i,
                                             err=cfg["err"], read_index0=lo)
        out = assemble_batch(len(lengths), reads_for, dict(k=k, min_count=cfg["min_count"], min_qual=20), mode="rounds",
                             rank=rank, world=world, torch_comm=comm)
        res = {}
        for i, (pre, asm, t) in out.items():
            res[str(i)] = {"pre": pre, "asm_sha256": hashlib.sha256(asm.encode()).hexdigest()}
            if cfg.get("keep"):
                res[str(i)]["asm"] = asm
        with open(f"{out_path}.{rank}", "w") as f:
            json.dump(res, f)
    elif mode == "rccl_many":
        # several inputs through shk_shard_preprocess + the collective shk_assemble in ONE launch (process start-up is
        # most of a small case's time): cfg["cases"] = [{fastq, k, min_count, min_qual, do_fit, no_bubble_collapse,
        # no_dead_end_removal, P}]; every rank writes the list of its results
        from sparrowhawk_amd import AssemblyHelper, pack_fastq, ShkError
        from sparrowhawk_amd.dist import LibComm, sharded_preprocess_rccl
        dev = torch.device("cuda", rank % max(1, torch.cuda.device_count()))
        torch.cuda.set_device(dev)
        cfg = json.load(open(sys.argv[3]))
        comm = LibComm(rank, world)
        results = []
        if os.environ.get("SHK_DIST_FUZZ_LOG"):          # a file per rank: the case it is in and whatever the library prints (SHK_STAGE_LOG)
            lf = os.open(os.environ["SHK_DIST_FUZZ_LOG"] + f".{rank}", os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
            os.dup2(lf, 2)
        for ci, cs in enumerate(cfg["cases"]):
            if os.environ.get("SHK_DIST_FUZZ_LOG"):
                os.write(2, f"case {ci}: {dict((a, b) for a, b in cs.items() if a != 'fastq')}\n".encode())
            fq = open(cs["fastq"], "rb").read()
            k = cs["k"]
            recs = fq.decode().split("@r")[1:]
            mine = ("@r" + "@r".join(recs[rank::world])).encode() if recs[rank::world] else b""
            bases, seg, nb, nr = pack_fastq(mine, k, cs["min_qual"])
            d_bases = torch.from_numpy(bases.view(np.int32)).to(dev)
            d_seg = torch.from_numpy(seg.view(np.int32)).to(dev)
            torch.cuda.synchronize()
            h = AssemblyHelper.new(k, bool(cs.get("verbose", False)), cs["min_count"], cs["min_qual"], 0, False, bool(cs.get("do_fit", False)),
                                   bool(cs.get("no_bubble_collapse", False)), bool(cs.get("no_dead_end_removal", False)))
            set_dedupe(cs, rank)
            try:
                sharded_preprocess_rccl(h, d_bases.data_ptr(), d_seg.data_ptr(), len(seg) - 1, nb, nr, comm, n_partitions=cs.get("P") or 0)
                h.assemble()
                res = {"pre": h.get_preprocessing_info(), "asm": h.get_assembly(), "n_solid_local": h.n_solid}
                if cs.get("timings"):
                    res["timings"] = h.timings()
            except ShkError as e:
                res = {"error": str(e)}
            results.append(res)
            h.free()
        with open(f"{out_path}.{rank}", "w") as f:
            json.dump(results, f)
        comm.free()
    elif mode == "gloo_many":
        # the same many-case list, but the five shk_shard_* pieces driven by torch.distributed collectives (gloo, staged through
        # the host) instead of the library's communicator: no stand-in transport, no collective code of the library
        from sparrowhawk_amd import AssemblyHelper, pack_fastq, ShkError
        from sparrowhawk_amd.dist import sharded_preprocess
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        comm = Comm(device=dev)
        cfg = json.load(open(sys.argv[3]))
        results = []
        for cs in cfg["cases"]:
            fq = open(cs["fastq"], "rb").read()
            k = cs["k"]
            recs = fq.decode().split("@r")[1:]
            mine = ("@r" + "@r".join(recs[rank::world])).encode() if recs[rank::world] else b""
            bases, seg, nb, nr = pack_fastq(mine, k, cs["min_qual"])
            d_bases = torch.from_numpy(bases.view(np.int32)).to(dev)
            d_seg = torch.from_numpy(seg.view(np.int32)).to(dev)
            torch.cuda.synchronize()
            h = AssemblyHelper.new(k, False, cs["min_count"], cs["min_qual"], 0, False, bool(cs.get("do_fit", False)),
                                   bool(cs.get("no_bubble_collapse", False)), bool(cs.get("no_dead_end_removal", False)))
            try:
                sharded_preprocess(h, d_bases, d_seg, len(seg) - 1, nb, nr, comm, n_partitions=cs.get("P") or None)
                h.assemble()
                res = {"pre": h.get_preprocessing_info(), "asm": h.get_assembly()}
            except ShkError as e:
                res = {"error": str(e)}
            results.append(res)
            h.free()
        with open(f"{out_path}.{rank}", "w") as f:
            json.dump(results, f)
    elif mode == "rccl":
        from sparrowhawk_amd import AssemblyHelper, pack_fastq, ShkError
        from sparrowhawk_amd.dist import LibComm, sharded_preprocess_rccl
        dev = torch.device("cuda", rank % max(1, torch.cuda.device_count()))
        torch.cuda.set_device(dev)
        cfg = json.load(open(sys.argv[3]))
        set_piece(cfg)
        fq = open(cfg["fastq"], "rb").read()
        k = cfg["k"]
        try:
            comm = LibComm(rank, world)
        except ShkError as e:
            with open(f"{out_path}.{rank}", "w") as f:
                json.dump({"skipped": str(e)}, f)
            dist.barrier()
            dist.destroy_process_group()
            return
        recs = fq.decode().split("@r")[1:]
        mine = ("@r" + "@r".join(recs[rank::world])).encode() if recs[rank::world] else b""
        bases, seg, nb, nr = pack_fastq(mine, k, cfg["min_qual"])
        d_bases = torch.from_numpy(bases.view(np.int32)).to(dev)
        d_seg = torch.from_numpy(seg.view(np.int32)).to(dev)
        torch.cuda.synchronize()
        h = AssemblyHelper.new(k, True, cfg["min_count"], cfg["min_qual"], 0, False, cfg["do_fit"], False, False)
        set_dedupe(cfg, rank)
        if cfg.get("replicated"):                      # round 2's path: gather the solid set, every rank assembles the whole graph
            os.environ["SHK_SHARD_GRAPH"] = "0"
        if cfg.get("truncate"):                        # the stand-in transport delivers only the first bytes of every received block
            os.environ["MOCK_RCCL_TRUNCATE_BYTES"] = str(cfg["truncate"])
        inj = cfg.get("inject")                        # {"rank": r, "step": s}: that rank's local step fails (SHK_FAULT_INJECT)
        if inj and inj["rank"] == rank:
            os.environ["SHK_FAULT_INJECT"] = inj["step"]
        try:
            sharded_preprocess_rccl(h, d_bases.data_ptr(), d_seg.data_ptr(), len(seg) - 1, nb, nr, comm,
                                    n_partitions=cfg.get("P") or 0)
        except ShkError as e:
            # every rank must get here together (the error agreement inside shk_shard_preprocess): none may hang
            with open(f"{out_path}.{rank}", "w") as f:
                json.dump({"error": str(e), "code": e.code}, f)
            h.free()
            comm.free()
            dist.barrier()
            dist.destroy_process_group()
            return
        h.assemble()
        res = {"pre": h.get_preprocessing_info(), "asm": h.get_assembly(), "states": h.states,
               "total_instances": h.total_instances, "timings": h.timings()}
        with open(f"{out_path}.{rank}", "w") as f:
            json.dump(res, f)
        h.free()
        comm.free()
    else:
        from sparrowhawk_amd import AssemblyHelper, pack_fastq
        from sparrowhawk_amd.dist import sharded_preprocess
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        comm = Comm(device=dev)
        cfg = json.load(open(sys.argv[3]))
        fq = open(cfg["fastq"], "rb").read()
        k = cfg["k"]
        # this rank's share of the reads: records rank, rank+world, ...
        recs = fq.decode().split("@r")[1:]
        mine = ("@r" + "@r".join(recs[rank::world])).encode() if recs[rank::world] else b""
        bases, seg, nb, nr = pack_fastq(mine, k, cfg["min_qual"])
        d_bases = torch.from_numpy(bases.view(np.int32)).to(dev)
        d_seg = torch.from_numpy(seg.view(np.int32)).to(dev)
        h = AssemblyHelper.new(k, True, cfg["min_count"], cfg["min_qual"], 0, False, cfg["do_fit"], False, False)
        info = sharded_preprocess(h, d_bases, d_seg, len(seg) - 1, nb, nr, comm, n_partitions=cfg.get("P"))
        h.assemble()
        res = {"pre": h.get_preprocessing_info(), "asm": h.get_assembly(), "info": info, "states": h.states,
               "total_instances": h.total_instances}
        with open(f"{out_path}.{rank}", "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
