"""Batch of isolates on N GPUs (BASELINE.json configs[3]: 96 bacterial isolates on 8 MI355X).

The reference assembles one isolate per `AssemblyHelper` (www/src/workers/Assembler.ts:92-100: one handle,
one preprocess, one assemble); a batch is a loop over handles.  Two ways to spread that loop over ranks
(cfg 4):

  rounds    one isolate per round, ALL ranks work on it: every rank holds 1/world of the isolate's reads,
            the k-mer space is sharded by minimiser partition and the records cross in one pairwise RCCL
            exchange (shk_shard_preprocess); with the library's communicator the graph stays sharded too and
            shk_assemble runs collectively (csrc/shard_graph.h); every rank ends with the identical output.
            Isolates never share a table, so no isolate id has to ride in the keys.
  replicas  isolate i belongs to rank i % world, which assembles it alone: independent objects, no
            data-path collective — the comparison point.

Both give byte-identical per-isolate results (tests/test_gpu_configs.py); they differ in what scales.
"""
from .helper import AssemblyHelper


def isolates_of_rank(n_isolates, rank, world):
    """replicas: the isolates rank `rank` assembles."""
    return list(range(rank, n_isolates, world))


def assemble_batch(n_isolates, reads_for, params, mode="rounds", rank=0, world=1, comm=None, torch_comm=None,
                   keep=True, on_result=None, inflight=2):
    """Assembles isolates 0..n_isolates-1.

    reads_for(i, share_rank, share_world) -> object with .words/.seg_off device tensors and n_seg / n_bases /
        n_reads: the packed reads of isolate i (share_world == 1: all of them; otherwise this rank's share).
    params: dict(k, min_count, min_qual, do_fit, no_bubble_collapse, no_dead_end_removal).
    comm: sparrowhawk_amd.dist.LibComm (RCCL inside the library) — or torch_comm: dist.Comm (the torch/gloo
        rehearsal of the same pieces) — for mode 'rounds' with world > 1.
    inflight: replicas — handles kept in flight on this rank's GPU.  A handle's stream idles while its host side reads
        counters back, sizes the next phase and writes the FASTA / GFA text (about a fifth of an isolate's wall time);
        with two handles, each driven by its own host thread on its own stream, the kernels of one fill the gaps of the
        other.  The reads are still produced one isolate after the other, by the calling thread.
    Returns {isolate: (preprocessing_json, assembly_json, timings)} for the isolates this rank finished
    (rounds: every isolate on every rank; replicas: this rank's own).  on_result(i, helper) is called before
    the handle is freed (for stage inspection; with inflight > 1 from a worker thread)."""
    if mode not in ("rounds", "replicas"):
        raise ValueError("mode must be 'rounds' or 'replicas'")
    out = {}

    def new_helper():
        return AssemblyHelper.new(params["k"], False, params.get("min_count", 5), params.get("min_qual", 20), 0, False,
                                  params.get("do_fit", False), params.get("no_bubble_collapse", False),
                                  params.get("no_dead_end_removal", False))

    if mode == "replicas" or world == 1:
        mine = isolates_of_rank(n_isolates, rank, world) if mode == "replicas" else list(range(n_isolates))

        def one(i, d):
            h = new_helper()
            h.preprocess_packed_device(d.words.data_ptr(), d.seg_off.data_ptr(), d.n_seg, d.n_bases, d.n_reads)
            h.assemble()
            if on_result:
                on_result(i, h)
            out[i] = (h.get_preprocessing_info(), h.get_assembly() if keep else None, h.timings())
            h.free()

        if inflight <= 1 or len(mine) <= 1:
            for i in mine:
                d = reads_for(i, 0, 1)
                one(i, d)
                del d
            return out
        # several handles in flight: the library's calls block, so every handle gets a host thread of its own
        # (ctypes releases the interpreter lock inside them) and — inside the library — a stream of its own
        import queue
        import threading
        todo = queue.Queue(maxsize=inflight)          # bounded: at most `inflight` isolates' reads wait on the device
        errors = []

        def worker():
            while True:
                item = todo.get()
                if item is None:
                    return
                try:
                    if not errors:
                        one(*item)
                except BaseException as e:             # noqa: BLE001 (handed to the caller below)
                    errors.append(e)
        threads = [threading.Thread(target=worker, daemon=True) for _ in range(inflight)]
        for t in threads:
            t.start()
        try:
            for i in mine:
                if errors:
                    break
                todo.put((i, reads_for(i, 0, 1)))
        finally:
            for _ in threads:
                todo.put(None)
            for t in threads:
                t.join()
        if errors:
            raise errors[0]
        return out

    from .dist import sharded_preprocess, sharded_preprocess_rccl
    for i in range(n_isolates):
        d = reads_for(i, rank, world)
        h = new_helper()
        if comm is not None:
            sharded_preprocess_rccl(h, d.words.data_ptr(), d.seg_off.data_ptr(), d.n_seg, d.n_bases, d.n_reads, comm)
        elif torch_comm is not None:
            sharded_preprocess(h, d.words, d.seg_off, d.n_seg, d.n_bases, d.n_reads, torch_comm)
        else:
            raise ValueError("mode 'rounds' on several ranks needs a communicator")
        h.assemble()
        if on_result:
            on_result(i, h)
        out[i] = (h.get_preprocessing_info(), h.get_assembly() if keep else None, h.timings())
        h.free()
        del d
    return out
