"""Shard layer launcher: one process per GPU.

The production path is `LibComm` + `sharded_preprocess_rccl`: the collectives run INSIDE libshk_hip.so
(shk_shard_preprocess, csrc/shard_comm.hip: RCCL over xGMI); Python only carries the 128-byte ncclUniqueId
from rank 0 to the other ranks (over whatever torch.distributed group the launcher has) and makes the one
call.  `Comm` + `sharded_preprocess` below drive the same five shk_shard_* pieces with torch.distributed
collectives instead; they exist so the sequence can be rehearsed where RCCL cannot run (gloo on CPU, or
several ranks on the one GPU of a test box) and so the exchange plan can be tested without a GPU.

Replaces the read-parallel (rayon) driver the north_star attributes to the crate's native build
(not in the reference tree: SURVEY.md §8a row a15).  The k-mer space is partitioned by
minimiser-hash partition — partition p belongs to rank p % world — so every instance of a
canonical k-mer, from any rank's reads, is counted on exactly one GPU:

    every rank: its share of the reads --k_partition--> super-k-mer records per partition
    ONE all-to-all of the packed records (RCCL over xGMI; pairwise messages, all links busy)
    every rank: k_count_partitions over the partitions it owns -> local rows + local histogram
    all-reduce of the 500-bin histogram (4 KB), then the fit / threshold (identical everywhere)
    all-gather of the solid rows (12-20 B per solid k-mer), then the graph phases on every rank

Only the record exchange is big; the other two collectives move kilobytes and the solid set.
PyTorch is plumbing here: device buffers for the exchange and the collectives themselves.  With a
non-NCCL backend (gloo, used by the tests) tensors are staged through host memory.
"""
import ctypes as C

import numpy as np

from . import _lib
from .helper import AssemblyHelper, ShkError


def choose_partitions(total_instances_ub, world, per_part=100_000):
    """Power of two in [64, 16384], >= world, ~per_part k-mer instances per partition."""
    P = 64
    while P < 16384 and P * per_part < total_instances_ub:
        P <<= 1
    while P < world:
        P <<= 1
    return P


def plan_exchange(part_records_all, rank):
    """Pure host logic of the record exchange.

    part_records_all: uint64 [world, P] — records rank s holds for partition p.
    Returns a dict with, for `rank`:
      owned        partitions this rank counts (p % world == rank), ascending
      base         uint64 [P]: record offset of partition p in this rank's send buffer
                   (destination-major: all partitions of dest 0, then dest 1, ...)
      send_counts  uint64 [world]: records sent to each destination
      recv_counts  uint64 [world]: records received from each source
      run_off/run_cnt  [n_owned, world]: where source s's run of owned partition j sits in the
                   receive buffer (sources concatenated in rank order)
    """
    pr = np.asarray(part_records_all, dtype=np.uint64)
    world, P = pr.shape
    mine = pr[rank]
    base = np.zeros(P, dtype=np.uint64)
    send_counts = np.zeros(world, dtype=np.uint64)
    off = 0
    for d in range(world):
        for p in range(d, P, world):
            base[p] = off
            off += int(mine[p])
            send_counts[d] += mine[p]
    owned = np.arange(rank, P, world)
    recv_counts = pr[:, owned].sum(axis=1).astype(np.uint64)
    recv_base = np.concatenate([[0], np.cumsum(recv_counts)[:-1]]).astype(np.uint64)
    run_cnt = pr[:, owned].T.astype(np.uint32)                      # [n_owned, world]
    within = np.cumsum(pr[:, owned], axis=1) - pr[:, owned]         # [world, n_owned] prefix inside a source block
    run_off = (recv_base[:, None] + within).T.astype(np.uint64)     # [n_owned, world]
    return dict(owned=owned, base=base, send_counts=send_counts, recv_counts=recv_counts,
                run_off=np.ascontiguousarray(run_off), run_cnt=np.ascontiguousarray(run_cnt))


class Comm:
    """The three collectives the shard layer needs, on device tensors (nccl) or staged (gloo)."""

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        self.staged = dist.get_backend(group) != "nccl"

    def _cdev(self):
        return self.torch.device("cpu") if self.staged else self.device

    def all_gather_u64(self, arr):
        """host uint64[n] from every rank -> uint64 [world, n]"""
        t = self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint64).view(np.int64)).to(self._cdev())
        out = self.torch.empty(self.world * t.numel(), dtype=self.torch.int64, device=self._cdev())
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out.cpu().numpy().view(np.uint64).reshape(self.world, t.numel())

    def all_reduce_u64(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint64).view(np.int64)).to(self._cdev())
        self.dist.all_reduce(t, group=self.group)
        return t.cpu().numpy().view(np.uint64)

    def all_to_all_bytes(self, send, send_bytes, recv_bytes):
        """send: uint8 device tensor laid out destination-major; returns the uint8 receive tensor."""
        torch = self.torch
        total = int(sum(recv_bytes))
        if self.staged:
            src = send.cpu()
            dst = torch.empty(total, dtype=torch.uint8)
            self.dist.all_to_all_single(dst, src, [int(x) for x in recv_bytes], [int(x) for x in send_bytes], group=self.group)
            return dst.to(self.device)
        dst = torch.empty(total, dtype=torch.uint8, device=self.device)
        self.dist.all_to_all_single(dst, send, [int(x) for x in recv_bytes], [int(x) for x in send_bytes], group=self.group)
        return dst

    def all_gather_var(self, t, counts):
        """t: device tensor with counts[rank] leading elements used; returns the concatenation over ranks."""
        torch = self.torch
        mx = int(max(counts)) if len(counts) else 0
        pad = torch.zeros(max(mx, 1), dtype=t.dtype, device=t.device)
        n = int(counts[self.rank])
        if n:
            pad[:n] = t[:n]
        if self.staged:
            pad = pad.cpu()
        out = torch.empty(self.world * pad.numel(), dtype=pad.dtype, device=pad.device)
        self.dist.all_gather_into_tensor(out, pad, group=self.group)
        out = out.view(self.world, pad.numel())
        pieces = [out[s, :int(counts[s])] for s in range(self.world)]
        return torch.cat(pieces).to(t.device) if pieces else pad[:0].to(t.device)


class LibComm:
    """An RCCL communicator owned by libshk_hip.so (include/shk.h: shk_comm_*), one per process and GPU.
    The unique id travels over an existing torch.distributed group (any backend) or, for world 1, nowhere."""

    def __init__(self, rank=0, world=1, group=None):
        self._L = _lib.load()
        self._c = None
        ident = (C.c_uint8 * 128)()
        status, why = 0, ""
        if rank == 0:
            if self._L.shk_comm_unique_id(ident) != 0:
                status, why = 1, self._L.shk_comm_error().decode()
        if world > 1:
            # Every rank takes part in the broadcast whatever happened on rank 0: the id travels with a status byte
            # (129 bytes), and a failure is raised on EVERY rank after it — the callers' next collective
            # ("did all communicators come up?") then still matches across ranks.
            import torch
            import torch.distributed as dist
            t = torch.tensor(list(bytes(ident)) + [status], dtype=torch.uint8)
            if dist.get_backend(group) == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0, group=group)
            got = t.cpu().tolist()
            ident = (C.c_uint8 * 128)(*got[:128])
            if got[128] and rank != 0:
                status, why = 1, "rank 0 could not create the RCCL unique id"
        if status:
            raise ShkError(-5, why)
        self._c = self._L.shk_comm_init(ident, int(rank), int(world))
        if not self._c:
            raise ShkError(-5, self._L.shk_comm_error().decode())
        self.rank, self.world = int(rank), int(world)

    def free(self):
        if self._c:
            self._L.shk_comm_free(self._c)
            self._c = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sharded_preprocess_rccl(helper: AssemblyHelper, d_bases_ptr, d_seg_off_ptr, n_seg, n_bases, n_reads,
                            comm: LibComm, n_partitions=0):
    """Collective over `comm`: every rank hands in its share of the packed reads (device pointers); on return
    every rank's helper is 'preprocessed' with the same global solid set.  One C call, no Python collectives."""
    helper._check(helper._L.shk_shard_preprocess(helper._h, comm._c, d_bases_ptr if n_seg else None,
                                                 d_seg_off_ptr if n_seg else None, int(n_seg), int(n_bases),
                                                 int(n_reads), int(n_partitions or 0)))


def lib_plan_exchange(part_records_all, rank):
    """The exchange plan as the library computes it (shk_plan_exchange) — same dict as plan_exchange()."""
    L = _lib.load()
    pr = np.ascontiguousarray(part_records_all, dtype=np.uint64)
    world, P = pr.shape
    n_owned = len(range(rank, P, world))
    base = np.zeros(P, dtype=np.uint64)
    sc, rc = np.zeros(world, dtype=np.uint64), np.zeros(world, dtype=np.uint64)
    ro, rn = np.zeros((n_owned, world), dtype=np.uint64), np.zeros((n_owned, world), dtype=np.uint32)
    r = L.shk_plan_exchange(pr.ctypes.data, world, P, rank, base.ctypes.data, sc.ctypes.data, rc.ctypes.data,
                            ro.ctypes.data, rn.ctypes.data)
    if r != 0:
        raise ShkError(r, "shk_plan_exchange")
    return dict(owned=np.arange(rank, P, world), base=base, send_counts=sc, recv_counts=rc, run_off=ro, run_cnt=rn)


def _ptr_tensor(torch, ptr, nbytes, device):
    """uint8 tensor view of library-owned device memory (copied by the caller before the next call)."""
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device=device)

    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)


def sharded_preprocess(helper: AssemblyHelper, d_bases, d_seg_off, n_seg, n_bases, n_reads, comm: Comm,
                       n_partitions=None):
    """Runs preprocess across all ranks of `comm` for one pooled sample.  d_bases / d_seg_off are
    torch device tensors with this rank's packed reads.  On return every rank's helper is in the
    'preprocessed' state with the same global solid set; call helper.assemble() next."""
    torch = comm.torch
    L = helper._L
    k = helper.k
    W = (2 * k + 63) // 64
    world, rank = comm.world, comm.rank
    inst_ub = max(0, int(n_bases) - int(n_seg) * (k - 1))
    if n_partitions is None:
        total = int(comm.all_reduce_u64(np.array([inst_ub], dtype=np.uint64))[0])
        n_partitions = choose_partitions(total, world)
    P = int(n_partitions)

    part = np.zeros(P, dtype=np.uint64)
    helper._check(L.shk_shard_partition(helper._h, d_bases.data_ptr() if n_seg else None,
                                        d_seg_off.data_ptr() if n_seg else None, int(n_seg), int(n_bases),
                                        int(n_reads), P, part.ctypes.data))
    rec_bytes = int(L.shk_shard_record_bytes(helper._h))
    plan = plan_exchange(comm.all_gather_u64(part), rank)

    send = torch.empty(max(1, int(part.sum()) * rec_bytes), dtype=torch.uint8, device=comm.device)
    helper._check(L.shk_shard_pack(helper._h, send.data_ptr(), plan["base"].ctypes.data, P))
    torch.cuda.synchronize(comm.device)
    recv = comm.all_to_all_bytes(send[:int(part.sum()) * rec_bytes], plan["send_counts"] * rec_bytes,
                                 plan["recv_counts"] * rec_bytes)
    torch.cuda.synchronize(comm.device)
    del send

    histo = np.zeros(500, dtype=np.uint64)
    inst = C.c_uint64(0)
    if recv.numel() == 0:
        recv = torch.zeros(64, dtype=torch.uint8, device=comm.device)
    helper._check(L.shk_shard_count(helper._h, recv.data_ptr(), plan["run_off"].ctypes.data,
                                    plan["run_cnt"].ctypes.data, len(plan["owned"]), world, histo.ctypes.data,
                                    C.byref(inst)))
    red = comm.all_reduce_u64(np.concatenate([histo, np.array([inst.value], dtype=np.uint64)]))
    g_histo, g_inst = np.ascontiguousarray(red[:500]), int(red[500])

    keys = (C.c_void_p * W)()
    cnt = C.c_void_p()
    n_rows = C.c_uint64(0)
    used = C.c_uint32(0)
    helper._check(L.shk_shard_rows(helper._h, g_histo.ctypes.data, keys, C.byref(cnt), C.byref(n_rows), C.byref(used)))
    n_local = int(n_rows.value)
    counts = comm.all_gather_u64(np.array([n_local], dtype=np.uint64))[:, 0]
    g_keys = []
    for j in range(W):
        t = _ptr_tensor(torch, keys[j], n_local * 8, comm.device).view(torch.int64)
        g_keys.append(comm.all_gather_var(t, counts))
    g_cnt = comm.all_gather_var(_ptr_tensor(torch, cnt.value, n_local * 4, comm.device).view(torch.int32), counts)
    torch.cuda.synchronize(comm.device)
    n_total = int(counts.sum())
    kp = (C.c_void_p * W)(*[t.data_ptr() if n_total else None for t in g_keys])
    helper._check(L.shk_shard_set_solid(helper._h, kp, g_cnt.data_ptr() if n_total else None, n_total, g_inst))
    del recv
    return dict(n_partitions=P, records_sent=int(part.sum()), record_bytes=rec_bytes, n_solid=n_total,
                used_min_count=int(used.value), total_instances=g_inst)
