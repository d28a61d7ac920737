"""The counting stage of the shard layer with N "ranks" in ONE process, no transport: every rank is a handle, the exchange is a
set of device-tensor copies laid out by plan_exchange (the shk_shard_partition / _pack / _count / _rows pieces the library's
own multi-rank call is made of).  The union of the ranks' solid rows must be the oracle's solid set, every k-mer once.
Usage: python tools/emulate_ranks.py <seed> <n_cases> <world> [first index]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_dist
from util import run_oracle, sorted_table
from sparrowhawk_amd import AssemblyHelper, pack_fastq
from sparrowhawk_amd.dist import plan_exchange, choose_partitions, _ptr_tensor
seed, n, world = (int(x) for x in sys.argv[1:4])
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda", 0)
cases = test_dist._graph_cases(seed + world, n, first_case=world)
bad = 0
for idx in range(first, n):
    fq, pr = cases[idx]
    k = pr["k"]; W = (2 * k + 63) // 64
    recs = fq.decode().split("@r")[1:]
    hs, dbs, dss, meta = [], [], [], []
    for r in range(world):
        mine = ("@r" + "@r".join(recs[r::world])).encode() if recs[r::world] else b""
        bases, seg, nb, nr = pack_fastq(mine, k, pr["min_qual"])
        dbs.append(torch.from_numpy(bases.view(np.int32)).to(dev)); dss.append(torch.from_numpy(seg.view(np.int32)).to(dev))
        meta.append((len(seg) - 1, nb, nr))
        hs.append(AssemblyHelper.new(k, False, pr["min_count"], pr["min_qual"], 0, False, False, False, False))
    torch.cuda.synchronize()
    L = hs[0]._L
    inst = sum(max(0, nb - ns * (k - 1)) for ns, nb, nr in meta)
    P = choose_partitions(inst, world, 100000 if W == 1 else 40000)
    parts = np.zeros((world, P), dtype=np.uint64)
    for r in range(world):
        ns, nb, nr = meta[r]
        hs[r]._check(L.shk_shard_partition(hs[r]._h, dbs[r].data_ptr() if ns else None, dss[r].data_ptr() if ns else None, ns, nb, nr, P, parts[r].ctypes.data))
    rec_bytes = int(L.shk_shard_record_bytes(hs[0]._h))
    plans = [plan_exchange(parts, r) for r in range(world)]
    sends = []
    for r in range(world):
        s = torch.zeros(max(1, int(parts[r].sum()) * rec_bytes), dtype=torch.uint8, device=dev)
        hs[r]._check(L.shk_shard_pack(hs[r]._h, s.data_ptr(), plans[r]["base"].ctypes.data, P))
        sends.append(s)
    torch.cuda.synchronize()
    histos, insts, recvs = [], [], []
    for r in range(world):
        pieces = []
        for s in range(world):                                   # source s's block for destination r
            off = int(plans[s]["send_counts"][:r].sum()) * rec_bytes
            ln = int(plans[s]["send_counts"][r]) * rec_bytes
            pieces.append(sends[s][off:off + ln])
        recv = torch.cat(pieces) if sum(p.numel() for p in pieces) else torch.zeros(64, dtype=torch.uint8, device=dev)
        recvs.append(recv)
        histo = np.zeros(500, dtype=np.uint64); ins = C.c_uint64(0)
        torch.cuda.synchronize()
        hs[r]._check(L.shk_shard_count(hs[r]._h, recv.data_ptr(), plans[r]["run_off"].ctypes.data, plans[r]["run_cnt"].ctypes.data,
                                       len(plans[r]["owned"]), world, histo.ctypes.data, C.byref(ins)))
        histos.append(histo); insts.append(ins.value)
    g_histo = np.ascontiguousarray(np.sum(histos, axis=0).astype(np.uint64))
    all_keys, all_cnt = [], []
    for r in range(world):
        keys = (C.c_void_p * W)(); cnt = C.c_void_p(); n_rows = C.c_uint64(0); used = C.c_uint32(0)
        hs[r]._check(L.shk_shard_rows(hs[r]._h, g_histo.ctypes.data, keys, C.byref(cnt), C.byref(n_rows), C.byref(used)))
        nl = int(n_rows.value)
        kk = np.stack([_ptr_tensor(torch, keys[j], nl * 8, dev).view(torch.int64).cpu().numpy().view(np.uint64) for j in range(W)], axis=1) if nl else np.zeros((0, W), dtype=np.uint64)
        cc = _ptr_tensor(torch, cnt.value, nl * 4, dev).view(torch.int32).cpu().numpy().view(np.uint32) if nl else np.zeros(0, dtype=np.uint32)
        all_keys.append(kk); all_cnt.append(cc)
    hk, hc, _ = sorted_table(np.concatenate(all_keys), np.concatenate(all_cnt))
    o = run_oracle([fq], k=k, min_count=pr["min_count"], min_qual=pr["min_qual"])
    ok_, oc_ = o.solid()
    good = np.array_equal(hk, ok_) and np.array_equal(hc, oc_) and sum(insts) == o.total_instances
    if not good:
        bad += 1
        print("case", idx, pr, "rows", len(hc), "oracle", len(oc_), "instances", sum(insts), o.total_instances, "DIFFER", flush=True)
    for h in hs: h.free()
print("%d of %d cases differ (world %d)" % (bad, n - first, world))
