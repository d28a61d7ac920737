"""Multi-process tests of the shard layer (sparrowhawk_amd/dist.py).

CPU (gloo, world_size 2 and 3): the exchange plan and the all-to-all of tagged records.
GPU (-m gpu): 2 ranks on the one GPU of the test box, gloo with host staging, full pipeline —
the pooled result must equal the single-process result and the oracle."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def launch(nproc, args, port, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER] + args
    env = dict(os.environ, OMP_NUM_THREADS="1")
    # own process group: on a timeout the launcher AND its ranks are ended (no rank may outlive the test)
    pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        out, errs = pr.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(pr.pid, signal.SIGKILL)
        out, errs = pr.communicate()
        raise AssertionError(f"ranks did not finish within {timeout} s\n" + out[-2000:] + errs[-2000:])
    assert pr.returncode == 0, out[-3000:] + errs[-3000:]


def test_plan_exchange_is_consistent_single_process():
    from sparrowhawk_amd.dist import plan_exchange, choose_partitions
    rng = np.random.default_rng(1)
    world, P = 4, 32
    allp = rng.integers(0, 100, (world, P)).astype(np.uint64)
    plans = [plan_exchange(allp, r) for r in range(world)]
    for r, pl in enumerate(plans):
        assert list(pl["owned"]) == list(range(r, P, world))
        assert int(pl["send_counts"].sum()) == int(allp[r].sum())
        for d in range(world):
            assert int(pl["send_counts"][d]) == int(allp[r, d::world].sum())
            assert int(plans[d]["recv_counts"][r]) == int(pl["send_counts"][d])   # what r sends d is what d expects
        # bases are a permutation-prefix: destination-major, partitions ascending
        order = [p for d in range(world) for p in range(d, P, world)]
        off = 0
        for p in order:
            assert int(pl["base"][p]) == off
            off += int(allp[r, p])
    assert choose_partitions(10, 8) == 64 and choose_partitions(4 * 10 ** 9, 8) == 16384
    assert choose_partitions(100_000 * 1000, 2) == 1024


def test_library_plan_equals_the_python_plan():
    """shk_plan_exchange (what shk_shard_preprocess runs between its collectives) against dist.plan_exchange."""
    from sparrowhawk_amd.dist import plan_exchange, lib_plan_exchange
    from sparrowhawk_amd import _lib
    rng = np.random.default_rng(7)
    for world, P in ((1, 64), (2, 64), (3, 64), (8, 128), (8, 16384), (5, 8)):
        allp = rng.integers(0, 1000, (world, P)).astype(np.uint64)
        allp[rng.integers(0, world, 4), rng.integers(0, P, 4)] = 0
        for r in range(world):
            a, b = plan_exchange(allp, r), lib_plan_exchange(allp, r)
            for key in ("owned", "base", "send_counts", "recv_counts", "run_off", "run_cnt"):
                assert np.array_equal(np.asarray(a[key], dtype=np.uint64), np.asarray(b[key], dtype=np.uint64)), (world, P, r, key)
    L = _lib.load()
    from sparrowhawk_amd.dist import choose_partitions
    for tot in (10, 10 ** 6, 10 ** 8, 4 * 10 ** 9, 3 * 10 ** 10):
        for world in (1, 2, 8):
            assert L.shk_choose_partitions(tot, world, 1) == choose_partitions(tot, world)
            assert L.shk_choose_partitions(tot, world, 2) == choose_partitions(tot, world, per_part=40_000)


@pytest.mark.parametrize("world", [2, 3])
def test_record_exchange_gloo_cpu(world):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "res")
        launch(world, ["plan", out], 29600 + world)
        for r in range(world):
            res = json.load(open(f"{out}.{r}"))
            assert res["ok"] and res["rank"] == r


@pytest.mark.gpu
@pytest.mark.parametrize("k,do_fit,P", [(31, False, None), (51, True, 64), (89, False, 128)])
def test_sharded_pipeline_two_ranks_one_gpu(k, do_fit, P):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_dataset, run_oracle
    from sparrowhawk_amd import AssemblyHelper
    g, fq = make_dataset(40000, 40, err=0.01, seed=300 + k)
    with tempfile.TemporaryDirectory() as d:
        fqp = os.path.join(d, "reads.fq")
        open(fqp, "wb").write(fq)
        cfg = {"fastq": fqp, "k": k, "min_count": 3, "min_qual": 20, "do_fit": do_fit}
        if P:
            cfg["P"] = P
        cfgp = os.path.join(d, "cfg.json")
        json.dump(cfg, open(cfgp, "w"))
        out = os.path.join(d, "res")
        launch(2, ["gpu", out, cfgp], 29650 + k)
        res = [json.load(open(f"{out}.{r}")) for r in range(2)]
    assert res[0]["asm"] == res[1]["asm"] and res[0]["pre"] == res[1]["pre"]
    o = run_oracle([fq], k=k, min_count=3, min_qual=20, do_fit=do_fit)
    o.assemble()
    assert res[0]["pre"] == o.preprocessing_json()
    assert res[0]["asm"] == o.assembly_json()
    assert res[0]["total_instances"] == o.total_instances
    h = AssemblyHelper.new(k, True, 3, 20, 0, False, do_fit, False, False)
    h.preprocess(fq)
    h.assemble()
    assert h.get_assembly() == res[0]["asm"]
    assert res[0]["states"][0] == "preprocess:start" and res[0]["states"][-1] == "assembly:end"


@pytest.mark.gpu
@pytest.mark.parametrize("k,do_fit,P", [(31, False, 0), (51, True, 64)])
def test_rccl_inside_the_library_world_1(k, do_fit, P):
    """shk_shard_preprocess with a one-rank RCCL communicator (ncclCommInitRank, grouped send/recv to itself,
    all-reduce, broadcast-gather): same bytes as the plain single-GPU path and as the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from util import make_dataset, run_oracle
    from sparrowhawk_amd import AssemblyHelper, pack_fastq
    from sparrowhawk_amd.dist import LibComm, sharded_preprocess_rccl
    g, fq = make_dataset(40000, 40, err=0.01, seed=500 + k)
    dev = torch.device("cuda", 0)
    comm = LibComm(0, 1)
    bases, seg, nb, nr = pack_fastq(fq, k, 20)
    d_bases = torch.from_numpy(bases.view(np.int32)).to(dev)
    d_seg = torch.from_numpy(seg.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    h = AssemblyHelper.new(k, True, 3, 20, 0, False, do_fit, False, False)
    sharded_preprocess_rccl(h, d_bases.data_ptr(), d_seg.data_ptr(), len(seg) - 1, nb, nr, comm, n_partitions=P)
    h.assemble()
    o = run_oracle([fq], k=k, min_count=3, min_qual=20, do_fit=do_fit)
    o.assemble()
    assert h.get_preprocessing_info() == o.preprocessing_json()
    assert h.get_assembly() == o.assembly_json()
    assert h.total_instances == o.total_instances
    assert "shard_exchange_host_clock" in h.timings()
    h2 = AssemblyHelper.new(k, True, 3, 20, 0, False, do_fit, False, False)
    h2.preprocess(fq)
    h2.assemble()
    assert h2.get_assembly() == h.get_assembly()
    comm.free()


@pytest.mark.gpu
def test_record_blocks_beyond_one_gibibyte_cross_intact():
    """Round 3: a grouped ncclSend / ncclRecv of a rank to ITSELF delivered only part of a 1.2 GB block on the GPU box (the
    rest of the receive buffer kept stale pool memory: half the k-mers lost, duplicate rows, a broken graph — found with
    the configs[4] share).  The block a rank keeps is a device copy now and every other block travels in pieces of
    <= 256 MiB.  Here: 5.8 M reads of an isolate (77 M records = 1.23 GB, sent raw) through a one-rank communicator
    must count exactly what the plain path counts."""
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from sparrowhawk_amd import AssemblyHelper
    from sparrowhawk_amd.dist import LibComm, sharded_preprocess_rccl
    dev = torch.device("cuda", 0)
    d_bases, d_seg, n_reads, n_bases, _g = bench.make_reads_on_device(torch, dev, 5_000_000, 174, 150, 0xEC07)
    assert n_reads == 5_800_000
    h = AssemblyHelper.new(31, False, 5, 20, 0, False, False, False, False)
    h.preprocess_packed_device(d_bases.data_ptr(), d_seg.data_ptr(), n_reads, n_bases, n_reads)
    h.assemble()
    want = (h.n_distinct, h.n_solid, h.total_instances, h.get_preprocessing_info(), h.get_assembly())
    h.free()
    comm = LibComm(0, 1)
    for dedupe in ("0", "1"):
        os.environ["SHK_SHARD_DEDUPE"] = dedupe
        try:
            h = AssemblyHelper.new(31, False, 5, 20, 0, False, False, False, False)
            sharded_preprocess_rccl(h, d_bases.data_ptr(), d_seg.data_ptr(), n_reads, n_bases, n_reads, comm)
            t = h.timings()
            assert (t["shard_exchange_sent_MB"] > 1100.0) == (dedupe == "0"), t
            assert (h.n_distinct, h.n_solid, h.total_instances, h.get_preprocessing_info()) == want[:4]
            h.assemble()
            assert h.get_assembly() == want[4]
            h.free()
        finally:
            os.environ.pop("SHK_SHARD_DEDUPE", None)
    comm.free()


@pytest.mark.gpu
def test_rccl_inside_the_library_two_ranks():
    """Two ranks through shk_shard_preprocess.  On a one-GPU box RCCL refuses the second rank on the same
    device (duplicate GPU): the test then skips, saying so; on a box with >= 2 GPUs it compares with the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_dataset, run_oracle
    k = 31
    g, fq = make_dataset(40000, 40, err=0.01, seed=777)
    with tempfile.TemporaryDirectory() as d:
        fqp = os.path.join(d, "reads.fq")
        open(fqp, "wb").write(fq)
        cfgp = os.path.join(d, "cfg.json")
        json.dump({"fastq": fqp, "k": k, "min_count": 3, "min_qual": 20, "do_fit": False}, open(cfgp, "w"))
        out = os.path.join(d, "res")
        launch(2, ["rccl", out, cfgp], 29711, timeout=240)
        res = [json.load(open(f"{out}.{r}")) for r in range(2)]
    if any("skipped" in r for r in res):
        pytest.skip("RCCL with two ranks needs two GPUs: " + "; ".join(r.get("skipped", "ok") for r in res))
    o = run_oracle([fq], k=k, min_count=3, min_qual=20)
    o.assemble()
    assert res[0]["asm"] == res[1]["asm"] == o.assembly_json()
    assert res[0]["pre"] == res[1]["pre"] == o.preprocessing_json()


@pytest.mark.gpu
def test_torch_nccl_collectives_branch_world_1():
    """The rehearsal layer's nccl branch (dist.Comm with a non-staged backend: device tensors straight into
    all_to_all_single / all_gather_into_tensor, library-owned rows wrapped by _ptr_tensor) with a one-rank nccl
    (= RCCL) process group — in a child process, so that the test runner itself never joins a process group."""
    code = r'''
import json, os, sys
sys.path.insert(0, os.environ["SHK_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SHK_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29741")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from sparrowhawk_amd import AssemblyHelper, pack_fastq
from sparrowhawk_amd.dist import Comm, sharded_preprocess
from util import make_dataset, run_oracle
g, fq = make_dataset(40000, 40, err=0.01, seed=4242)
comm = Comm(device=dev)
assert not comm.staged
out = {}
for k, fit in ((31, False), (51, True)):
    bases, seg, nb, nr = pack_fastq(fq, k, 20)
    d_bases = torch.from_numpy(bases.view(np.int32)).to(dev); d_seg = torch.from_numpy(seg.view(np.int32)).to(dev)
    h = AssemblyHelper.new(k, True, 3, 20, 0, False, fit, False, False)
    sharded_preprocess(h, d_bases, d_seg, len(seg) - 1, nb, nr, comm)
    h.assemble()
    o = run_oracle([fq], k=k, min_count=3, min_qual=20, do_fit=fit); o.assemble()
    out[str(k)] = bool(h.get_assembly() == o.assembly_json() and h.get_preprocessing_info() == o.preprocessing_json())
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''
    env = dict(os.environ, SHK_ROOT=ROOT, OMP_NUM_THREADS="1")
    pr = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                          start_new_session=True)
    try:
        out, errs = pr.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(pr.pid, signal.SIGKILL)
        out, errs = pr.communicate()
        raise AssertionError("timed out\n" + errs[-2000:])
    assert pr.returncode == 0, out[-2000:] + errs[-3000:]
    res = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res == {"31": True, "51": True}


MOCK_DIR = os.path.join(ROOT, "tests", "mock_rccl")


def mock_rccl_library():
    """tests/mock_rccl/libmockrccl.so: the RCCL entry points over POSIX shared memory (built on demand)."""
    so, src = os.path.join(MOCK_DIR, "libmockrccl.so"), os.path.join(MOCK_DIR, "mock_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-x", "hip",
                               src, "-o", so, "-lrt", "-lpthread"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return so


@pytest.mark.gpu
@pytest.mark.parametrize("world,k,do_fit,P,replicated,dedupe", [(2, 31, False, 0, False, "1"), (3, 51, True, 64, False, "1"), (4, 31, False, 128, False, "0"),
                                                                (3, 31, True, 0, True, "auto"), (3, 31, False, 64, False, ["1", "0", "1"]),
                                                                (2, 89, False, 64, False, "1"),
                                                                (3, 31, False, 64, True, "piece"), (4, 31, False, 64, False, "piece")])
def test_shard_preprocess_several_ranks_over_a_stand_in_transport(world, k, do_fit, P, replicated, dedupe):
    """shk_shard_preprocess with 2, 3 and 4 ranks on the one GPU: the library's own multi-rank code (size exchange,
    plan, pack, the pairwise exchange with its offsets, histogram all-reduce, gather of the solid rows) runs exactly
    as on a node, only the bytes travel through tests/mock_rccl instead of RCCL (which refuses two ranks on one
    device).  Every rank must end with the oracle's bytes.  dedupe: the records cross deduplicated by their sender, with
    weights ("1"), raw ("0"), as the library decides ("auto"), or with ranks that were told different things (they must
    still agree: raw)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_dataset, run_oracle
    g, fq = make_dataset(60000, 40, err=0.01, seed=900 + world)
    piece = 0
    if dedupe == "piece":                                   # every exchange / gather of the library in pieces of 4 KiB (production: 256 MiB)
        piece, dedupe = 4096, "0"
    os.environ["SHK_RCCL_LIBRARY"] = mock_rccl_library()
    try:
        with tempfile.TemporaryDirectory() as d:
            fqp = os.path.join(d, "reads.fq")
            open(fqp, "wb").write(fq)
            cfgp = os.path.join(d, "cfg.json")
            json.dump({"fastq": fqp, "k": k, "min_count": 3, "min_qual": 20, "do_fit": do_fit, "P": P, "replicated": replicated, "dedupe": dedupe, "piece": piece}, open(cfgp, "w"))
            out = os.path.join(d, "res")
            launch(world, ["rccl", out, cfgp], 29760 + world + (10 if replicated else 0) + (20 if isinstance(dedupe, list) else 0) + (40 if k == 89 else 0) + (60 if piece else 0), timeout=300)
            res = [json.load(open(f"{out}.{r}")) for r in range(world)]
    finally:
        os.environ.pop("SHK_RCCL_LIBRARY", None)
    assert not any("skipped" in r for r in res), res
    o = run_oracle([fq], k=k, min_count=3, min_qual=20, do_fit=do_fit)
    o.assemble()
    for r in res:
        assert r["pre"] == o.preprocessing_json() and r["asm"] == o.assembly_json()
        assert r["total_instances"] == o.total_instances
        assert r["timings"]["shard_exchange_sent_MB"] > 0
        assert r["timings"]["shard_records_deduplicated_x1"] == (1.0 if dedupe == "1" else 0.0 if dedupe != "auto" else r["timings"]["shard_records_deduplicated_x1"])


@pytest.mark.gpu
@pytest.mark.parametrize("world,step,bad_rank", [(2, "pass1", 1), (3, "pack", 2), (2, "count", 0), (3, "rows", 1), (2, "keep", 1), (2, "alloc", 1)])
def test_a_failure_on_one_rank_ends_the_collective_call_on_every_rank(world, step, bad_rank):
    """A local failure on ONE rank of shk_shard_preprocess (device memory, a slice or partition that overflows: all
    depend on that rank's share of the reads) must not leave the other ranks blocked in the next collective: the
    failure travels with the next small collective and every rank returns an error.  SHK_FAULT_INJECT makes the
    named local step fail on one process; the launch would time out if a rank hung."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_dataset
    g, fq = make_dataset(30000, 30, err=0.01, seed=77)
    os.environ["SHK_RCCL_LIBRARY"] = mock_rccl_library()
    try:
        with tempfile.TemporaryDirectory() as d:
            fqp = os.path.join(d, "reads.fq")
            open(fqp, "wb").write(fq)
            cfgp = os.path.join(d, "cfg.json")
            json.dump({"fastq": fqp, "k": 31, "min_count": 3, "min_qual": 20, "do_fit": False, "P": 64,
                       "replicated": step == "alloc",        # (that step exists only where the solid set is gathered)
                       "inject": {"rank": bad_rank, "step": step}}, open(cfgp, "w"))
            out = os.path.join(d, "res")
            launch(world, ["rccl", out, cfgp], 29780 + world, timeout=180)
            res = [json.load(open(f"{out}.{r}")) for r in range(world)]
    finally:
        os.environ.pop("SHK_RCCL_LIBRARY", None)
    for r, x in enumerate(res):
        assert "error" in x, (r, x)
        if r == bad_rank:
            assert "injected fault" in x["error"]
        else:
            assert "another rank failed" in x["error"]


@pytest.mark.gpu
@pytest.mark.parametrize("world,dedupe", [(2, "0"), (3, "1")])
def test_a_record_exchange_that_loses_data_is_reported(world, dedupe):
    """Round 3 found a transport that delivered only part of a large block (the rest of the receive buffer kept stale
    bytes): counts were silently wrong.  shk_shard_preprocess now carries the k-mer instances the ranks' READS hold beside
    the instances they COUNTED through its histogram all-reduce; here the stand-in transport truncates every received
    block to 4 KiB — every rank must come back with that error (not hang, not hand out an assembly)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_dataset
    g, fq = make_dataset(60000, 40, err=0.01, seed=78)
    os.environ["SHK_RCCL_LIBRARY"] = mock_rccl_library()
    try:
        with tempfile.TemporaryDirectory() as d:
            fqp = os.path.join(d, "reads.fq")
            open(fqp, "wb").write(fq)
            cfgp = os.path.join(d, "cfg.json")
            json.dump({"fastq": fqp, "k": 31, "min_count": 3, "min_qual": 20, "do_fit": False, "P": 64, "dedupe": dedupe,
                       "truncate": 4096}, open(cfgp, "w"))
            out = os.path.join(d, "res")
            launch(world, ["rccl", out, cfgp], 29890 + world, timeout=180)
            res = [json.load(open(f"{out}.{r}")) for r in range(world)]
    finally:
        os.environ.pop("SHK_RCCL_LIBRARY", None)
    for r, x in enumerate(res):
        assert "error" in x and "lost or duplicated data" in x["error"], (r, x)


@pytest.mark.gpu
@pytest.mark.parametrize("truncate", [0, 4096, -1])
def test_bench_launches_its_own_ranks_and_steps_down_loudly(truncate):
    """`python bench.py --gpus 2` without a launcher: it starts its own two ranks (here both on the one GPU,
    gloo for torch's group, the library's exchange through tests/mock_rccl), prints ONE JSON line and exits 0.  With a
    transport that loses data the library reports it on every rank; the bench then measures the same shk_shard_* pieces with
    torch.distributed's collectives and SAYS so in the line (`fallbacks`, `config.parallelism`) — never a silent switch; when
    that path fails too (injected), every rank assembles an isolate of its own and the line says that."""
    env = dict(os.environ, SHK_RCCL_LIBRARY=mock_rccl_library(), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MOCK_RCCL_TRUNCATE_BYTES", None)
    if truncate > 0:
        env["MOCK_RCCL_TRUNCATE_BYTES"] = str(truncate)
    if truncate < 0:                                        # both sharded paths fail: one isolate per rank, said in the line
        env["BENCH_TEST_FAIL_PATHS"] = "lib,torch"
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-gpu", "--backend", "gloo", "--genome", "300000",
                         "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                        env=env, timeout=400)
    assert pr.returncode == 0, pr.stdout[-1500:] + pr.stderr[-3000:]
    lines = [x for x in pr.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, pr.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0 and line["config"]["ncontigs"] >= 1
    if truncate < 0:
        assert [f["path"] for f in line["fallbacks"]] == ["sharded, collectives=lib", "sharded, collectives=torch"]
        assert "one isolate per rank" in line["config"]["parallelism"] and "FALLBACK" in line["config"]["parallelism"]
        assert line["config"]["ncontigs"] == 1
    elif truncate:
        assert line["fallbacks"][0]["path"] == "sharded, collectives=lib" and "lost or duplicated data" in line["fallbacks"][0]["error"]
        assert "torch.distributed" in line["config"]["parallelism"] and "SECOND PATH" in line["config"]["parallelism"]
        assert "the sharded step failed with collectives=lib" in pr.stderr
    else:
        assert "fallbacks" not in line and "inside the library" in line["config"]["parallelism"]


def test_a_missing_rccl_library_is_an_error_not_a_crash():
    """dlerror() clears its message when read — reading it twice handed std::string a null pointer and the
    process died instead of returning SHK_E_DEVICE.  Runs in a child (the library resolves RCCL once per process)."""
    code = r'''
import os, sys
sys.path.insert(0, os.environ["SHK_ROOT"])
import ctypes as C
from sparrowhawk_amd import _lib
L = _lib.load()
ident = (C.c_uint8 * 128)()
rc = L.shk_comm_unique_id(ident)
msg = L.shk_comm_error().decode()
assert rc != 0 and "SHK_RCCL_LIBRARY" in msg and "nonexistent" in msg, (rc, msg)
assert not L.shk_comm_init(ident, 0, 1)
print("OK", msg)
'''
    env = dict(os.environ, SHK_ROOT=ROOT, SHK_RCCL_LIBRARY="/nonexistent/librccl.so.1")
    pr = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=120)
    assert pr.returncode == 0, pr.stdout[-1500:] + pr.stderr[-1500:]


# ---- the graph phases sharded over the ranks (csrc/shard_graph.h) -----------------------------------------------------
def _write_cases(d, cases):
    out = []
    for i, (fq, params) in enumerate(cases):
        fqp = os.path.join(d, f"reads{i}.fq")
        open(fqp, "wb").write(fq)
        # (records deduplicated by their sender / raw / the library's choice, in turn: dist_worker.set_dedupe)
        out.append(dict({"dedupe": os.environ.get("SHK_DIST_FUZZ_DEDUPE") or ("1", "0", "auto")[i % 3]}, **dict(params, fastq=fqp)))
    cfgp = os.path.join(d, "cfg.json")
    json.dump({"cases": out}, open(cfgp, "w"))
    return cfgp


def _oracle_jsons(fq, params):
    from util import run_oracle
    o = run_oracle([fq], k=params["k"], min_count=params["min_count"], min_qual=params["min_qual"], do_fit=params.get("do_fit", False),
                   no_bubble_collapse=params.get("no_bubble_collapse", False), no_dead_end_removal=params.get("no_dead_end_removal", False))
    o.assemble()
    return o.preprocessing_json(), o.assembly_json()


def _graph_cases(seed, n, first_case=0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle import _random_graph_case
    rng = np.random.default_rng(seed)
    cases = []
    for case in range(first_case, first_case + n):
        fq, k, min_count, flags = _random_graph_case(rng, case)
        cases.append((fq, dict(k=k, min_count=min_count, min_qual=0, **flags)))
    return cases


@pytest.mark.gpu
def test_sharded_graph_one_rank_random_graphs():
    """The sharded assembly (local contraction -> stitching -> tips / bubbles on the UNITIG graph -> emission) with a
    one-rank RCCL communicator, in this process: 120 small random graphs with errors, repeats, hairpins, plasmids and
    tandem rings (the inputs that pin the oracle in test_oracle.py) must give the oracle's bytes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from sparrowhawk_amd import AssemblyHelper, pack_fastq
    from sparrowhawk_amd.dist import LibComm, sharded_preprocess_rccl
    dev = torch.device("cuda", 0)
    comm = LibComm(0, 1)
    try:
        for fq, pr in _graph_cases(8100, 120):
            bases, seg, nb, nr = pack_fastq(fq, pr["k"], 0)
            d_bases = torch.from_numpy(bases.view(np.int32)).to(dev)
            d_seg = torch.from_numpy(seg.view(np.int32)).to(dev)
            torch.cuda.synchronize()
            h = AssemblyHelper.new(pr["k"], False, pr["min_count"], 0, 0, False, False, pr["no_bubble_collapse"], pr["no_dead_end_removal"])
            sharded_preprocess_rccl(h, d_bases.data_ptr(), d_seg.data_ptr(), len(seg) - 1, nb, nr, comm)
            h.assemble()
            pre, asm = _oracle_jsons(fq, pr)
            assert h.get_preprocessing_info() == pre
            assert h.get_assembly() == asm, pr
            assert "shard_graph_stitch" in h.timings()
            h.free()
    finally:
        comm.free()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_graph_several_ranks(world):
    """The graph sharded over 2, 3 and 4 ranks (on the one GPU, bytes through tests/mock_rccl): neighbour queries and
    answers across ranks, half links, local chains stitched across ranks, rings that span ranks, tips and bubbles on
    the unitig graph, every rank emitting its own bases.  30 random small graphs, a circular chromosome with two
    plasmids, an error-rich linear genome with the fitted threshold and a two-word k: every rank ends with the
    oracle's bytes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_dataset
    from sparrowhawk_amd import synth
    # (SHK_DIST_FUZZ_CASES / SHK_DIST_FUZZ_SEED: a longer campaign of the random graphs — tools/fuzz_sharded.sh)
    cases = _graph_cases(int(os.environ.get("SHK_DIST_FUZZ_SEED", 8200)) + world, int(os.environ.get("SHK_DIST_FUZZ_CASES", 30)), first_case=world)
    # a circular chromosome + plasmids (rings across ranks), clean and with errors
    for seed, err, k, mc in ((31, 0.0, 31, 3), (32, 0.01, 31, 2), (33, 0.005, 51, 2)):
        chrom = synth.random_genome(60000, seed)
        p1, p2 = synth.random_genome(5000, seed + 100), synth.random_genome(700, seed + 200)
        fqs = []
        for j, (g, cov) in enumerate(((chrom, 40), (p1, 60), (p2, 80))):
            codes, quals = synth.sample_reads(g, len(g) * cov // 150, 150, seed * 10 + j, err=err, circular=True)
            fqs.append(synth.to_fastq(codes, quals))
        # (the worker deals the records out by rank, cutting the text at "@r")
        recs = []
        for fqx in fqs:
            recs.extend(fqx.decode().split("@r")[1:])
        fq = ("@r" + "@r".join(recs)).encode()
        cases.append((fq, dict(k=k, min_count=mc, min_qual=20)))
    # a small metagenome (configs[4] in miniature): 60 genomes of 4-12 kbp at log-normal abundances, 0.5 % errors, min_count 2 —
    # hundreds of unitigs, tips and bubbles in the unitig graph, low-coverage genomes falling apart
    rng = np.random.default_rng(8250 + world)
    recs = []
    for gi in range(60):
        gl = int(rng.integers(4000, 12001))
        gm = synth.random_genome(gl, 9000 + gi)
        cov = float(np.exp(rng.normal(np.log(12.0), 1.0)))
        nr = max(1, int(gl * cov / 150))
        codes, quals = synth.sample_reads(gm, nr, 150, 9100 + gi, err=0.005, circular=bool(gi % 3 == 0))
        recs.extend(synth.to_fastq(codes, quals).decode().split("@r")[1:])
    cases.append((("@r" + "@r".join(recs)).encode(), dict(k=31, min_count=2, min_qual=0)))
    g, fq = make_dataset(150000, 40, err=0.01, seed=8300 + world)
    cases.append((fq, dict(k=31, min_count=3, min_qual=20, do_fit=True)))
    g, fq = make_dataset(50000, 30, err=0.01, seed=8400 + world)
    cases.append((fq, dict(k=63, min_count=2, min_qual=0, P=64)))
    g, fq = make_dataset(40000, 30, err=0.005, seed=8450 + world, circular=True)       # three- and four-word keys
    cases.append((fq, dict(k=89, min_count=2, min_qual=20)))
    cases.append((fq, dict(k=127, min_count=1, min_qual=0)))
    g, fq = make_dataset(40000, 30, read_len=400, err=0.003, seed=8460 + world, circular=True)       # six- and eight-word keys (k up to 255)
    cases.append((fq, dict(k=161, min_count=2, min_qual=0)))
    cases.append((fq, dict(k=255, min_count=1, min_qual=0)))
    # a genome of k + 5 bases: six solid k-mers under one or two minimisers, so some rank owns NO solid k-mer and still has
    # to answer neighbour queries (empty graph tables), join every exchange and agree on the stitching (found by the
    # 250-case campaign at 4 ranks)
    tiny = synth.random_genome(46, 8470 + world)
    codes, quals = synth.sample_reads(tiny, 12, 46, 8471 + world, err=0.0, circular=False)
    cases.append((synth.to_fastq(codes, quals), dict(k=41, min_count=2, min_qual=0)))
    i_tiny = len(cases) - 1
    # degenerate inputs on several ranks (test_empty_and_degenerate_inputs has them on one GPU): nothing solid anywhere, a
    # single k-mer, a homopolymer (one node with a self-loop), one read shorter than k — with fewer records than ranks, so
    # that some ranks are handed no read at all
    cases.append((synth.to_fastq(codes, quals), dict(k=41, min_count=50, min_qual=0)))
    one = "ACGTTGCATGCCGATAGCTAGCTAGGATCCA"
    cases.append(((f"@r0\n{one}\n+\n{'I' * 31}\n" * 3).encode(), dict(k=31, min_count=1, min_qual=0)))
    cases.append(((f"@r0\n{'A' * 60}\n+\n{'I' * 60}\n" * 2).encode(), dict(k=31, min_count=0, min_qual=0)))
    cases.append((b"@r0\nACGTACGT\n+\nIIIIIIII\n", dict(k=31, min_count=0, min_qual=0)))
    os.environ["SHK_RCCL_LIBRARY"] = mock_rccl_library()
    # the stand-in transport at its most hostile for 3 and 4 ranks (random per-peer delays, out-of-order completion across
    # peers, eager sends, no barrier anywhere: tests/mock_rccl/mock_rccl.cpp), plain for 2; MOCK_RCCL_JITTER overrides
    jitter_was = os.environ.get("MOCK_RCCL_JITTER")
    if jitter_was is None:
        os.environ["MOCK_RCCL_JITTER"] = "1" if world >= 3 else "0"
    try:
        with tempfile.TemporaryDirectory() as d:
            cfgp = _write_cases(d, cases)
            out = os.path.join(d, "res")
            launch(world, ["rccl_many", out, cfgp], 29800 + world, timeout=420 + 3 * int(os.environ.get("SHK_DIST_FUZZ_CASES", 30)))
            res = [json.load(open(f"{out}.{r}")) for r in range(world)]
    finally:
        os.environ.pop("SHK_RCCL_LIBRARY", None)
        if jitter_was is None:
            os.environ.pop("MOCK_RCCL_JITTER", None)
    for i, (fq, pr) in enumerate(cases):
        pre, asm = _oracle_jsons(fq, pr)
        for r in range(world):
            assert "error" not in res[r][i], (i, r, res[r][i])
            assert res[r][i]["pre"] == pre, (i, r)
            assert res[r][i]["asm"] == asm, (i, r, pr)
        if i == i_tiny and world >= 3:
            assert min(res[r][i]["n_solid_local"] for r in range(world)) == 0      # a rank without a single solid k-mer
        if i >= 30:
            assert sum(res[r][i]["n_solid_local"] for r in range(world)) == json.loads(pre)["nkmers"]      # nobody holds the whole set


@pytest.mark.gpu
def test_sharded_graph_work_per_rank_falls_with_the_rank_count():
    """'per-rank graph + collapse kernel time falls ~ 1/N'.  A 3 Mbp genome at 30x through 1 and 4 ranks.
    What can be asserted on a ONE-GPU box: every rank holds about a quarter of the rows (nobody holds the solid set),
    the local chains are cut at rank edges (about one node in ten ends one) and the results are identical.  The four
    ranks SHARE the one GPU here, so their node-level kernels run interleaved and a rank's elapsed kernel time is
    inflated by up to the number of ranks: it is printed, and only bounded loosely (measured: 0.87 ms alone, 1.2 ms
    elapsed per rank with four ranks on the card, i.e. ~0.3 ms of GPU time each)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_dataset
    g, fq = make_dataset(3000000, 30, seed=8500)
    pr = dict(k=31, min_count=3, min_qual=20, timings=True)
    keys = ("graph_table_kernel", "adjacency_kernel", "collapse_rank_device")        # (the last one: simple links to ranked chains, one launch sequence)
    os.environ["SHK_RCCL_LIBRARY"] = mock_rccl_library()
    per_world = {}
    try:
        for world in (1, 4):
            with tempfile.TemporaryDirectory() as d:
                cfgp = _write_cases(d, [(fq, pr), (fq, pr)])          # (twice: the second run is warm)
                out = os.path.join(d, "res")
                launch(world, ["rccl_many", out, cfgp], 29810 + world, timeout=420)
                per_world[world] = [json.load(open(f"{out}.{r}")) for r in range(world)]
    finally:
        os.environ.pop("SHK_RCCL_LIBRARY", None)
    assert per_world[1][0][1]["asm"] == per_world[4][0][1]["asm"] == per_world[4][3][1]["asm"]
    t1 = sum(per_world[1][0][1]["timings"][k] for k in keys)
    t4 = [sum(per_world[4][r][1]["timings"][k] for k in keys) for r in range(4)]
    rows = [per_world[4][r][1]["n_solid_local"] for r in range(4)]
    chains = [per_world[4][r][1]["timings"]["shard_graph_local_chains_x1e-3"] * 1e3 for r in range(4)]
    print("node-level kernel ms: one rank", t1, "four ranks sharing the GPU", t4, "rows", rows, "local chains", chains)
    assert max(rows) < 0.3 * sum(rows), rows
    assert all(0.03 * r < c < 0.25 * r for r, c in zip(rows, chains)), (rows, chains)
    # (GPU time per rank ~ elapsed / ranks on the card; the ranks are not in step — the stand-in transport has no barrier — so
    # a rank's elapsed stage time also holds what it waited for the card: loose bound)
    assert max(t4) / 4 < 0.6 * t1, (t1, t4)
