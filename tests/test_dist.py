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
@pytest.mark.parametrize("world,k,do_fit,P", [(2, 31, False, 0), (3, 51, True, 64), (4, 31, False, 128)])
def test_shard_preprocess_several_ranks_over_a_stand_in_transport(world, k, do_fit, P):
    """shk_shard_preprocess with 2, 3 and 4 ranks on the one GPU: the library's own multi-rank code (size exchange,
    plan, pack, the pairwise exchange with its offsets, histogram all-reduce, gather of the solid rows) runs exactly
    as on a node, only the bytes travel through tests/mock_rccl instead of RCCL (which refuses two ranks on one
    device).  Every rank must end with the oracle's bytes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_dataset, run_oracle
    g, fq = make_dataset(60000, 40, err=0.01, seed=900 + world)
    os.environ["SHK_RCCL_LIBRARY"] = mock_rccl_library()
    try:
        with tempfile.TemporaryDirectory() as d:
            fqp = os.path.join(d, "reads.fq")
            open(fqp, "wb").write(fq)
            cfgp = os.path.join(d, "cfg.json")
            json.dump({"fastq": fqp, "k": k, "min_count": 3, "min_qual": 20, "do_fit": do_fit, "P": P}, open(cfgp, "w"))
            out = os.path.join(d, "res")
            launch(world, ["rccl", out, cfgp], 29760 + world, timeout=300)
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


@pytest.mark.gpu
@pytest.mark.parametrize("world,step,bad_rank", [(2, "pass1", 1), (3, "pack", 2), (2, "count", 0), (3, "rows", 1), (2, "alloc", 1)])
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
