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


def launch(nproc, args, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER] + args
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


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
