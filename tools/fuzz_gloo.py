"""The multi-rank random-graph campaign through torch.distributed collectives (gloo) and the five shk_shard_* pieces — no stand-in
transport, none of the library's own collective code.  Usage: python tools/fuzz_gloo.py <seed> <n_cases> <world>"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dist
seed, n, world = (int(x) for x in sys.argv[1:4])
cases = test_dist._graph_cases(seed + world, n, first_case=world)
with tempfile.TemporaryDirectory() as d:
    cfgp = test_dist._write_cases(d, cases)
    out = os.path.join(d, "res")
    test_dist.launch(world, ["gloo_many", out, cfgp], 29970, timeout=280)
    res = [json.load(open(f"{out}.{r}")) for r in range(world)]
bad = 0
for i, (fq, pr) in enumerate(cases):
    pre, asm = test_dist._oracle_jsons(fq, pr)
    ok = all("error" not in res[r][i] and res[r][i]["pre"] == pre and res[r][i]["asm"] == asm for r in range(world))
    if not ok:
        bad += 1
        print("case", i, pr, [("error" in res[r][i] and res[r][i]["error"][:60]) or (res[r][i]["pre"] == pre, res[r][i]["asm"] == asm) for r in range(world)], flush=True)
print("%d of %d cases differ (world %d, torch collectives)" % (bad, n, world))
