"""CPU tests of csrc/unitig_graph.cpp — SPEC S9 / S10 run on UNITIG records, the host stage of the sharded assembly
(every rank corrects the small graph of unitigs identically once the GPUs have contracted the k-mer graph).

The records come from tests/pygraph.py (the uncorrected graph's maximal chains, both strands, rings cut anywhere);
the contigs the library builds from them must equal the ORACLE's contigs for the same reads — tips, bubbles, rings
formed by removals, hairpins and all."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pygraph import PyGraph, rc
from sparrowhawk_amd import _lib
from test_oracle import _random_graph_case
from util import parse_fastq, revcomp, run_oracle

CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def kmer_words(s, W):
    v = 0
    for ch in s:
        v = (v << 2) | CODE[ch]
    return [(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(W)]


def strand_records(pg):
    """every maximal chain of simple links of pg's CURRENT graph, on both strands; rings cut at an arbitrary node"""
    seen, recs = set(), []
    nodes = list(pg.oriented_nodes())
    for v in nodes:
        if pg.simple_pred(v) is None:
            chain, cur = [v], v
            while True:
                n = pg.simple_succ(cur)
                if n is None:
                    break
                chain.append(n)
                cur = n
            seen.update(chain)
            recs.append((chain, False))
    for v in nodes:
        if v in seen:
            continue
        cyc, cur = [v], v
        while True:
            cur = pg.simple_succ(cur)
            if cur == v:
                break
            cyc.append(cur)
        seen.update(cyc)
        recs.append((cyc, True))
    return recs


def library_contigs(L, pg, k, tips, bubbles, drop_mirror_of_short_rings=False):
    W = (2 * k + 63) // 64
    recs = strand_records(pg)
    if drop_mirror_of_short_rings:
        # the device reports a ring without a sampled node on ONE strand only: the one spelled from (xmin, 0)
        keep = []
        for chain, circ in recs:
            if circ and len(chain) < 40:
                m = min(chain)
                if m[1] != 0:
                    continue
                i = chain.index(m)
                chain = chain[i:] + chain[:i]
            keep.append((chain, circ))
        recs = keep
    n = len(recs)
    first = np.zeros((n, W), dtype=np.uint64); last = np.zeros((n, W), dtype=np.uint64)
    ln = np.zeros(n, dtype=np.uint64); kc = np.zeros(n, dtype=np.uint64); circ = np.zeros(n, dtype=np.uint8)
    mk = np.zeros((n, W), dtype=np.uint64); mo = np.zeros(n, dtype=np.uint8); mp = np.zeros(n, dtype=np.uint64)
    for r, (chain, c) in enumerate(recs):
        first[r] = kmer_words(pg.seq(chain[0]), W); last[r] = kmer_words(pg.seq(chain[-1]), W)
        ln[r] = len(chain); kc[r] = sum(pg.count[u[0]] for u in chain); circ[r] = int(c)
        m = min(chain)
        mk[r] = kmer_words(m[0], W); mo[r] = m[1]; mp[r] = chain.index(m)
    ptr = L.shk_host_unitig_assemble(k, n, first.ctypes.data, last.ctypes.data, ln.ctypes.data, kc.ctypes.data, circ.ctypes.data,
                                     mk.ctypes.data, mo.ctypes.data, mp.ctypes.data, int(tips), int(bubbles))
    assert ptr, "shk_host_unitig_assemble failed"
    text = C.string_at(ptr).decode()
    L.shk_host_free(ptr)
    assert not text.startswith("error:"), text
    lines = text.strip().split("\n")
    removed = tuple(int(x) for x in lines[0].split()[1:])
    out = []
    for line in lines[1:]:
        head, ids = line.split(":")
        ring, rot, ln_, kc_ = (int(x) for x in head.split())
        path = []
        for r in ids.split():
            path.extend(recs[int(r)][0])
        assert len(path) == ln_
        if ring:
            path = path[rot:] + path[:rot]
        s = pg.seq(path[0]) + "".join(pg.seq(u)[-1] for u in path[1:])
        out.append((min(s, rc(s)), kc_))
        assert kc_ == sum(pg.count[u[0]] for u in path)
    out.sort(key=lambda t: (-len(t[0]), t[0]))
    return out, removed


@pytest.mark.parametrize("block", range(8))
def test_unitig_level_correction_equals_the_oracle(block):
    L = _lib.load()
    rng = np.random.default_rng(7000 + block)                # the inputs of test_graph_stages_against_a_brute_force_python_graph
    n_removed = 0
    for case in range(block * 40, block * 40 + 40):
        fq, k, min_count, flags = _random_graph_case(rng, case)
        counts = {}
        for rd, _q in parse_fastq(fq):
            for i in range(len(rd) - k + 1):
                s = rd[i:i + k]
                r = revcomp(s)
                x = s if s < r else r
                counts[x] = counts.get(x, 0) + 1
        pg = PyGraph(counts, k, min_count)                   # the UNCORRECTED graph: the library corrects it at unitig level
        o = run_oracle([fq], k=k, min_count=min_count, min_qual=0, **flags)
        o.assemble()
        got, removed = library_contigs(L, pg, k, not flags["no_dead_end_removal"], not flags["no_bubble_collapse"],
                                       drop_mirror_of_short_rings=bool(case % 2))
        want = list(zip(o.contigs(), [int(x) for x in o.contig_kc()]))
        assert got == want, f"case {case}: contigs differ (k={k}, {len(got)} vs {len(want)})"
        assert removed == (o.tips_removed, o.bubbles_removed), f"case {case}"
        n_removed += sum(removed)
    assert n_removed > 0
