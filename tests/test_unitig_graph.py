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
    # (the inputs of test_graph_stages_against_a_brute_force_python_graph; SHK_UG_FUZZ_SEED: a longer campaign with other seeds)
    rng = np.random.default_rng(int(os.environ.get("SHK_UG_FUZZ_SEED", 7000)) + block)
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


def test_large_unitig_graph_takes_the_threaded_passes():
    """A metagenome leaves millions of unitig records: from 65 536 records on, the sorting of the ends, the lookup of every
    record's out-neighbours and the search for a round's candidates run on several host threads.  70 000 synthetic unitigs
    (both strands: 140 000 records): most isolated, 10 000 pairs joined by a simple link (they must merge), 2 000 forks
    with a dead-end branch of 5 nodes beside a long one (the short one is a tip and must go)."""
    L = _lib.load()
    k, W = 31, 1
    rng = np.random.default_rng(5)

    def rnd(n):
        return "".join(rng.choice(list("ACGT"), n))
    units = []                                              # (first k-mer, last k-mer, nodes, kc) of one strand each
    expect = {}                                             # frozenset of unit indices -> nodes of the contig
    for i in range(46000):                                  # isolated
        units.append((rnd(k), rnd(k), 40, 400))
        expect[frozenset([len(units) - 1])] = 40
    for i in range(10000):                                  # A -> B simple link: one contig of 30 + 50 nodes
        x = rnd(k - 1)
        units.append((rnd(k), rnd(1) + x, 30, 300)); a = len(units) - 1
        units.append((x + rnd(1), rnd(k), 50, 500)); b = len(units) - 1
        expect[frozenset([a, b])] = 80
    for i in range(2000):                                   # J has two in-edges: a long chain and a 5-node dead end (a tip: removed)
        x = rnd(k - 2)
        j_first = rnd(1) + x + rnd(1)                        # J's first k-mer; predecessors end with ? + j_first[:-1]
        pre = j_first[:-1]
        units.append((j_first, rnd(k), 100, 1000)); j = len(units) - 1
        units.append((rnd(k), "A" + pre, 200, 2000)); long_ = len(units) - 1
        units.append((rnd(k), "C" + pre, 5, 50)); tip = len(units) - 1
        expect[frozenset([long_, j])] = 300                 # after the tip is gone the link long -> J is simple
    n = 2 * len(units)
    first = np.zeros((n, W), dtype=np.uint64); last = np.zeros((n, W), dtype=np.uint64)
    ln = np.zeros(n, dtype=np.uint64); kc = np.zeros(n, dtype=np.uint64); circ = np.zeros(n, dtype=np.uint8)
    for u, (f, l, nn, c) in enumerate(units):
        first[2 * u] = kmer_words(f, W); last[2 * u] = kmer_words(l, W)
        first[2 * u + 1] = kmer_words(rc(l), W); last[2 * u + 1] = kmer_words(rc(f), W)
        ln[2 * u] = ln[2 * u + 1] = nn; kc[2 * u] = kc[2 * u + 1] = c
    ptr = L.shk_host_unitig_assemble(k, n, first.ctypes.data, last.ctypes.data, ln.ctypes.data, kc.ctypes.data, circ.ctypes.data,
                                     None, None, None, 1, 1)
    assert ptr
    text = C.string_at(ptr).decode()
    L.shk_host_free(ptr)
    assert not text.startswith("error:"), text[:200]
    lines = text.strip().split("\n")
    assert lines[0] == "removed %d 0" % (2000 * 5)
    got = {}
    for line in lines[1:]:
        head, ids = line.split(":")
        ring, rot, nodes, kc_ = (int(x) for x in head.split())
        assert ring == 0
        got[frozenset(int(r) // 2 for r in ids.split())] = nodes
    assert got == expect
