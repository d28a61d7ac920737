"""CPU tests that pin the oracle (parity with upstream is unpinned — SPEC.md — so these pin it
to SPEC.md through independent pure-Python restatements, closed-form cases and hand-built
graphs whose answers follow by reasoning)."""
import gzip
import json

import numpy as np
import pytest

import cases
from oracle import Oracle, oracle_fit
from sparrowhawk_amd import synth
from util import (canonical_int, int_to_words, make_dataset, parse_fastq, py_count, revcomp,
                  run_oracle)


def oracle_contigs(fq, **kw):
    o = run_oracle([fq], **kw)
    o.assemble()
    return o, set(o.contigs())


@pytest.mark.parametrize("k", [15, 31, 33, 51, 63])
def test_count_matches_python_dict(k):
    g, fq = make_dataset(3000, 12, read_len=100, err=0.01, seed=k)
    ref = py_count(parse_fastq(fq), k, min_qual=20)
    for naive in (True, False):
        o = Oracle(k=k, min_count=0, min_qual=20)
        o.add_fastq(fq)
        o.count(naive=naive)
        keys, cnt = o.distinct()
        W = (2 * k + 63) // 64
        got = {tuple(int(x) for x in keys[i]): int(cnt[i]) for i in range(len(cnt))}
        want = {int_to_words(v, W): c for v, c in ref.items()}
        assert got == want
        assert o.total_instances == sum(ref.values())
        h = o.histo()
        assert int(h.sum()) == len(ref)
        assert int((h * np.arange(1, 501, dtype=np.uint64)).sum()) == sum(ref.values())
        # sorted ascending as integers
        ints = [sum(int(keys[i, j]) << (64 * j) for j in range(W)) for i in range(len(cnt))]
        assert ints == sorted(ints)


def test_closed_form_single_contig():
    """Error-free reads from a repeat-free genome at high coverage: exactly one contig, a
    substring of the genome (ends trimmed by the count filter); node counts equal coverage."""
    g, fq = make_dataset(20000, 40, seed=3)
    o, contigs = oracle_contigs(fq, k=31, min_count=3, min_qual=20)
    assert len(contigs) == 1
    c = next(iter(contigs))
    gs = synth.codes_to_str(g)
    assert c in gs or revcomp(c) in gs
    assert len(c) > 19900
    # independent coverage check
    ref = py_count(parse_fastq(fq), 31)
    keys, cnt = o.solid()
    assert all(ref[int(keys[i, 0])] == int(cnt[i]) for i in range(0, len(cnt), 97))
    assert o.contig_kc()[0] == int(cnt.sum())


def test_tip_case():
    c = cases.tip_case()
    _, got = oracle_contigs(c["fastq"], k=c["k"], min_count=c["min_count"], min_qual=0)
    assert got == c["with_removal"]
    _, got = oracle_contigs(c["fastq"], k=c["k"], min_count=c["min_count"], min_qual=0, no_dead_end_removal=True)
    assert got == c["without_removal"]


def test_bubble_case():
    c = cases.bubble_case()
    _, got = oracle_contigs(c["fastq"], k=c["k"], min_count=c["min_count"], min_qual=0)
    assert got == c["with_collapse"]
    _, got = oracle_contigs(c["fastq"], k=c["k"], min_count=c["min_count"], min_qual=0, no_bubble_collapse=True)
    assert got == c["without_collapse"]


def test_cycle_case():
    c = cases.cycle_case()
    o, got = oracle_contigs(c["fastq"], k=c["k"], min_count=c["min_count"], min_qual=0)
    assert got == c["expect"]
    # a circular unitig links to itself in the GFA
    assert "L\t1\t+\t1\t+\t14M" in o.gfa1()


def test_palindrome_case():
    c = cases.palindrome_case()
    _, got = oracle_contigs(c["fastq"], k=c["k"], min_count=c["min_count"], min_qual=0)
    assert got == c["expect"]


def test_quality_mask():
    c = cases.qual_mask_case()
    M, cut, k = c["M"], c["cut"], c["k"]
    o = run_oracle([c["fastq"]], k=k, min_count=0, min_qual=20)
    want = set()
    for seg in (M[:cut], M[cut + 1:]):
        for i in range(len(seg) - k + 1):
            want.add(canonical_int(seg[i:i + k]))
    keys, cnt = o.distinct()
    assert {int(x) for x in keys[:, 0]} == want
    assert set(cnt.tolist()) == {2}
    # min_qual=0 accepts the base
    o0 = run_oracle([c["fastq"]], k=k, min_count=0, min_qual=0)
    assert o0.total_instances == 2 * (len(M) - k + 1)


def test_gzip_paired_and_order_invariance():
    g, fq = make_dataset(5000, 20, seed=5, err=0.005)
    recs = fq.decode().split("@r")[1:]
    half = len(recs) // 2
    f1 = ("@r" + "@r".join(recs[:half])).encode()
    f2 = ("@r" + "@r".join(recs[half:])).encode()
    a = run_oracle([fq], k=31, min_count=2); a.assemble()
    b = run_oracle([gzip.compress(f1), gzip.compress(f2[:len(f2) // 2]) + gzip.compress(f2[len(f2) // 2:])],
                   k=31, min_count=2)
    # multi-member gzip may split inside a record only if both members concatenate to valid text
    b.assemble()
    assert a.assembly_json() == b.assembly_json()
    # reverse-complementing reads and shuffling them changes nothing
    rng = np.random.default_rng(0)
    lines = fq.decode().strip().split("\n")
    reads = [(lines[i + 1], lines[i + 3]) for i in range(0, len(lines), 4)]
    perm = rng.permutation(len(reads))
    out = []
    for j, i in enumerate(perm):
        s, q = reads[i]
        if j % 2:
            s, q = revcomp(s), q[::-1]
        out.append(f"@x{j}\n{s}\n+\n{q}\n")
    c = run_oracle(["".join(out).encode()], k=31, min_count=2); c.assemble()
    assert a.assembly_json() == c.assembly_json()


def test_parse_errors():
    o = Oracle(k=31)
    with pytest.raises(ValueError):
        o.add_fastq(b"@r\nACGT\n+\nIII\n")           # length mismatch
    with pytest.raises(ValueError):
        o.add_fastq(b"r\nACGT\n+\nIIII\n")           # no '@'
    with pytest.raises(ValueError):
        o.add_fastq(b"@r\nACGT\n+\n")                # truncated
    o.add_fastq(b"@r\r\nACGT\r\n+\r\nIIII\r\n\n")     # CRLF + trailing blank line is fine
    assert o.n_reads == 1 and o.n_bases == 4


def test_fit_bimodal_spectrum():
    # error peak at 1, coverage peak near 40
    h = np.zeros(500, dtype=np.uint64)
    from math import exp, lgamma, log
    for c in range(1, 200):
        h[c - 1] = int(50000 * exp(c * log(1.0) - 1.0 - lgamma(c + 1))) + int(100000 * exp(c * log(40.0) - 40.0 - lgamma(c + 1)))
    ok, v = oracle_fit(h)
    assert ok and 3 <= v <= 15
    # no coverage peak -> fit fails
    h2 = np.zeros(500, dtype=np.uint64); h2[0] = 1000; h2[1] = 300
    ok2, _ = oracle_fit(h2)
    assert not ok2
    ok3, _ = oracle_fit(np.zeros(500, dtype=np.uint64))
    assert not ok3


def test_do_fit_in_pipeline():
    g, fq = make_dataset(20000, 40, seed=8, err=0.01)
    o = run_oracle([fq], k=31, min_count=5, do_fit=True)
    assert o.fit_ok and 1 <= o.used_min_count <= 30
    info = json.loads(o.preprocessing_json())
    assert info["used_min_count"] == o.used_min_count and len(info["histo"]) == 500
    keys, cnt = o.solid()
    assert info["nkmers"] == len(cnt) and (cnt > o.used_min_count).all()


def test_empty_and_short_inputs():
    o = run_oracle([b""], k=31); o.assemble()
    assert json.loads(o.assembly_json())["ncontigs"] == 0
    o = run_oracle([b"@r\nACGTACGT\n+\nIIIIIIII\n"], k=31, min_count=0); o.assemble()
    assert o.total_instances == 0 and o.contigs() == []
    o = run_oracle([b"@r\n" + b"N" * 100 + b"\n+\n" + b"I" * 100 + b"\n"], k=31, min_count=0); o.assemble()
    assert o.total_instances == 0


def test_output_formats():
    c = cases.bubble_case()
    o, _ = oracle_contigs(c["fastq"], k=c["k"], min_count=0, min_qual=0, no_bubble_collapse=True)
    gfa = o.gfa1().split("\n")
    assert gfa[0] == "H\tVN:Z:1.0"
    S = [l for l in gfa if l.startswith("S\t")]
    L = [l for l in gfa if l.startswith("L\t")]
    assert len(S) == 4 and len(L) == 4               # two forks x two branches
    assert all(l.endswith("\t14M") for l in L)
    fa = o.fasta().split("\n")
    assert fa[0].startswith(">contig_1 len=") and " kc=" in fa[0]
    lens = [int(l.split("len=")[1].split()[0]) for l in fa if l.startswith(">")]
    assert lens == sorted(lens, reverse=True)
    g2 = o.gfa2().split("\n")
    assert g2[0] == "H\tVN:Z:2.0" and sum(l.startswith("E\t") for l in g2) == 4
    assert o.dot().startswith("digraph sparrowhawk {\n") and o.dot().endswith("}\n")
    j = json.loads(o.assembly_json())
    assert list(j.keys()) == ["outfasta", "ncontigs", "outdot", "outgfa", "outgfav2"]
    assert j["outfasta"] == o.fasta() and j["ncontigs"] == 4
