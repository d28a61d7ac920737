"""Shared helpers for the parity tests: pure-Python restatements used to pin the oracle,
and the stage-by-stage product-vs-oracle comparison."""
import json

import numpy as np

from oracle import Oracle
from sparrowhawk_amd import synth

COMP = str.maketrans("ACGT", "TGCA")
CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def revcomp(s):
    return s.translate(COMP)[::-1]


def kmer_int(s):
    v = 0
    for ch in s:
        v = (v << 2) | CODE[ch]
    return v


def canonical_int(s):
    return min(kmer_int(s), kmer_int(revcomp(s)))


def int_to_words(v, W):
    return tuple((v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(W))


def py_count(reads, k, min_qual=0):
    """Independent dict-based canonical k-mer count over (seq, qual) pairs (SPEC S2-S4)."""
    counts = {}
    for seq, qual in reads:
        seq = seq.upper()
        ok = [(c in CODE) and (qual is None or (qual[i] - 33) >= min_qual) for i, c in enumerate(seq)]
        for i in range(len(seq) - k + 1):
            if all(ok[i:i + k]):
                c = canonical_int(seq[i:i + k])
                counts[c] = counts.get(c, 0) + 1
    return counts


def parse_fastq(data: bytes):
    lines = data.decode().split("\n")
    out = []
    i = 0
    while i + 3 < len(lines) + 1 and i < len(lines):
        if lines[i] == "":
            i += 1
            continue
        out.append((lines[i + 1].rstrip("\r"), lines[i + 3].rstrip("\r").encode()))
        i += 4
    return out


NT_SEED = {"A": 0x3c8bfbb395c60474, "C": 0x3193c18562a02b4c, "G": 0x20323ed082572324, "T": 0x295549f54be24456}
M64 = (1 << 64) - 1


def rol(v, s):
    s %= 64
    return ((v << s) | (v >> (64 - s))) & M64 if s else v


def py_nthash(s):
    """Canonical ntHash of s by the non-rolling definition (SPEC S3)."""
    m = len(s)
    fh = 0
    for i, ch in enumerate(s):
        fh ^= rol(NT_SEED[ch], m - 1 - i)
    rh = 0
    for i, ch in enumerate(revcomp(s)):
        rh ^= rol(NT_SEED[ch], m - 1 - i)
    return min(fh, rh)


def py_fit(histo, min_count_fallback=None, iters=200):
    """SPEC S6 written from the text of the spec, in Python floats (IEEE binary64) with math.lgamma/log/exp —
    independent of csrc/fit.cpp and oracle/shk_oracle.c (which share their loop structure).  Vectorless on
    purpose: sums run in ascending c like the spec's, so ties in the last bit cannot flip an integer result.
    Returns (ok, used_min_count) — used_min_count is None when the fit fails."""
    from math import exp, lgamma, log
    h = [float(int(x)) for x in histo]
    assert len(h) == 500
    tot = sum(h)
    if tot == 0.0:
        return False, min_count_fallback
    den = sum(h[c - 1] for c in range(2, 501))
    lam = (sum(c * h[c - 1] for c in range(2, 501)) / den) if den > 0 else 2.0
    lam = max(2.0, lam)
    w = 0.5
    lg = [lgamma(c + 1.0) for c in range(0, 501)]

    def logp(c, mean, logmean):
        return c * logmean - mean - lg[c]

    def exp(x, _e=exp):                     # IEEE semantics: overflow is +inf (C's exp), not an exception
        try:
            return _e(x)
        except OverflowError:
            return float("inf")

    for _ in range(iters):
        l1w, lw, llam = log(1.0 - w), log(w), log(lam)
        sw = sn = sd = 0.0
        for c in range(1, 501):
            hc = h[c - 1]
            if hc == 0.0:
                continue
            r = 1.0 / (1.0 + exp((l1w + logp(c, lam, llam)) - (lw + logp(c, 1.0, 0.0))))
            sw += hc * r
            sn += hc * (1.0 - r) * c
            sd += hc * (1.0 - r)
        w = min(max(sw / tot, 1e-9), 1.0 - 1e-9)
        if sd > 0.0:
            lam = sn / sd
        lam = max(lam, 1.000001)
    if lam < 2.5:
        return False, min_count_fallback
    l1w, lw, llam = log(1.0 - w), log(w), log(lam)
    for c in range(2, 501):
        if l1w + logp(c, lam, llam) > lw + logp(c, 1.0, 0.0):
            return True, min(max(c - 1, 1), 30)
    return False, min_count_fallback


def sorted_table(keys, cnt):
    """Sort rows of (keys[n,W], cnt[n]) by key, most significant word first."""
    if len(cnt) == 0:
        return keys, cnt, np.zeros(0, dtype=np.int64)
    W = keys.shape[1]
    order = np.lexsort([keys[:, j] for j in range(W)])      # last key = most significant
    return keys[order], cnt[order], order


def make_dataset(genome_len, coverage, read_len=150, err=0.0, seed=1, circular=False):
    g = synth.random_genome(genome_len, seed)
    n_reads = genome_len * coverage // read_len
    codes, quals = synth.sample_reads(g, n_reads, read_len, seed + 1000, err=err, circular=circular)
    return g, synth.to_fastq(codes, quals)


def run_oracle(fq_list, **kw):
    o = Oracle(**kw)
    for fq in fq_list:
        o.add_fastq(fq)
    o.count(naive=False)
    return o


def compare_all(helper, oracle, check_graph=True):
    """Stage-by-stage equality of a finished product run (preprocess+assemble done) with an
    oracle on which count() has been called; runs the oracle's assemble itself."""
    # a) solid set (the full distinct table is only readable before assemble)
    hk, hc, order = sorted_table(*helper.solid())
    ok_, oc_ = oracle.solid()
    assert helper.n_solid == len(oc_)
    assert np.array_equal(hk, ok_), "solid k-mer keys differ"
    assert np.array_equal(hc, oc_), "solid k-mer counts differ"
    # b) histogram, threshold
    assert np.array_equal(helper.histo(), oracle.histo())
    assert helper.used_min_count == oracle.used_min_count
    assert helper.total_instances == oracle.total_instances
    info = json.loads(helper.get_preprocessing_info())
    assert info == json.loads(oracle.preprocessing_json())
    if check_graph:
        a0, a1, alive = helper.adjacency()
        adj_before = oracle.adjacency()
        assert np.array_equal(a0[order], adj_before), "initial adjacency differs"
        oracle.assemble()
        assert np.array_equal(alive[order], oracle.alive()), "post-correction node set differs"
        assert np.array_equal(a1[order], oracle.adjacency()), "post-correction adjacency differs"
    else:
        oracle.assemble()
    out = json.loads(helper.get_assembly())
    ref = json.loads(oracle.assembly_json())
    assert out["ncontigs"] == ref["ncontigs"]
    assert out["outfasta"] == ref["outfasta"], "FASTA differs"
    assert out["outgfa"] == ref["outgfa"], "GFA1 differs"
    assert out["outgfav2"] == ref["outgfav2"], "GFA2 differs"
    assert out["outdot"] == ref["outdot"], "DOT differs"
    assert helper.get_assembly() == oracle.assembly_json()
    return out


def damaged_fastq_texts(seed, n):
    """n small FASTQ texts, most of them damaged in one way (SPEC S1 / S2): clean, CRLF on all or some lines, blank lines at the
    end or in the middle, a missing '@' / '+', qualities of another length, records cut anywhere, lower case, IUPAC codes,
    quality bytes below '!', tabs, bytes >= 0x80, a last line without its newline.  Yields (text, min_qual, what)."""
    rng = np.random.default_rng(seed)
    for case in range(n):
        recs = []
        for i in range(int(rng.integers(0, 6))):
            L_ = int(rng.choice([0, 1, 14, 15, 16, 40, 90]))
            alphabet = "ACGT" if rng.random() < 0.6 else str(rng.choice(["ACGTN", "ACGTacgtn", "ACGTRYKM.-*"]))
            seq = "".join(rng.choice(list(alphabet), L_)) if L_ else ""
            qual = "".join(chr(int(c)) for c in rng.integers(33, 75, L_))
            recs.append([f"@r{i} extra", seq, "+" if rng.random() < 0.7 else f"+r{i} extra", qual])
        what = str(rng.choice(["none", "none", "crlf", "crlf_some", "no_last_newline", "blank_end", "blank_middle", "no_plus", "no_at",
                               "qual_short", "qual_long", "cut", "low_qual_byte", "tabs", "high_bytes", "fasta", "only_newlines"]))
        if recs and what == "no_plus": recs[int(rng.integers(len(recs)))][2] = "ACGT"
        if recs and what == "no_at": r_ = recs[int(rng.integers(len(recs)))]; r_[0] = r_[0][1:]
        if recs and what == "qual_short": r_ = recs[int(rng.integers(len(recs)))]; r_[3] = r_[3][:-1] if r_[3] else "I"
        if recs and what == "qual_long": recs[int(rng.integers(len(recs)))][3] += "I"
        if recs and what == "low_qual_byte": r_ = recs[int(rng.integers(len(recs)))]; r_[3] = (" " + r_[3][1:]) if r_[3] else r_[3]
        if recs and what == "tabs": r_ = recs[int(rng.integers(len(recs)))]; r_[1] = r_[1].replace("A", "\t", 1); 
        if recs and what == "fasta": recs[0][0] = ">" + recs[0][0][1:]
        lines = [x for r_ in recs for x in r_]
        nl = "\r\n" if what == "crlf" else "\n"
        text = "".join(x + (("\r\n" if rng.random() < 0.5 else "\n") if what == "crlf_some" else nl) for x in lines)
        if what == "no_last_newline" and text: text = text.rstrip("\r\n")
        if what == "blank_end": text += "\n" * int(rng.integers(1, 4))
        if what == "blank_middle" and len(recs) >= 2: text = text.replace(nl + "@r1", nl + nl + "@r1", 1)
        if what == "only_newlines": text = "\n" * int(rng.integers(0, 5))
        data = text.encode()
        if what == "cut" and data: data = data[:int(rng.integers(0, len(data)))]
        if what == "high_bytes" and data:
            b_ = bytearray(data); b_[int(rng.integers(len(b_)))] = int(rng.integers(128, 256)); data = bytes(b_)
        yield data, int(rng.choice([0, 10, 20])), what
