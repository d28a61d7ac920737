"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, stage by stage,
bit-exact (integer / byte / index work throughout).  Run on the MI355X box: pytest -m gpu."""
import gzip
import json

import numpy as np
import pytest

import cases
from sparrowhawk_amd import AssemblyHelper, ShkError, synth
from util import (compare_all, int_to_words, make_dataset, parse_fastq, py_count, revcomp,
                  run_oracle, sorted_table)

pytestmark = pytest.mark.gpu


def product(fq1, fq2=None, k=31, min_count=5, min_qual=20, csize=0, do_bloom=False, do_fit=False,
            no_bubble=False, no_deadend=False, assemble=True):
    h = AssemblyHelper.new(k, True, min_count, min_qual, csize, do_bloom, do_fit, no_bubble, no_deadend)
    h.preprocess(fq1, fq2)
    if assemble:
        h.assemble()
    return h


def contig_set(h):
    fa = json.loads(h.get_assembly())["outfasta"].split("\n")
    return {l for l in fa if l and not l.startswith(">")}


@pytest.mark.parametrize("k,err,cov", [(31, 0.0, 30), (31, 0.01, 40), (51, 0.01, 40), (21, 0.005, 25),
                                        (63, 0.01, 40), (33, 0.0, 20),
                                        (65, 0.005, 40), (89, 0.005, 40), (95, 0.0, 30), (97, 0.005, 40),
                                        (127, 0.002, 40)])
def test_full_pipeline_parity(k, err, cov):
    g, fq = make_dataset(30000, cov, err=err, seed=k + int(err * 1000))
    h = product(fq, k=k, min_count=3)
    o = run_oracle([fq], k=k, min_count=3)
    out = compare_all(h, o)
    assert out["ncontigs"] >= 1


@pytest.mark.parametrize("k", [31, 51, 89, 127, 129, 191, 255])
def test_distinct_table_parity(k):
    """Stage (a): the complete (canonical k-mer, count) table before filtering.  (k up to 255 — docs/src/assembly.md:13 —
    in five- to eight-word keys; reads of 400 bases there.)"""
    g, fq = make_dataset(20000, 20, read_len=150 if k <= 127 else 400, err=0.01 if k <= 127 else 0.002, seed=40 + k)
    h = product(fq, k=k, min_count=0, min_qual=20, assemble=False)
    hk, hc, _ = sorted_table(*h.distinct())
    o = run_oracle([fq], k=k, min_count=0, min_qual=20)
    ok_, oc_ = o.distinct()
    assert np.array_equal(hk, ok_) and np.array_equal(hc, oc_)
    assert h.total_instances == o.total_instances
    # and against the independent pure-Python count
    ref = py_count(parse_fastq(fq), k, min_qual=20)
    W = (2 * k + 63) // 64
    got = {tuple(int(x) for x in hk[i]): int(hc[i]) for i in range(0, len(hc), 37)}
    for key, c in got.items():
        v = sum(key[j] << (64 * j) for j in range(W))
        assert ref[v] == c


@pytest.mark.parametrize("k,circular", [(129, False), (161, True), (223, False), (255, True)])
def test_wide_keys_whole_pipeline(k, circular):
    """k = 129 ... 255 (five- to eight-word keys): counting, graph, correction, collapse and the four output formats against
    the oracle, stage by stage; a circular replicon among them."""
    g, fq = make_dataset(30000, 40, read_len=400, err=0.002, seed=300 + k, circular=circular)
    for mc in (0, 3):
        h = product(fq, k=k, min_count=mc, min_qual=0)
        o = run_oracle([fq], k=k, min_count=mc, min_qual=0)
        compare_all(h, o)


def test_error_reads_with_correction_flags():
    g, fq = make_dataset(40000, 50, err=0.01, seed=77)
    for nb, nd in ((False, False), (True, False), (False, True), (True, True)):
        h = product(fq, k=31, min_count=2, no_bubble=nb, no_deadend=nd)
        o = run_oracle([fq], k=31, min_count=2, no_bubble_collapse=nb, no_dead_end_removal=nd)
        compare_all(h, o)


def test_low_filter_many_tips_and_bubbles():
    """min_count=0 keeps every error k-mer: thousands of tips and bubbles, several rounds."""
    g, fq = make_dataset(15000, 30, err=0.01, seed=5)
    h = product(fq, k=31, min_count=0, min_qual=0)
    o = run_oracle([fq], k=31, min_count=0, min_qual=0)
    compare_all(h, o)
    assert o.tips_removed > 0 and o.bubbles_removed > 0


def test_quality_masking_and_fit():
    g, fq = make_dataset(30000, 60, err=0.01, seed=9)
    for mq in (0, 20):
        h = product(fq, k=31, min_count=5, min_qual=mq, do_fit=True)
        o = run_oracle([fq], k=31, min_count=5, min_qual=mq, do_fit=True)
        compare_all(h, o)
        assert h.used_min_count == o.used_min_count


def test_hand_cases():
    c = cases.tip_case()
    assert contig_set(product(c["fastq"], k=c["k"], min_count=0, min_qual=0)) == c["with_removal"]
    assert contig_set(product(c["fastq"], k=c["k"], min_count=0, min_qual=0, no_deadend=True)) == c["without_removal"]
    c = cases.bubble_case()
    assert contig_set(product(c["fastq"], k=c["k"], min_count=0, min_qual=0)) == c["with_collapse"]
    assert contig_set(product(c["fastq"], k=c["k"], min_count=0, min_qual=0, no_bubble=True)) == c["without_collapse"]
    c = cases.cycle_case()
    h = product(c["fastq"], k=c["k"], min_count=0, min_qual=0)
    assert contig_set(h) == c["expect"]
    compare_all(h, run_oracle([c["fastq"]], k=c["k"], min_count=0, min_qual=0))
    c = cases.palindrome_case()
    h = product(c["fastq"], k=c["k"], min_count=0, min_qual=0)
    assert contig_set(h) == c["expect"]
    compare_all(h, run_oracle([c["fastq"]], k=c["k"], min_count=0, min_qual=0))


def test_circular_genome_with_and_without_splitters():
    # long cycle (has sampled splitters) and short cycle (may have none)
    for n, seed in ((5000, 21), (40, 22)):
        g = synth.random_genome(n, seed)
        codes, quals = synth.sample_reads(g, n * 40 // 100 + 50, 100, seed, circular=True)
        fq = synth.to_fastq(codes, quals)
        h = product(fq, k=31, min_count=1)
        o = run_oracle([fq], k=31, min_count=1)
        out = compare_all(h, o)
        assert out["ncontigs"] == 1
        assert "L\t1\t+\t1\t+\t30M" in out["outgfa"]


@pytest.mark.parametrize("k,split_log,tile_rows", [(31, None, None), (31, "0", None), (31, "2", "1"), (31, "9", "50"), (51, None, "7"),
                                                  (51, "12", "1000"), (21, "1", None), (31, "14", "3"), (41, None, "129"), (31, None, "4096"), (51, "3", "4000")])
def test_many_circular_and_linear_replicons(k, split_log, tile_rows, monkeypatch):
    """Circular unitigs of every size next to linear ones, all on the device (SPEC S10): cycles with many sampled
    splitters, with exactly one, with none (the sampling rate is moved around to force each), on both strands; the
    LDS tiles of the fragment pass cut down to a few rows, so that fragments, rings and orphan rings cross many
    tile edges (SHK_TILE_ROWS); or as large as the LDS arrays, so that a tile whose edge moves on to the next partition
    start overflows them and is worked off in chunks."""
    if split_log is not None:
        monkeypatch.setenv("SHK_SPLIT_LOG", split_log)
    if tile_rows is not None:
        monkeypatch.setenv("SHK_TILE_ROWS", tile_rows)
    rng = np.random.default_rng(1000 + k)
    texts = []
    sizes = [k + 9, 64, 100, 333, 1000, 4000, 20000]
    for j, n in enumerate(sizes):                                       # plasmids
        g = synth.random_genome(n, 700 + 10 * k + j)
        codes, quals = synth.sample_reads(g, max(60, n * 30 // 100), 100, 900 + j, circular=True)
        texts.append(synth.to_fastq(codes, quals, prefix=f"c{j}_"))
    for j, n in enumerate((500, 7000)):                                 # linear pieces
        g = synth.random_genome(n, 800 + 10 * k + j)
        codes, quals = synth.sample_reads(g, n * 30 // 100, 100, 950 + j)
        texts.append(synth.to_fastq(codes, quals, prefix=f"l{j}_"))
    fq = b"".join(texts)
    h = product(fq, k=k, min_count=1)
    o = run_oracle([fq], k=k, min_count=1)
    out = compare_all(h, o)
    assert out["ncontigs"] >= len(sizes) + 2
    t = h.timings()
    assert "collapse_host_cycles" not in t                              # no host walk any more
    lens_ = sorted(len(l) for l in out["outfasta"].split("\n") if l and not l.startswith(">"))
    for n in sizes:
        assert n + k - 1 in lens_, (n, lens_)                           # a circular unitig of n nodes spells n + k - 1 bases


def test_repeats_make_a_branching_graph():
    rng = np.random.default_rng(3)
    rep = synth.random_genome(400, 100)
    parts = [synth.random_genome(3000, 101), rep, synth.random_genome(3000, 102), rep,
             synth.random_genome(3000, 103), 3 - rep[::-1], synth.random_genome(2000, 104)]
    g = np.concatenate(parts).astype(np.uint8)
    codes, quals = synth.sample_reads(g, len(g) * 40 // 150, 150, 7)
    fq = synth.to_fastq(codes, quals)
    h = product(fq, k=31, min_count=2)
    o = run_oracle([fq], k=31, min_count=2)
    out = compare_all(h, o)
    assert out["ncontigs"] > 3 and "L\t" in out["outgfa"]


def test_paired_gzip_and_streaming_equal_single_file():
    g, fq = make_dataset(20000, 30, err=0.005, seed=31)
    recs = fq.decode().split("@r")[1:]
    half = len(recs) // 2
    f1 = ("@r" + "@r".join(recs[:half])).encode()
    f2 = ("@r" + "@r".join(recs[half:])).encode()
    a = product(fq, k=31, min_count=2)
    b = product(gzip.compress(f1), gzip.compress(f2), k=31, min_count=2)
    assert a.get_assembly() == b.get_assembly()
    assert a.get_preprocessing_info() == b.get_preprocessing_info()
    c = AssemblyHelper.new(31, True, 2, 20, 150000, False, False, False, False)
    c.push_reads(f1); c.push_reads(f2); c.finish_reads(); c.assemble()
    assert a.get_assembly() == c.get_assembly()
    o = run_oracle([f1, f2], k=31, min_count=2)
    compare_all(b, o)


def test_large_single_member_gzip_goes_through_the_multithreaded_reader(monkeypatch):
    """a single-member .fastq.gz of >= 64 MB of text (the reference's real input: fastx_wasm.rs:53-70,
    docs/src/assembly.md:28) is inflated by csrc/inflate_mt.cpp; the assembly equals the one from the plain text and, as
    file 2 of a pair next to a plain file 1, the pooled result equals the oracle's.  (SHK_GUNZIP_DEVICE=0: the HOST reader —
    since round 4 a member like this goes to the device inflater first: the tests below.)"""
    monkeypatch.setenv("SHK_GUNZIP_DEVICE", "0")
    g = synth.random_genome(300000, 5)
    codes, quals = synth.sample_reads(g, 230000, 150, 6, err=0.01)
    fq = bytes(synth.to_fastq_fixed(codes, quals))
    assert len(fq) >= 64_000_000
    z = gzip.compress(fq, compresslevel=6)
    a = product(fq, k=31, min_count=3)
    b = product(z, k=31, min_count=3)
    assert a.get_assembly() == b.get_assembly() and a.get_preprocessing_info() == b.get_preprocessing_info()
    assert b.timings().get("gunzip_mt_members_x1", 0) == 1, b.timings()
    g2, small = make_dataset(20000, 30, err=0.005, seed=35)
    c = product(small, z, k=31, min_count=3)
    assert c.timings().get("gunzip_mt_members_x1", 0) == 1
    o = run_oracle([small, fq], k=31, min_count=3)
    compare_all(c, o, check_graph=False)


def _device_gunzip(lib, z):
    import ctypes as C
    out, n, why = C.c_void_p(), C.c_size_t(), C.c_char_p()
    rc = lib.shk_device_gunzip(z, len(z), C.byref(out), C.byref(n), C.byref(why), None)
    if rc == 0:
        got = C.string_at(out.value, n.value)
        lib.shk_host_free(out)
        return 0, got, ""
    return rc, None, (why.value or b"").decode()


def test_device_inflater_gives_zlibs_bytes_or_declines(lib, monkeypatch):
    """csrc/inflate_gpu.hip alone (shk_device_gunzip): members of FASTQ text at several levels, windows and chunk sizes come
    back byte for byte as zlib gives them; several members, BGZF, binary data, truncated and bit-flipped streams are DECLINED
    (the product then reads them on the host) or — harmless damage — still equal zlib's bytes.  Never other bytes."""
    import zlib
    monkeypatch.setenv("SHK_GUNZIP_DEVICE_MIN", "32768")
    rng = np.random.default_rng(77)
    g = synth.random_genome(100000, 9)
    taken = 0
    for level, wbits, n_reads, rl, chunk in ((1, 31, 60000, 150, 0), (6, 31, 60000, 150, 16384), (9, 31, 20000, 251, 4096), (6, 28, 30000, 100, 0),
                                             (1, 31, 5000, 75, 0), (4, 31, 120000, 150, 65536)):
        codes, quals = synth.sample_reads(g, n_reads, rl, int(rng.integers(1 << 30)), err=0.01)
        fq = bytes(synth.to_fastq_fixed(codes, quals))
        co = zlib.compressobj(level, zlib.DEFLATED, wbits)
        z = co.compress(fq) + co.flush()
        if chunk:
            monkeypatch.setenv("SHK_GUNZIP_DEVICE_CHUNK", str(chunk))
        else:
            monkeypatch.delenv("SHK_GUNZIP_DEVICE_CHUNK", raising=False)
        rc, got, why = _device_gunzip(lib, z)
        assert rc in (0, 1), (rc, why)
        if rc == 0:
            assert got == fq, (level, wbits, len(got), len(fq))
            taken += 1
        # two members behind one another: not this reader's case
        rc2, got2, why2 = _device_gunzip(lib, z + z)
        assert rc2 == 1, why2
        # damaged: declined, or zlib's bytes
        for _ in range(4):
            pos = int(rng.integers(12, len(z)))
            bad = bytearray(z); bad[pos] ^= 1 << int(rng.integers(0, 8)); bad = bytes(bad)
            try:
                ref = zlib.decompress(bad, 31)
            except zlib.error:
                ref = None
            rc3, got3, _ = _device_gunzip(lib, bad)
            assert rc3 == 1 or (rc3 == 0 and ref is not None and got3 == ref), (pos, rc3)
        rc4, _, _ = _device_gunzip(lib, z[:len(z) // 2])
        assert rc4 == 1
    assert taken >= 4, taken
    # binary data, a tiny member: declined
    assert _device_gunzip(lib, gzip.compress(rng.integers(0, 256, 3_000_000, dtype=np.uint8).tobytes()))[0] == 1
    assert _device_gunzip(lib, gzip.compress(b"@r\nACGT\n+\nIIII\n"))[0] == 1


def test_fastq_gz_is_inflated_on_the_device(monkeypatch):
    """The reference's real input (a .fastq.gz, or a pair of them: fastx_wasm.rs:53-70, docs/src/assembly.md:25-28) through
    shk_preprocess: a plain gzip member is inflated on the device (csrc/inflate_gpu.hip — the compressed bytes are what
    crosses PCIe) and its text goes straight to the device parser; results equal those of the plain text and the oracle's,
    one file and two, with masked bases and with the host reader taking over what the device inflater declines."""
    monkeypatch.setenv("SHK_GUNZIP_DEVICE_MIN", "65536")
    g = synth.random_genome(200000, 15)
    codes, quals = synth.sample_reads(g, 60000, 150, 16, err=0.01)
    fq = bytes(synth.to_fastq_fixed(codes, quals))
    z = gzip.compress(fq, compresslevel=6)
    a = product(fq, k=31, min_count=3)
    b = product(z, k=31, min_count=3)
    assert b.timings().get("gunzip_device_members_x1", 0) == 1, b.timings()
    assert a.get_assembly() == b.get_assembly() and a.get_preprocessing_info() == b.get_preprocessing_info()
    o = run_oracle([fq], k=31, min_count=3)
    compare_all(b, o, check_graph=False)
    # a pair of .fastq.gz: both on the device, pooled
    recs = fq.decode().split("@r")[1:]
    half = len(recs) // 2
    f1 = ("@r" + "@r".join(recs[:half])).encode(); f2 = ("@r" + "@r".join(recs[half:])).encode()
    c = product(gzip.compress(f1, compresslevel=1), gzip.compress(f2, compresslevel=9), k=31, min_count=3)
    assert c.timings().get("gunzip_device_members_x1", 0) == 2
    assert c.get_assembly() == a.get_assembly() and c.get_preprocessing_info() == a.get_preprocessing_info()
    # file 2 is two members: the device inflater declines, the host reader takes the pair
    two = gzip.compress(f2[:f2.rfind(b"@r", 0, len(f2) // 2)]) + gzip.compress(f2[f2.rfind(b"@r", 0, len(f2) // 2):])
    e = product(gzip.compress(f1), two, k=31, min_count=3)
    assert e.timings().get("gunzip_device_members_x1", 0) == 0 and e.timings().get("gunzip_device_not_taken_x1", 0) == 1
    assert e.get_assembly() == a.get_assembly() and e.get_preprocessing_info() == a.get_preprocessing_info()
    # a damaged stream: the error is the host reader's
    bad = bytearray(z); bad[len(bad) // 2] ^= 0x10
    with pytest.raises(ShkError) as ei:
        product(bytes(bad), k=31, min_count=3)
    assert ei.value.code == -3


def test_metamorphic_read_order_and_strand():
    g, fq = make_dataset(20000, 30, err=0.005, seed=32)
    lines = fq.decode().strip().split("\n")
    reads = [(lines[i + 1], lines[i + 3]) for i in range(0, len(lines), 4)]
    rng = np.random.default_rng(1)
    out = []
    for j, i in enumerate(rng.permutation(len(reads))):
        s, q = reads[i]
        if j % 2:
            s, q = revcomp(s), q[::-1]
        out.append(f"@x{j}\n{s}\n+\n{q}\n")
    a = product(fq, k=31, min_count=2)
    b = product("".join(out).encode(), k=31, min_count=2)
    assert a.get_assembly() == b.get_assembly()
    # determinism: the same input twice gives identical bytes
    c = product(fq, k=31, min_count=2)
    assert a.get_assembly() == c.get_assembly()


def test_progress_states_and_modes():
    g, fq = make_dataset(5000, 20, seed=33)
    for kw, mode in ((dict(csize=0), "bulk"), (dict(csize=100), "chunked"),
                     (dict(do_bloom=True), "bloom")):
        h = product(fq, k=31, min_count=5, do_fit=True, **kw)
        s = h.states
        assert s[0] == "preprocess:start" and s[1] == f"preprocess:{mode}:start"
        # loop:start / loop:end exist for bulk and bloom only (AssemblyPage.vue:467,492,508,533); the chunked
        # branch of the reference UI (:548-579) knows :start, :fitting, :filtering and :loop:<n>[:<pct>] only
        assert (s[2] == f"preprocess:{mode}:loop:start") == (mode != "chunked")
        assert (f"preprocess:{mode}:loop:end" in s) == (mode != "chunked")
        ui_chunked = {"preprocess:chunked:start", "preprocess:chunked:fitting", "preprocess:chunked:filtering"}
        for st in s:
            if st.startswith("preprocess:chunked:") and st not in ui_chunked:
                parts = st.split(":")
                assert parts[2] == "loop" and parts[3].isdigit() and (len(parts) == 4 or parts[4].isdigit()), st
        if mode == "chunked":                          # (666 reads: only the chunked run, 100 reads per chunk, posts progress)
            assert sum(st.startswith("preprocess:chunked:loop:") for st in s) >= 6
        assert (f"preprocess:bulk:sorting" in s) == (mode == "bulk")
        assert f"preprocess:{mode}:fitting" in s and f"preprocess:{mode}:filtering" in s
        i = s.index("preprocess:saving")
        assert s[i:] == ["preprocess:saving", "preprocess:end", "assembly:start", "assembly:create_graph",
                         "assembly:correct_graph", "assembly:collapse_graph", "assembly:saving", "assembly:end"]
    # all modes count exactly the same here (SPEC S4)
    outs = {product(fq, k=31, min_count=5, **kw).get_assembly()
            for kw in (dict(csize=0), dict(csize=1000), dict(do_bloom=True))}
    assert len(outs) == 1


def test_state_machine_and_errors():
    g, fq = make_dataset(3000, 10, seed=34)
    h = AssemblyHelper.new(31, True, 2, 20, 0, False, False, False, False)
    with pytest.raises(ShkError) as e:
        h.assemble()
    assert e.value.code == -2
    with pytest.raises(ShkError):
        h.get_preprocessing_info()
    with pytest.raises(ShkError) as e:
        h.preprocess(b"@r\nACGT\n+\nII\n")
    assert e.value.code == -3
    h = AssemblyHelper.new(31, True, 2, 20, 0, False, False, False, False)
    h.preprocess(fq)
    with pytest.raises(ShkError) as e:
        h.preprocess(fq)                                  # one preprocess per handle (Assembler.ts:92)
    assert e.value.code == -2
    with pytest.raises(ShkError):
        h.get_assembly()
    h.assemble()
    with pytest.raises(ShkError):
        h.assemble()
    json.loads(h.get_assembly())


def test_empty_and_degenerate_inputs():
    for fq in (b"", b"@r\nACGTACGT\n+\nIIIIIIII\n", b"@r\n" + b"N" * 100 + b"\n+\n" + b"I" * 100 + b"\n"):
        h = product(fq, k=31, min_count=0)
        out = json.loads(h.get_assembly())
        assert out["ncontigs"] == 0 and out["outfasta"] == ""
        info = json.loads(h.get_preprocessing_info())
        assert info["nkmers"] == 0 and sum(info["histo"]) == 0
        compare_all(h, run_oracle([fq], k=31, min_count=0), check_graph=False)
    # a single k-mer
    fq = b"@r\n" + b"ACGTTGCATGCCGATAGCTAGCTAGGATCCA" + b"\n+\n" + b"I" * 31 + b"\n"
    h = product(fq, k=31, min_count=0)
    compare_all(h, run_oracle([fq], k=31, min_count=0))
    # homopolymer: one node with a self-loop
    fq = b"@r\n" + b"A" * 60 + b"\n+\n" + b"I" * 60 + b"\n"
    h = product(fq, k=31, min_count=0)
    compare_all(h, run_oracle([fq], k=31, min_count=0))
    # ragged read lengths
    rng = np.random.default_rng(5)
    g = synth.random_genome(5000, 55)
    recs = []
    for i in range(600):
        L = int(rng.integers(20, 260)); s = int(rng.integers(0, 5000 - L))
        seq = synth.codes_to_str(g[s:s + L])
        recs.append(f"@q{i}\n{seq}\n+\n{'I' * L}\n")
    fq = "".join(recs).encode()
    compare_all(product(fq, k=31, min_count=1), run_oracle([fq], k=31, min_count=1))


def test_count_saturating_histogram_bin():
    """A k-mer seen more than 500 times lands in the last bin (SPEC S5)."""
    s = synth.codes_to_str(synth.random_genome(60, 66))
    fq = (f"@r\n{s}\n+\n{'I' * 60}\n" * 700).encode()
    h = product(fq, k=31, min_count=5)
    o = run_oracle([fq], k=31, min_count=5)
    compare_all(h, o)
    assert h.histo()[499] == 30


def test_device_packed_entry_matches_host_entry():
    import torch
    from sparrowhawk_amd import pack_fastq
    g, fq = make_dataset(20000, 30, err=0.005, seed=35)
    a = product(fq, k=31, min_count=2)
    bases, seg, nb, nr = pack_fastq(fq, 31, 20)
    d_bases = torch.from_numpy(bases.view(np.int32)).cuda()
    d_seg = torch.from_numpy(seg.view(np.int32)).cuda()
    torch.cuda.synchronize()
    h = AssemblyHelper.new(31, True, 2, 20, 0, False, False, False, False)
    h.preprocess_packed_device(d_bases.data_ptr(), d_seg.data_ptr(), len(seg) - 1, nb, nr)
    h.assemble()
    assert a.get_assembly() == h.get_assembly()
    assert a.get_preprocessing_info() == h.get_preprocessing_info()


def _with_env(env, fn):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_counting_modes_agree():
    """The partitioned LDS counting path — pass 2 as two kernels (k_dedupe_partitions + k_count_weighted: the default for one-
    and two-word keys), with one / two / four partitions per table, and as the fused k_count_partitions — and the
    single-HBM-table path give identical tables."""
    g, fq = make_dataset(30000, 40, err=0.01, seed=91)
    for k in (31, 51):
        a = product(fq, k=k, min_count=0, assemble=False)
        assert "count_dedupe_kernel" in a.timings()
        ak, ac, _ = sorted_table(*a.distinct())
        for env in ({"SHK_COUNT_MODE_GLOBAL": 1}, {"SHK_COUNT_SPLIT": 0}, {"SHK_COUNT_MERGE": 1}, {"SHK_COUNT_MERGE": 4},
                    {"SHK_PART_P": 2048, "SHK_PROBE_PARTS": 128}, {"SHK_PART_P": 2048, "SHK_PROBE_PARTS": 128, "SHK_COUNT_SPLIT": 0}):
            b = _with_env(env, lambda: product(fq, k=k, min_count=0, assemble=False))
            assert ("count_dedupe_kernel" in b.timings()) == ("SHK_COUNT_SPLIT" not in env and "SHK_COUNT_MODE_GLOBAL" not in env), env
            bk, bc, _ = sorted_table(*b.distinct())
            assert np.array_equal(ak, bk) and np.array_equal(ac, bc), env
            assert np.array_equal(a.histo(), b.histo()) and a.total_instances == b.total_instances, env


@pytest.mark.parametrize("k,mode", [(31, "repartition"), (51, "repartition"), (89, "repartition"), (127, "repartition"),
                                    (31, "residue"), (51, "residue"), (31, "tiny_buckets"), (51, "rescatter"),
                                    (31, "probe"), (51, "probe")])
def test_partitions_that_exceed_the_lds_table(k, mode):
    """Few partitions + many distinct k-mers: each partition exceeds the LDS table.  It is then
    repartitioned at k-mer level (k_ovf_scatter / k_count_buckets); with that switched off, or when a
    bucket region overflows, it is re-run per residue class of the key hash (count_part.h).  Results
    must not change.  "probe": more partitions than workgroups in flight; the first ones find the
    input error-rich and the later ones hand their partitions over without trying them."""
    if mode == "probe":
        g, fq = make_dataset(400000, 45, err=0.02, seed=94)
        env = {"SHK_PART_P": 1024, "SHK_PROBE_PARTS": 128}
        n_tables = 1024
    else:
        g, fq = make_dataset(150000, 12, err=0.02, seed=92)
        env = {"SHK_PART_P": 64}
        n_tables = 64
    if mode == "residue":
        env["SHK_NO_REPARTITION"] = 1
    if mode == "tiny_buckets":
        env["SHK_OVF_CAP_PCT"] = 60                      # bucket regions overflow -> residue-class fallback per partition
        env["SHK_OVF_MAX_PASSES"] = 1
    if mode == "rescatter":
        env["SHK_OVF_CAP_PCT"] = 60                      # bucket regions overflow -> scattered again with the exact room
    def both():
        hh = product(fq, k=k, min_count=0, min_qual=0, assemble=False)
        tt = hh.timings()                                # of the preprocess run (distinct() counts again)
        return hh, tt, sorted_table(*hh.distinct())
    h, t, (hk, hc, _) = _with_env(env, both)
    o = run_oracle([fq], k=k, min_count=0, min_qual=0)
    ok_, oc_ = o.distinct()
    assert len(oc_) > n_tables * (6144 if k <= 63 else 3648)  # really more than the LDS tables hold
    assert np.array_equal(hk, ok_) and np.array_equal(hc, oc_)
    assert np.array_equal(h.histo(), o.histo()) and h.total_instances == o.total_instances
    if mode == "probe":
        assert t.get("count_deferred_untried_x1", 0) >= 256 and t.get("count_repartitioned_x1", 0) >= 900, t
    if mode == "repartition":
        assert t.get("count_repartitioned_x1", 0) > 0
    if mode == "residue":
        assert "count_repartitioned_x1" not in t
    if mode == "tiny_buckets":
        assert t.get("count_residue_rerun_x1", 0) > 0
    if mode == "rescatter":
        assert t.get("count_residue_rerun_x1", 0) == 0 and t.get("count_repartitioned_x1", 0) > 0


def test_partition_slice_overflow_retry():
    """Reads exactly k long give one record per k-mer: the first sizing guess overflows and the
    pass is rerun with exact slice sizes."""
    rng = np.random.default_rng(93)
    g = synth.random_genome(4000, 93)
    n = 300000
    starts = rng.integers(0, 4000 - 31, n)
    seqs = ["".join("ACGT"[c] for c in g[s:s + 31]) for s in starts[:2000]]
    recs = [f"@r{i}\n{seqs[i % 2000]}\n+\n{'I' * 31}\n" for i in range(n)]
    fq = "".join(recs).encode()
    h = _with_env({"SHK_PART_P": 64}, lambda: product(fq, k=31, min_count=1))
    assert "partition_retry" in h.timings()
    compare_all(h, run_oracle([fq], k=31, min_count=1))


def _both_parsers(f1, f2=None, k=21, min_count=0, min_qual=20, csize=0):
    """The same input through the device FASTQ parser (default) and the host parser."""
    a = product(f1, f2, k=k, min_count=min_count, min_qual=min_qual, csize=csize, assemble=False)
    b = _with_env({"SHK_HOST_PARSER": 1},
                  lambda: product(f1, f2, k=k, min_count=min_count, min_qual=min_qual, csize=csize, assemble=False))
    ak, ac, _ = sorted_table(*a.distinct())
    bk, bc, _ = sorted_table(*b.distinct())
    assert np.array_equal(ak, bk) and np.array_equal(ac, bc)
    assert a.total_instances == b.total_instances
    assert a.states == b.states                         # progress strings, read counts and percentages included
    assert a.get_preprocessing_info() == b.get_preprocessing_info()
    return a, b


def test_device_fastq_parser_equals_host_parser():
    rng = np.random.default_rng(97)
    g, fq = make_dataset(8000, 25, err=0.02, seed=97)
    recs = fq.decode().split("\n")
    # mixed case, N's, low qualities, CRLF on some lines
    lines = []
    for i in range(0, len(recs) - 1, 4):
        name, seq, plus, qual = recs[i:i + 4]
        seq = list(seq)
        for j in rng.integers(0, len(seq), size=3):
            seq[j] = "N" if rng.random() < 0.5 else seq[j].lower()
        eol = "\r\n" if (i // 4) % 3 == 0 else "\n"
        lines += [name + eol, "".join(seq) + eol, plus + "extra" + eol, qual + eol]
    text = "".join(lines).encode()
    half = text.index(b"@r" + str(len(lines) // 8).encode() + b"\r") if (b"@r" + str(len(lines) // 8).encode() + b"\r") in text else len(text) // 2
    _both_parsers(text, csize=300)                                     # progress every 300 reads
    _both_parsers(text[:-1] if text.endswith(b"\n") else text)         # last line not terminated
    _both_parsers(text + b"\n\r\n\n")                                  # trailing blank lines
    # two files; the first one unterminated; gzip (two members) for the second
    cut = text.rfind(b"\n@r", 0, len(text) // 2) + 1
    f1, f2 = text[:cut], text[cut:]
    a, _ = _both_parsers(f1.rstrip(b"\r\n"), gzip.compress(f2[:len(f2) // 2]) + gzip.compress(f2[len(f2) // 2:]), csize=200)
    b, _ = _both_parsers(text)
    ak, ac, _ = sorted_table(*a.distinct())
    bk, bc, _ = sorted_table(*b.distinct())
    assert np.array_equal(ak, bk) and np.array_equal(ac, bc)
    # against the oracle as well
    o = run_oracle([text], k=21, min_count=0, min_qual=20)
    ok_, oc_ = o.distinct()
    assert np.array_equal(bk, ok_) and np.array_equal(bc, oc_)
    # irregular framing (a blank line between records) is parsed by the host parser: same result
    irregular = f1 + b"\n" + f2
    c = product(irregular, k=21, min_count=0, assemble=False)
    ck, cc, _ = sorted_table(*c.distinct())
    assert np.array_equal(ck, bk) and np.array_equal(cc, bc)
    # malformed records keep their error code and message whichever parser looks first
    bad = text.replace(b"+extra", b"-extra", 1)
    for data in (bad, text[: len(text) // 2 + 7], b"@x\nACGT\n+\nIII\n"):
        h = AssemblyHelper.new(21, True, 0, 20, 0, False, False, False, False)
        with pytest.raises(ShkError) as ei:
            h.preprocess(data)
        assert ei.value.code == -3
    # empty inputs
    for data in (b"", b"\n", b"\r\n\n"):
        h = product(data, k=21, min_count=0, assemble=False)
        assert h.n_distinct == 0


def test_damaged_fastq_texts_are_taken_or_refused_as_the_oracle_does():
    """400 small FASTQ texts, most of them damaged (tests/util.py: damaged_fastq_texts — CRLF on some lines, blank lines,
    a missing '@' / '+', qualities of another length, records cut anywhere, IUPAC codes, bytes >= 0x80, ...), through
    shk_preprocess with the device parser first (default) and with the host parser alone: both must refuse exactly the
    texts the oracle refuses (SHK_E_PARSE) and count the oracle's k-mers from the others; a third of them also as the
    second file of a pair behind a clean first file, either file gzip-wrapped or plain."""
    from util import damaged_fastq_texts
    n_ok = n_bad = n_pairs = 0
    clean = None                                            # the last text that was taken and held k-mers
    for case, (data, min_qual, what) in enumerate(damaged_fastq_texts(78, 400)):
        try:
            o = run_oracle([data], k=15, min_count=0, min_qual=min_qual)
            want = o.distinct()
        except ValueError:
            want = None
        for host_parser in (False, True):
            def run():
                h = AssemblyHelper.new(15, False, 0, min_qual, 0, False, False, False, False)
                try:
                    h.preprocess(data)
                    if h.n_distinct == 0:
                        return (np.zeros((0, 1), dtype=np.uint64), np.zeros(0, dtype=np.uint32))
                    kk, cc, _ = sorted_table(*h.distinct())
                    return kk, cc
                except ShkError as e:
                    assert e.code == -3, (case, what, data, e.code)
                    return None
                finally:
                    h.free()
            got = _with_env({"SHK_HOST_PARSER": "1"} if host_parser else {}, run)
            who = "host parser" if host_parser else "device parser first"
            assert (got is None) == (want is None), (case, what, data, who, "product " + ("refuses" if got is None else "takes"))
            if got is not None:
                assert len(got[1]) == len(want[1]) and np.array_equal(got[0].reshape(-1), want[0].reshape(-1)) and np.array_equal(got[1], want[1]), (case, what, data, who)
        n_ok += want is not None
        n_bad += want is None
        # as the SECOND file of a pair behind a clean first one (pooled: S1), and gzip-wrapped: the same verdict
        if case % 3 == 0 and clean is not None:
            try:
                want2 = run_oracle([clean[0], data], k=15, min_count=0, min_qual=clean[1]).distinct()
            except ValueError:
                want2 = None
            h = AssemblyHelper.new(15, False, 0, clean[1], 0, False, False, False, False)
            try:
                h.preprocess(gzip.compress(clean[0]) if case % 2 else clean[0], gzip.compress(data) if case % 4 < 2 else data)
                got2 = sorted_table(*h.distinct())[:2] if h.n_distinct else (np.zeros((0, 1), dtype=np.uint64), np.zeros(0, dtype=np.uint32))
            except ShkError as e:
                assert e.code == -3, (case, what, data, e.code)
                got2 = None
            finally:
                h.free()
            assert (got2 is None) == (want2 is None), (case, what, data, "as second file", "product " + ("refuses" if got2 is None else "takes"))
            if got2 is not None:
                assert np.array_equal(got2[0].reshape(-1), want2[0].reshape(-1)) and np.array_equal(got2[1], want2[1]), (case, what, data, "as second file")
            n_pairs += 1
        if want is not None and len(want[1]) > 0:
            clean = (data, min_qual)
    assert n_ok > 100 and n_bad > 60 and n_pairs > 80, (n_ok, n_bad, n_pairs)


def test_several_batches_per_handle():
    """Inputs are handed to the device in batches (chunked mode: every chunk_size reads; any mode: when a
    batch would exceed its 32-bit base offsets; streaming: per pushed chunk).  Every batch keeps its own
    record buffer and pass 2 reads one run per batch and partition: results must not change."""
    g, fq = make_dataset(60000, 30, err=0.01, seed=77)
    ref = product(fq, k=31, min_count=2)
    o = run_oracle([fq], k=31, min_count=2)
    compare_all(ref, o)
    # host parser, small batches by base count (7 or so batches), bulk mode
    a = _with_env({"SHK_HOST_PARSER": 1, "SHK_BATCH_BASES": 300000}, lambda: product(fq, k=31, min_count=2))
    assert a.timings().get("batch_pack_kernel", 0) > 0
    assert a.get_assembly() == ref.get_assembly() and a.get_preprocessing_info() == ref.get_preprocessing_info()
    assert a.total_instances == ref.total_instances
    # chunked mode through the host parser: one batch per 1000 reads
    b = _with_env({"SHK_HOST_PARSER": 1}, lambda: product(fq, k=31, min_count=2, csize=1000))
    assert b.timings().get("batch_pack_kernel", 0) > 0
    assert b.get_assembly() == ref.get_assembly()
    # any number of batches: 60 batches of 200 reads, merged into one whenever 16 have piled up
    m = _with_env({"SHK_HOST_PARSER": 1, "SHK_MERGE_BATCHES_AT": 16}, lambda: product(fq, k=31, min_count=2, csize=200))
    assert m.timings().get("batch_merge_kernel", 0) > 0
    assert m.get_assembly() == ref.get_assembly() and m.get_preprocessing_info() == ref.get_preprocessing_info()
    assert m.total_instances == ref.total_instances
    # streaming entry point with tiny batches, two-word keys, the distinct table itself
    recs = fq.decode().split("@r")[1:]
    parts = [("@r" + "@r".join(recs[i::3])).encode() for i in range(3)]
    def stream():
        h = AssemblyHelper.new(51, True, 0, 20, 0, False, False, False, False)
        for p in parts:
            h.push_reads(p)
        h.finish_reads()
        return h, sorted_table(*h.distinct())
    c, (ck, cc, _) = _with_env({"SHK_BATCH_BASES": 200000}, stream)
    assert c.timings().get("batch_pack_kernel", 0) > 0
    # the same chunks through the device parser (large chunks take it by default), one batch per chunk
    d, (dk, dc, _) = _with_env({"SHK_STREAM_DEVICE_MIN": 1000}, stream)
    assert d.timings().get("fastq_device_chunks_x1", 0) == 3 and "fastq_device_chunks_x1" not in c.timings()
    assert np.array_equal(dk, ck) and np.array_equal(dc, cc) and d.states == c.states
    assert d.total_instances == c.total_instances
    o2 = run_oracle(parts, k=51, min_count=0)
    ok_, oc_ = o2.distinct()
    assert np.array_equal(ck, ok_) and np.array_equal(cc, oc_)
    assert c.total_instances == o2.total_instances


def test_device_parser_takes_large_texts_in_pieces():
    """A text of more than one batch is cut at record boundaries and every piece goes through the device
    parser as its own batch.  Counts, progress strings (read numbers at global multiples, percentages of the
    whole input), results and error messages equal the host parser's."""
    g, fq = make_dataset(60000, 30, err=0.01, seed=78)
    # qualities starting with '@' now and then: a quality line must not be taken for a header at a cut
    recs = fq.decode().split("\n")
    for i in range(3, len(recs), 4 * 7):
        recs[i] = "@" + recs[i][1:]
    fq = "\n".join(recs).encode()
    ref = product(fq, k=31, min_count=2, min_qual=0)
    compare_all(ref, run_oracle([fq], k=31, min_count=2, min_qual=0))
    env = {"SHK_BATCH_BASES": 150000}                    # ~300 kB of text per piece: a dozen pieces
    a = _with_env(env, lambda: product(fq, k=31, min_count=2, min_qual=0, csize=500))
    ta = a.timings()
    assert ta.get("fastq_device_pieces_x1", 0) >= 8 and ta.get("batch_pack_kernel", 0) > 0
    b = _with_env({**env, "SHK_HOST_PARSER": 1}, lambda: product(fq, k=31, min_count=2, min_qual=0, csize=500))
    assert "fastq_device_pieces_x1" not in b.timings()
    assert a.states == b.states
    assert a.get_assembly() == b.get_assembly() == _with_env({}, lambda: product(fq, k=31, min_count=2, min_qual=0, csize=500)).get_assembly()
    assert a.get_preprocessing_info() == b.get_preprocessing_info() and a.total_instances == b.total_instances
    # two files, the second one gzip; k = 51
    cut = fq.rfind(b"\n@r", 0, len(fq) // 3) + 1
    f1, f2 = fq[:cut], gzip.compress(fq[cut:])
    c = _with_env(env, lambda: product(f1, f2, k=51, min_count=1, csize=700))
    d = _with_env({**env, "SHK_HOST_PARSER": 1}, lambda: product(f1, f2, k=51, min_count=1, csize=700))
    assert c.timings().get("fastq_device_pieces_x1", 0) >= 8
    assert c.states == d.states and c.get_assembly() == d.get_assembly()
    compare_all(c, run_oracle([f1, f2], k=51, min_count=1))
    # a malformed record deep inside: same message from both; blank lines in a later piece: host takes over there
    lines = fq.split(b"\n")
    bad = list(lines)
    bad[4 * 5000 + 2] = b"-"                             # the '+' line of record 5000
    bad = b"\n".join(bad)
    msgs = []
    for e in (env, {**env, "SHK_HOST_PARSER": 1}):
        with pytest.raises(Exception) as ei:
            _with_env(e, lambda: product(bad, k=31, min_count=2))
        msgs.append(str(ei.value))
    assert msgs[0] == msgs[1] and "5000" in msgs[0]
    gap = list(lines)
    gap.insert(4 * 9000, b"")                            # a blank line between two records
    gap = b"\n".join(gap)
    e1 = _with_env(env, lambda: product(gap, k=31, min_count=2, min_qual=0, csize=500))
    e2 = _with_env({**env, "SHK_HOST_PARSER": 1}, lambda: product(gap, k=31, min_count=2, min_qual=0, csize=500))
    assert e1.timings().get("fastq_device_pieces_x1", 0) >= 1 and e1.timings().get("fastq_parse_pack_host_clock", 0) == 0
    assert e1.states == e2.states and e1.get_assembly() == e2.get_assembly() == ref.get_assembly()


def test_single_batch_text_parsed_in_pieces_under_the_upload():
    """A FASTQ text of one batch (from 64 MB; forced here on a small one) is cut at record boundaries, piece i + 1 is
    uploaded while piece i is parsed, and the packed pieces are counted together as ONE batch (pass 1 over the
    pieces into the same slices: no batch packing).  Everything observable equals the host parser's."""
    g, fq = make_dataset(60000, 30, err=0.01, seed=79)
    recs = fq.decode().split("\n")
    for i in range(3, len(recs), 4 * 5):
        recs[i] = "@" + recs[i][1:]                      # quality lines that start like headers
    fq = "\n".join(recs).encode()
    env = {"SHK_FASTQ_PIPELINE_MIN": 1, "SHK_FASTQ_PIECES": 5}
    ref = product(fq, k=31, min_count=2, min_qual=0, csize=500)          # (the single-shot device path)
    assert "fastq_device_pieces_x1" not in ref.timings()
    a = _with_env(env, lambda: product(fq, k=31, min_count=2, min_qual=0, csize=500))
    ta = a.timings()
    assert ta.get("fastq_device_pieces_x1", 0) >= 4 and "batch_pack_kernel" not in ta and "batch_merge_kernel" not in ta
    b = _with_env({**env, "SHK_HOST_PARSER": 1}, lambda: product(fq, k=31, min_count=2, min_qual=0, csize=500))
    assert a.states == b.states == ref.states
    assert a.get_assembly() == b.get_assembly() == ref.get_assembly()
    assert a.get_preprocessing_info() == b.get_preprocessing_info() and a.total_instances == b.total_instances
    compare_all(a, run_oracle([fq], k=31, min_count=2, min_qual=0))
    # two files, the second one gzip, the first one without its last newline; two-word keys; one piece per file and more
    cut = fq.rfind(b"\n@r", 0, len(fq) // 3) + 1
    f1, f2 = fq[:cut].rstrip(b"\n"), gzip.compress(fq[cut:])
    for pieces in (2, 7):
        e = {**env, "SHK_FASTQ_PIECES": pieces}
        c = _with_env(e, lambda: product(f1, f2, k=51, min_count=1, csize=700))
        d = _with_env({**e, "SHK_HOST_PARSER": 1}, lambda: product(f1, f2, k=51, min_count=1, csize=700))
        assert c.timings().get("fastq_device_pieces_x1", 0) >= 2
        assert c.states == d.states and c.get_assembly() == d.get_assembly()
    compare_all(c, run_oracle([f1, f2], k=51, min_count=1))
    # a malformed record deep inside: same message; a blank line in a later piece: the pieces parsed so far are
    # counted as a batch, the host parser takes the rest of the file, the result does not change
    lines = fq.split(b"\n")
    bad = list(lines)
    bad[4 * 5000 + 2] = b"-"
    bad = b"\n".join(bad)
    msgs = []
    for e in (env, {**env, "SHK_HOST_PARSER": 1}):
        with pytest.raises(Exception) as ei:
            _with_env(e, lambda: product(bad, k=31, min_count=2))
        msgs.append(str(ei.value))
    assert msgs[0] == msgs[1] and "5000" in msgs[0]
    gap = list(lines)
    gap.insert(4 * 9000, b"")
    gap = b"\n".join(gap)
    e1 = _with_env(env, lambda: product(gap, k=31, min_count=2, min_qual=0, csize=500))
    e2 = _with_env({**env, "SHK_HOST_PARSER": 1}, lambda: product(gap, k=31, min_count=2, min_qual=0, csize=500))
    assert e1.timings().get("fastq_device_pieces_x1", 0) >= 1 and e1.timings().get("fastq_parse_pack_host_clock", 0) == 0
    assert e1.states == e2.states and e1.get_assembly() == e2.get_assembly() == ref.get_assembly()
    # irregular from the first record on: the host parser does everything
    first = b"\n" + fq
    e3 = _with_env(env, lambda: product(first, k=31, min_count=2, min_qual=0))
    assert "fastq_device_pieces_x1" not in e3.timings() and e3.get_assembly() == product(fq, k=31, min_count=2, min_qual=0).get_assembly()


def test_threaded_output_writer_equals_the_serial_one():
    """Megabyte outputs are written by three threads into one buffer whose section offsets are measured
    first; forced here on a small fragmented assembly (many contigs, links) and compared byte for byte."""
    g, fq = make_dataset(40000, 12, err=0.01, seed=201)
    a = product(fq, k=31, min_count=1)
    b = _with_env({"SHK_WRITER_PAR_MIN": 1}, lambda: product(fq, k=31, min_count=1))
    assert json.loads(a.get_assembly())["ncontigs"] > 3
    assert a.get_assembly() == b.get_assembly()
    compare_all(b, run_oracle([fq], k=31, min_count=1))


def test_long_reads_are_split_into_overlapping_segments():
    """A 60 kbp read (longer than a kernel segment) goes through the host packer, which cuts it into
    pieces overlapping by k-1 bases: every k-mer is counted exactly once."""
    g = synth.random_genome(60000, 123)
    seq = synth.codes_to_str(g)
    fq = ("@long\n" + seq + "\n+\n" + "I" * len(seq) + "\n").encode()
    fq = fq + fq.replace(b"@long", b"@again")                      # twice: every k-mer has count 2
    for k in (31, 51):
        h = product(fq, k=k, min_count=1)
        o = run_oracle([fq], k=k, min_count=1)
        compare_all(h, o)
        assert h.total_instances == 2 * (len(seq) - k + 1)


@pytest.mark.parametrize("groups", [1, 3, 0])
def test_reads_of_mixed_lengths_many_tiles_per_wave(groups, monkeypatch):
    """Pass 1 gives every wave tiles of its own: 64 segments when they fit the wave's stage, fewer when they do not, a
    segment of more than 2048 k-mers in pieces of 160 k-mers (a lane per piece) — and the first tile of the next round
    travels in registers meanwhile.  Reads of 40 ... 9000 bases in random order, on one or three workgroups (SHK_PART_G)
    so that every wave walks many rounds, through the FASTQ entry and the packed entry: the oracle's counts and contigs."""
    import torch
    from sparrowhawk_amd import pack_fastq
    if groups:
        monkeypatch.setenv("SHK_PART_G", str(groups))
    rng = np.random.default_rng(4100 + groups)
    g = synth.random_genome(30000, 77)
    gs = synth.codes_to_str(g)
    lens = [40, 75, 151, 151, 151, 250, 300, 300, 700, 1500, 2078, 2079, 2400, 5200, 9000]
    recs = []
    for i in range(2600):
        L = int(lens[rng.integers(len(lens))])
        a = int(rng.integers(0, len(gs) - L))
        s = gs[a:a + L]
        if rng.integers(2):
            s = revcomp(s)
        recs.append(f"@r{i}\n{s}\n+\n{'I' * L}\n")
    fq = "".join(recs).encode()
    for k in (31, 51):
        o = run_oracle([fq], k=k, min_count=2)
        compare_all(product(fq, k=k, min_count=2), o)
        bases, seg, nb, nr = pack_fastq(fq, k, 20)
        d_bases = torch.from_numpy(bases.view(np.int32)).cuda()
        d_seg = torch.from_numpy(seg.view(np.int32)).cuda()
        torch.cuda.synchronize()
        h = AssemblyHelper.new(k, True, 2, 20, 0, False, False, False, False)
        h.preprocess_packed_device(d_bases.data_ptr(), d_seg.data_ptr(), len(seg) - 1, nb, nr)
        h.assemble()
        compare_all(h, o)


@pytest.mark.parametrize("P,genome", [(64, 3000), (1, 20000)])
def test_counts_saturate_at_u32_max(P, genome):
    """SPEC S4: counts are u32 and saturate at 0xFFFFFFFF.  Forced through the shard layer's run tables: the
    same packed records are listed 256 times as the runs of their partition (weights via repeated records),
    so a poly-A 31-mer seen 2^24 + a bit times counts past 2^32, while every other count is 256 x the oracle's.
    P = 1, 20 000 distinct k-mers: the one partition does not fit the LDS table AND holds >= 2^32 instances — it
    must not go through the k-mer-level repartition (no saturating add, 32-bit bucket cursors) but through the
    residue-class re-runs, which saturate."""
    import ctypes as C
    import torch
    from sparrowhawk_amd import pack_fastq
    k, reps = 31, 256
    n_polya = (1 << 24) // 120 + 50                       # 120 windows per 150-base read: just past 2^24 instances
    g, fq_small = make_dataset(genome, 20, seed=91)
    polya = b"".join(b"@a%d\n%s\n+\n%s\n" % (i, b"A" * 150, b"I" * 150) for i in range(n_polya))
    # (poly-A first: its record reaches pass 2's record table before the other records fill it, so it is counted by
    # multiplicity — expanded directly, 4.4 G same-address LDS atomics would take minutes)
    fq = polya + fq_small
    o = run_oracle([fq], k=k, min_count=0, min_qual=0)
    ok_, oc_ = o.solid()                                  # min_count 0: every distinct k-mer
    dev = torch.device("cuda", 0)
    bases, seg, nb, nr = pack_fastq(fq, k, 0)
    d_bases = torch.from_numpy(bases.view(np.int32)).to(dev)
    d_seg = torch.from_numpy(seg.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    h = AssemblyHelper.new(k, True, 0, 0, 0, False, False, False, False)
    L = h._L
    part = np.zeros(P, dtype=np.uint64)
    h._check(L.shk_shard_partition(h._h, d_bases.data_ptr(), d_seg.data_ptr(), len(seg) - 1, nb, nr, P, part.ctypes.data))
    rec_bytes = L.shk_shard_record_bytes(h._h)
    base = np.concatenate([[0], np.cumsum(part)[:-1]]).astype(np.uint64)
    send = torch.empty(int(part.sum()) * rec_bytes + 64, dtype=torch.uint8, device=dev)
    h._check(L.shk_shard_pack(h._h, send.data_ptr(), base.ctypes.data, P))
    run_off = np.ascontiguousarray(np.repeat(base[:, None], reps, axis=1))            # [P][reps]: the same run again and again
    run_cnt = np.ascontiguousarray(np.repeat(part[:, None], reps, axis=1).astype(np.uint32))
    assert int(run_cnt.sum(axis=1).max()) >= 1 << 26      # the poly-A partition takes the saturating path
    histo = np.zeros(500, dtype=np.uint64)
    inst = C.c_uint64(0)
    h._check(L.shk_shard_count(h._h, send.data_ptr(), run_off.ctypes.data, run_cnt.ctypes.data, P, reps, histo.ctypes.data, C.byref(inst)))
    assert inst.value == reps * o.total_instances
    if P == 1:
        assert h.timings().get("count_residue_rerun_x1", 0) == 1 and h.timings().get("count_repartitioned_x1", 0) == 0
    keys = (C.c_void_p * 1)()
    cnt = C.c_void_p()
    n_rows, used = C.c_uint64(0), C.c_uint32(0)
    h._check(L.shk_shard_rows(h._h, histo.ctypes.data, keys, C.byref(cnt), C.byref(n_rows), C.byref(used)))
    h._check(L.shk_shard_set_solid(h._h, keys, cnt, n_rows.value, inst.value))
    hk, hc, _ = sorted_table(*h.solid())
    assert np.array_equal(hk, ok_)
    expect = np.minimum(oc_.astype(np.uint64) * reps, 0xFFFFFFFF).astype(np.uint32)
    assert int((expect == 0xFFFFFFFF).sum()) == 1 and int(ok_[expect == 0xFFFFFFFF][0, 0]) == 0      # AAAA...A
    assert np.array_equal(hc, expect), "counts differ from min(256 x oracle, 2^32 - 1)"
    eh = np.zeros(500, dtype=np.uint64)
    np.add.at(eh, np.minimum(expect.astype(np.int64), 500) - 1, 1)
    assert np.array_equal(histo, eh)


def test_packed_reads_in_host_memory_entry_point():
    """shk_preprocess_packed_host (where SURVEY.md 8d starts the metric's clock): the packer's output handed over
    from host memory gives the same bytes as the FASTQ entry point and the oracle."""
    from sparrowhawk_amd import pack_fastq
    for k, err in ((31, 0.01), (51, 0.0)):
        g, fq = make_dataset(30000, 30, err=err, seed=640 + k)
        bases, seg, nb, nr = pack_fastq(fq, k, 20)
        h = AssemblyHelper.new(k, True, 3, 20, 0, False, False, False, False)
        h.preprocess_packed_host(bases.ctypes.data, seg.ctypes.data, len(seg) - 1, nb, nr)
        h.assemble()
        compare_all(h, run_oracle([fq], k=k, min_count=3))
        assert "h2d_packed_reads_MB" in h.timings()
        with pytest.raises(ShkError):
            h.preprocess_packed_host(bases.ctypes.data, seg.ctypes.data, len(seg) - 1, nb, nr)


@pytest.mark.parametrize("k,mode", [(31, "repartition"), (51, "repartition"), (51, "rescatter"), (31, "probe")])
def test_bloom_mode_overcounts_by_at_most_one_and_never_undercounts(k, mode):
    """do_bloom (docs/src/assembly.md:18: a Bloom pre-filter, "allowing for some degree of overcounting"; the UI then
    forces min_count >= 3, AssemblyPage.vue:430-432).  Here the filter sits in front of the k-mer-level repartition:
    singletons of error-rich partitions never reach HBM.  Parity with the oracle is statistical, in one direction:
    every stored count is the true count or one more; no k-mer above the threshold is lost; the instance total is
    exact; the histogram differs from the exact one only by what the false positives moved up a bin."""
    if mode == "probe":
        g, fq = make_dataset(400000, 45, err=0.02, seed=94)
        env = {"SHK_PART_P": 1024, "SHK_PROBE_PARTS": 128}
    else:
        g, fq = make_dataset(150000, 30, err=0.02, seed=92)
        env = {"SHK_PART_P": 64}
    if mode == "rescatter":
        env["SHK_OVF_CAP_PCT"] = 40
    mc = 3

    def run_bloom():
        hh = product(fq, k=k, min_count=mc, min_qual=0, do_bloom=True)
        return hh, hh.timings()
    h, t = _with_env(env, run_bloom)
    assert h.states[1] == "preprocess:bloom:start"
    o = run_oracle([fq], k=k, min_count=0, min_qual=0)
    ok_, oc_ = o.distinct()
    true = {tuple(r): int(c) for r, c in zip(ok_.tolist(), oc_.tolist())}
    sk, sc = h.solid()
    got = {tuple(r): int(c) for r, c in zip(sk.tolist(), sc.tolist())}
    assert t.get("bloom_singletons_never_stored_x1e-6", 0) > 0.1, t          # the filter really swallowed singletons
    n_over = 0
    for key, c in got.items():
        assert key in true and c in (true[key], true[key] + 1), (key, c, true.get(key))
        n_over += c != true[key]
    must = {key for key, c in true.items() if c > mc}
    may = {key for key, c in true.items() if c >= mc}
    assert must <= set(got) <= may
    assert n_over <= 0.25 * len(got)                      # (false positives: a minority — 2 bits in a 786 kbit filter per partition)
    assert h.total_instances == o.total_instances
    hist, ohist = h.histo().astype(np.int64), o.histo().astype(np.int64)
    assert abs(int(hist.sum()) - int(ohist.sum())) <= 0.02 * ohist.sum()     # distinct k-mers: new sightings, less false positives
    assert int((hist * np.arange(1, 501)).sum()) >= int((ohist * np.arange(1, 501)).sum())   # only ever moved up
    assert t["bloom_kmer_instances_MB_written"] < 0.8 * t["bloom_kmer_instances_MB_without_filter"]      # what went to HBM
    # contigs: the same sequences as the exact mode gives on these reads (kc may differ by the overcounts)
    e = _with_env(env, lambda: product(fq, k=k, min_count=mc, min_qual=0))
    # A k-mer seen exactly min_count times slips over the threshold when its first sighting was a false positive.  At
    # 2 % errors and min_count 3 a few such error k-mers become solid: one-node contigs, or a branch that cuts a
    # contig in two.  The assembly stays the same up to those (full size, 1 % errors, min_count 5: identical contigs,
    # tests/test_gpu_fullsize.py)
    slipped = len(set(got) - must)
    ch, ce = contig_set(h), contig_set(e)
    assert abs(sum(map(len, ch)) - sum(map(len, ce))) <= 0.01 * sum(map(len, ce)) + 3 * k * slipped
    assert len(ch ^ ce) <= 6 * slipped + 2, (len(ch ^ ce), slipped)
    if mode == "rescatter":
        assert t.get("count_repartitioned_x1", 0) > 0


def test_a_failure_after_counted_batches_poisons_the_handle():
    """A malformed record in a later chunk arrives after earlier batches went into the tables: the handle must not
    accept a second attempt (it would double-count them); every later call reports SHK_E_STATE until shk_free.
    A failure before anything was counted leaves the handle usable."""
    g, fq = make_dataset(20000, 20, seed=77)
    recs = fq.decode().split("@r")[1:]
    good = ("@r" + "@r".join(recs[:600])).encode()
    h = AssemblyHelper.new(31, True, 2, 20, 100, False, False, False, False)        # chunked: a batch every 100 reads
    h.push_reads(good)
    with pytest.raises(ShkError) as e:
        h.push_reads(b"@bad\nACGT\n+\nII\n")                                        # sequence / quality lengths differ
    assert e.value.code == -3
    for call in (lambda: h.push_reads(good), h.finish_reads, h.assemble):
        with pytest.raises(ShkError) as e2:
            call()
        assert e2.value.code == -2 and "failed earlier" in str(e2.value)
    h.free()
    # nothing counted yet: the same parse error leaves the handle fresh
    h = AssemblyHelper.new(31, True, 2, 20, 0, False, False, False, False)
    with pytest.raises(ShkError) as e:
        h.preprocess(b"@bad\nACGT\n+\nII\n")
    assert e.value.code == -3
    h.preprocess(fq)
    h.assemble()
    compare_all(h, run_oracle([fq], k=31, min_count=2))


@pytest.mark.parametrize("k,err,seed", [(31, 0.02, 1), (21, 0.03, 2), (51, 0.01, 3), (89, 0.005, 4)])
def test_device_writer_equals_the_host_writer(k, err, seed):
    """csrc/writer_gpu.h: the get_assembly() JSON of a fragmented assembly made on the device — contigs
    ordered by radix sort with ties settled by full comparison, links taken from the graph, FASTA / DOT / GFA1 / GFA2 records
    sized, scanned and written in place — must be the host writer's bytes and the oracle's.  Forced on small, error-rich,
    low-coverage inputs (hundreds to thousands of contigs, links, rings from plasmids)."""
    rng = np.random.default_rng(100 + seed)
    recs = []
    for gi in range(25):
        gl = int(rng.integers(3000, 9000))
        gm = synth.random_genome(gl, 700 + 31 * seed + gi)
        cov = float(np.exp(rng.normal(np.log(10.0), 0.8)))
        codes, quals = synth.sample_reads(gm, max(1, int(gl * cov / 150)), 150, 800 + 31 * seed + gi, err=err, circular=bool(gi % 4 == 0))
        recs.extend(synth.to_fastq(codes, quals).decode().split("@r")[1:])
    fq = ("@r" + "@r".join(recs)).encode()
    host = product(fq, k=k, min_count=1, min_qual=0)
    dev = _with_env({"SHK_DEVICE_WRITER_MIN": 1}, lambda: product(fq, k=k, min_count=1, min_qual=0))
    assert "device_writer_kernels" in dev.timings() and "device_writer_kernels" not in host.timings()
    j = json.loads(dev.get_assembly())
    assert j["ncontigs"] > 100 and (k > 31 or j["outgfa"].count("\nL\t") > 10)
    assert dev.get_assembly() == host.get_assembly()
    o = run_oracle([fq], k=k, min_count=1, min_qual=0)
    o.assemble()
    assert dev.get_assembly() == o.assembly_json()
