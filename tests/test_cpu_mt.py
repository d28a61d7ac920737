"""The multi-threaded CPU baseline (oracle/cpu_mt.cpp, timed by bench.py's cpu_baseline leg) against the
single-threaded oracle: same solid set, histogram and FASTA for one and for several threads, with errors,
correction switches, two-word keys and circular replicons.  CPU only."""
import numpy as np
import pytest

from oracle import CpuMt
from sparrowhawk_amd import pack_fastq, synth
from util import make_dataset, run_oracle, sorted_table


def run_mt(fq, k, min_count, threads, min_qual=20, **flags):
    bases, seg, nb, nr = pack_fastq(fq, k, min_qual)
    m = CpuMt(k, threads)
    m.count(bases, seg, emit_threshold=min_count)
    m.filter(min_count)
    m.assemble(**flags)
    return m


@pytest.mark.parametrize("k,err,threads", [(31, 0.0, 1), (31, 0.01, 4), (51, 0.01, 3), (21, 0.02, 8), (63, 0.005, 2)])
def test_cpu_mt_equals_the_oracle(k, err, threads):
    g, fq = make_dataset(30000, 30, err=err, seed=400 + k)
    for min_qual, min_count in ((20, 3), (0, 1)):
        o = run_oracle([fq], k=k, min_count=min_count, min_qual=min_qual)
        m = run_mt(fq, k, min_count, threads, min_qual=min_qual)
        assert m.total_instances == o.total_instances
        assert np.array_equal(m.histo(), o.histo())
        mk, mc = m.solid()
        ok_, oc_ = o.solid()
        assert np.array_equal(mk, ok_) and np.array_equal(mc, oc_)
        o.assemble()
        assert m.fasta() == o.fasta()


@pytest.mark.parametrize("flags", [dict(no_bubble_collapse=True), dict(no_dead_end_removal=True),
                                   dict(no_bubble_collapse=True, no_dead_end_removal=True)])
def test_cpu_mt_correction_switches(flags):
    g, fq = make_dataset(20000, 25, err=0.02, seed=77)
    o = run_oracle([fq], k=31, min_count=1, min_qual=0, **flags)
    o.assemble()
    m = run_mt(fq, 31, 1, 4, min_qual=0, **flags)
    assert m.fasta() == o.fasta() and m.n_contigs > 3


def test_cpu_mt_circular_replicons():
    texts = []
    for j, n in enumerate((45, 300, 5000)):
        g = synth.random_genome(n, 600 + j)
        codes, quals = synth.sample_reads(g, max(60, n * 30 // 100), 100, 650 + j, circular=True)
        texts.append(synth.to_fastq(codes, quals, prefix=f"c{j}_"))
    g = synth.random_genome(3000, 699)
    codes, quals = synth.sample_reads(g, 900, 100, 698)
    texts.append(synth.to_fastq(codes, quals, prefix="l_"))
    fq = b"".join(texts)
    for k in (31, 41):
        o = run_oracle([fq], k=k, min_count=1)
        o.assemble()
        m = run_mt(fq, k, 1, 3)
        assert m.fasta() == o.fasta()


def test_cpu_mt_is_independent_of_the_thread_count():
    g, fq = make_dataset(40000, 20, err=0.01, seed=5)
    outs = {run_mt(fq, 31, 2, t, min_qual=0).fasta() for t in (1, 2, 5, 8)}
    assert len(outs) == 1
