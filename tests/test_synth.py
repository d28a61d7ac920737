"""CPU tests of the many-genome generator (sparrowhawk_amd/synth.py) that feeds the configs[3] / configs[4]
tests and bench workloads: every rank of a sharded run must see its share of ONE sample, and a check must be
able to recompute the substitution flags of any read."""
import numpy as np
import torch

from sparrowhawk_amd import synth

DEV = torch.device("cpu")


def unpack(d):
    sh = 2 * torch.arange(16, dtype=torch.int32)
    w = d.words[: (d.n_bases + 15) // 16]
    return ((w[:, None] >> sh[None, :]) & 3).reshape(-1)[: d.n_bases]


def test_shares_of_a_sample_concatenate_to_the_sample():
    lens, w = synth.metagenome_spec(20, 5000, 1.0, 1)
    assert abs(w.sum() - 1) < 1e-12 and lens.min() >= 4000 and lens.max() <= 6000
    g, off = synth.device_genomes(torch, DEV, lens, 3)
    g2, _ = synth.device_genomes(torch, DEV, lens, 3)
    assert torch.equal(g, g2)
    whole = synth.device_sample_reads(torch, DEV, g, off, w, 1000, 150, 31, 7, err=0.01, chunk=256)
    parts = [synth.device_sample_reads(torch, DEV, g, off, w, n, 150, 31, 7, err=0.01, read_index0=i0, chunk=128)
             for i0, n in ((0, 300), (300, 450), (750, 250))]
    assert torch.equal(unpack(whole), torch.cat([unpack(p) for p in parts]))
    assert torch.equal(whole.starts, torch.cat([p.starts for p in parts]))
    assert whole.instances == 1000 * 120 and whole.n_bases == 150_000
    # reads follow the abundance: the most abundant genome collects the most reads
    counts = np.bincount(whole.gid.numpy(), minlength=20)
    assert counts.argmax() in np.argsort(-w)[:3]


def test_reads_are_the_genome_with_the_recomputable_substitutions():
    lens = synth.isolate_batch_spec(3, 3000, 6000, 5)
    g, off = synth.device_genomes(torch, DEV, [int(lens[1])], 5 + 1)
    d = synth.device_sample_reads(torch, DEV, g, off, np.array([1.0]), 500, 150, 31, 11, err=0.02, read_index0=1234)
    codes = unpack(d).reshape(500, 150)
    flag, shift = synth.substitution_flags(torch, torch.arange(1234, 1734), 150, 0.02, 11)
    assert 0.012 < flag.float().mean().item() < 0.028 and int(shift.min()) == 1 and int(shift.max()) == 3
    ar = torch.arange(150)
    ref = g[d.starts[:, None] + ar[None, :]].to(torch.int32)
    ref = torch.where(d.strand[:, None], (3 - ref).flip(1), ref)
    assert torch.equal(codes != ref, flag)                  # a flagged base always differs, an unflagged one never
    assert torch.equal(codes, torch.where(flag, (ref + shift) & 3, ref))
    assert int(d.starts.min()) >= 0 and int(d.starts.max()) <= int(lens[1]) - 150
