"""Seeded synthetic genomes and reads (SURVEY.md §8d inputs).  numpy only; no device code.

Codes: A=0 C=1 G=2 T=3.  Reads are sampled uniformly from both strands; substitution errors at
rate `err` are tagged with quality `q_err` (default Phred 10, '+'), correct bases with `q_ok`
(default Phred 40, 'I').
"""
import numpy as np

_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_genome(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 4, size=n, dtype=np.uint8)


def sample_reads(genome, n_reads, read_len, seed, err=0.0, q_ok=40, q_err=10, circular=False):
    """Returns (codes[n_reads, read_len] u8, quals[n_reads, read_len] u8 Phred+33 bytes)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    G = genome.shape[0]
    if circular:
        starts = rng.integers(0, G, size=n_reads)
        idx = (starts[:, None] + np.arange(read_len)[None, :]) % G
    else:
        starts = rng.integers(0, G - read_len + 1, size=n_reads)
        idx = starts[:, None] + np.arange(read_len)[None, :]
    codes = genome[idx]
    strand = rng.integers(0, 2, size=n_reads).astype(bool)
    codes[strand] = (3 - codes[strand])[:, ::-1]
    quals = np.full(codes.shape, q_ok + 33, dtype=np.uint8)
    if err > 0:
        e = rng.random(codes.shape) < err
        shift = rng.integers(1, 4, size=codes.shape, dtype=np.uint8)
        codes = np.where(e, (codes + shift) & 3, codes).astype(np.uint8)
        quals[e] = q_err + 33
    return np.ascontiguousarray(codes), quals


def to_fastq(codes, quals, prefix="r"):
    """FASTQ text (bytes) for code/quality matrices (all reads the same length)."""
    n, L = codes.shape
    seqs = _ASCII[codes]
    out = bytearray()
    for i in range(n):
        out += b"@" + prefix.encode() + str(i).encode() + b"\n"
        out += seqs[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n"
    return bytes(out)


def codes_to_str(codes):
    return _ASCII[np.asarray(codes, dtype=np.uint8)].tobytes().decode()


def revcomp_str(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))
