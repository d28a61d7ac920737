"""Hand-built inputs with results that follow from SPEC.md by reasoning, not from any
implementation: used to pin the oracle (CPU) and the HIP path (GPU) alike."""
import numpy as np

from sparrowhawk_amd import synth
from util import revcomp


def rand_seq(n, seed):
    return synth.codes_to_str(synth.random_genome(n, seed))


def fastq_of(seqs, qual_char="I"):
    out = []
    for i, s in enumerate(seqs):
        out.append(f"@r{i}\n{s}\n+\n{qual_char * len(s)}\n")
    return "".join(out).encode()


def canon(s):
    r = revcomp(s)
    return min(s, r)


def other_base(c):
    return {"A": "C", "C": "G", "G": "T", "T": "A"}[c]


def tip_case(k=15):
    """Main path M plus one short dead-end branch.  min_count=0: everything solid.
    With tip removal -> {M}; without -> M cut at the fork into 2 pieces + the tip."""
    M = rand_seq(160, 11)
    D = "".join(other_base(c) for c in M[80:88])          # diverges right after M[..80)
    tip_read = M[80 - 30:80] + D                          # shares 30 bases, then 8 novel bases
    reads = [M, M, tip_read]
    # fork node = M[80-k:80]; it keeps outdeg 2 without tip removal
    left = M[:80]                                          # unitig ends at the fork node
    right = M[80 - k + 1:]                                 # starts at the first node after the fork
    tip = M[80 - k + 1:80] + D
    return dict(k=k, fastq=fastq_of(reads), min_count=0,
                with_removal={canon(M)},
                without_removal={canon(left), canon(right), canon(tip)})


def bubble_case(k=15):
    """SNP bubble: 3 copies of M, 1 copy of M' (one substitution).  Bubble collapse keeps the
    higher-coverage branch -> {M}; without collapse -> 4 unitigs."""
    M = rand_seq(140, 12)
    p = 70
    M2 = M[:p] + other_base(M[p]) + M[p + 1:]
    reads = [M, M, M, M2]
    left = M[:p]                                           # ends at node M[p-k:p]
    right = M[p + 1:]                                      # starts at node M[p+1:p+1+k]
    b1 = M[p - k + 1:p + k]
    b2 = M2[p - k + 1:p + k]
    return dict(k=k, fastq=fastq_of(reads), min_count=0,
                with_collapse={canon(M)},
                without_collapse={canon(left), canon(right), canon(b1), canon(b2)})


def cycle_case(k=15, n=90):
    """Circular genome: one read spelling every cyclic k-mer.  One circular unitig, cut before
    the smallest canonical k-mer (SPEC S10)."""
    C = rand_seq(n, 13)
    read = C + C[:k - 1]
    from util import canonical_int, kmer_int
    # oriented nodes on the forward cycle and the mirrored cycle; the global minimum key is
    # (x_min, 0): find which strand spells x_min canonically and rotate that strand to start there
    fw = [(C + C)[i:i + k] for i in range(n)]
    rc_c = revcomp(C)
    bw = [(rc_c + rc_c)[i:i + k] for i in range(n)]
    best = min(range(n), key=lambda i: canonical_int(fw[i]))
    xmin = canonical_int(fw[best])
    if kmer_int(fw[best]) == xmin:
        strand, start = C, best
    else:
        j = next(i for i in range(n) if kmer_int(bw[i]) == xmin)
        strand, start = rc_c, j
    rot = strand[start:] + strand[:start]
    contig = rot + rot[:k - 1]
    return dict(k=k, fastq=fastq_of([read, read]), min_count=0, expect={canon(contig)})


def palindrome_case(k=15):
    """M = X + revcomp(X) is its own reverse complement: the k-mer at p and the one at
    len-k-p are the same node in opposite orientations, so the path folds back on itself through
    a hairpin link v -> rc(v), which is not simple (SPEC S10).  One unitig: the first half."""
    X = rand_seq(60, 14)
    M = X + revcomp(X)
    half = M[:(len(M) - k) // 2 + k]
    return dict(k=k, fastq=fastq_of([M, M]), min_count=0, expect={canon(half)})


def qual_mask_case(k=15):
    """A low-quality base splits a read into two segments; k-mers over it are not counted."""
    M = rand_seq(100, 15)
    qual = ["I"] * 100
    qual[50] = "+"                                          # Phred 10
    fq = f"@r\n{M}\n+\n{''.join(qual)}\n".encode() * 2
    return dict(k=k, fastq=fq, M=M, cut=50)
