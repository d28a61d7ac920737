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


# ---- device-side generator (torch on the GPU: plumbing for bench.py and the full-size tests) ----
class DeviceReads:
    """Packed reads resident in HBM in the layout shk_preprocess_packed_device takes, plus what the
    generator knows about them (for closed-form checks)."""
    def __init__(self):
        self.words = None       # int32[ceil(n_bases/16)+1]   2-bit packed stream
        self.seg_off = None     # int32[n_seg+1]
        self.n_seg = 0
        self.n_bases = 0
        self.n_reads = 0
        self.n_input_bases = 0
        self.genome = None      # int32[genome_len] codes
        self.starts = None      # int64[n_reads] (forward-strand start of each read)
        self.strand = None      # bool[n_reads]  (True: read is the reverse complement)
        self.err_fwd = None     # bool[n_reads, read_len] substitution flags in FORWARD-strand order (or None)
        self.instances = 0      # valid k-mer windows (SPEC S4), from the generator's own masks


def device_reads(torch, dev, genome_len, n_reads, read_len, k, seed, read_seed=None, err=0.0,
                 mask_errors=False, keep_meta=False, chunk=1 << 18):
    """Random genome + reads sampled uniformly from both strands, substitution errors at rate `err`.
    mask_errors=True reproduces min_qual masking (SPEC S2) of erroneous bases tagged with a low
    quality: reads are cut into their error-free segments (>= k bases), which gives ragged input."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    genome = torch.randint(0, 4, (genome_len,), generator=g, device=dev, dtype=torch.int32)
    if read_seed is not None:
        g.manual_seed(read_seed)
    ar = torch.arange(read_len, device=dev)
    out = DeviceReads()
    out.genome, out.n_reads, out.n_input_bases = genome, n_reads, n_reads * read_len
    code_chunks, len_chunks, metas = [], [], []
    instances = 0
    for r0 in range(0, n_reads, chunk):
        r1 = min(n_reads, r0 + chunk)
        R = r1 - r0
        starts = torch.randint(0, genome_len - read_len + 1, (R,), generator=g, device=dev)
        strand = torch.randint(0, 2, (R,), generator=g, device=dev).bool()
        codes = genome[starts[:, None] + ar[None, :]]
        e = None
        if err > 0:
            e = torch.rand((R, read_len), generator=g, device=dev) < err
            shift = torch.randint(1, 4, (R, read_len), generator=g, device=dev, dtype=torch.int32)
            codes = torch.where(e, (codes + shift) & 3, codes)
        if keep_meta:
            metas.append((starts, strand, e))
        codes = torch.where(strand[:, None], (3 - codes).flip(1), codes)
        if err > 0 and mask_errors:
            ev = torch.where(strand[:, None], e.flip(1), e)
            valid = ~ev
            prev = torch.zeros_like(valid)
            prev[:, 1:] = valid[:, :-1]
            first = (valid & ~prev).reshape(-1)
            v = valid.reshape(-1)
            run_id = torch.cumsum(first.to(torch.int64), 0) - 1
            n_runs = int(first.sum().item())
            lens = torch.bincount(run_id[v], minlength=n_runs)
            keep = lens >= k
            keep_base = v.clone()
            keep_base[v] = keep[run_id[v]]
            code_chunks.append(codes.reshape(-1)[keep_base].to(torch.int8))
            kl = lens[keep]
            len_chunks.append(kl)
            instances += int((kl - (k - 1)).sum().item())
        else:
            code_chunks.append(codes.reshape(-1).to(torch.int8))
            len_chunks.append(torch.full((R,), read_len, device=dev, dtype=torch.int64))
            instances += R * (read_len - k + 1)
    codes = torch.cat(code_chunks)
    lens = torch.cat(len_chunks)
    del code_chunks
    n_bases = codes.numel()
    pad = (-n_bases) % 16
    if pad:
        codes = torch.cat([codes, torch.zeros(pad, dtype=torch.int8, device=dev)])
    shifts = 2 * torch.arange(16, device=dev, dtype=torch.int32)
    words = torch.empty(codes.numel() // 16 + 1, dtype=torch.int32, device=dev)
    step = 1 << 26
    for b0 in range(0, codes.numel(), step):
        c = codes[b0:b0 + step].to(torch.int32).reshape(-1, 16)
        words[b0 // 16:b0 // 16 + c.shape[0]] = (c << shifts[None, :]).sum(dim=1, dtype=torch.int32)
    words[-1] = 0
    seg_off = torch.zeros(lens.numel() + 1, dtype=torch.int64, device=dev)
    seg_off[1:] = torch.cumsum(lens, 0)
    out.words, out.seg_off = words, seg_off.to(torch.int32)
    out.n_seg, out.n_bases, out.instances = int(lens.numel()), int(n_bases), instances
    if keep_meta:
        out.starts = torch.cat([m[0] for m in metas])
        out.strand = torch.cat([m[1] for m in metas])
        out.err_fwd = torch.cat([m[2] for m in metas]) if err > 0 else None
    if dev.type == "cuda":
        torch.cuda.synchronize()
    return out


def to_fastq_fixed(codes, quals):
    """Vectorised FASTQ text for large inputs: fixed-width records '@r%09d\\n<seq>\\n+\\n<qual>\\n'."""
    n, L = codes.shape
    rec = 11 + 1 + L + 1 + 2 + L + 1
    out = np.empty((n, rec), dtype=np.uint8)
    out[:, 0] = ord("@"); out[:, 1] = ord("r")
    ids = np.arange(n, dtype=np.int64)
    for d in range(9):
        out[:, 10 - d] = 48 + (ids // 10 ** d) % 10
    out[:, 11] = 10
    out[:, 12:12 + L] = _ASCII[codes]
    out[:, 12 + L] = 10; out[:, 13 + L] = ord("+"); out[:, 14 + L] = 10
    out[:, 15 + L:15 + 2 * L] = quals
    out[:, 15 + 2 * L] = 10
    return out.tobytes()


# ---- configs[3] / configs[4] inputs: many genomes, generated on the device from one seed -------------------
_M64 = (1 << 64) - 1


def _mix64(torch, x):
    """splitmix64 finaliser on int64 tensors (two's complement wrap-around = arithmetic mod 2^64)."""
    def c(v):                                   # python int -> the int64 with the same bit pattern
        return v - (1 << 64) if v >= (1 << 63) else v
    x = x + c(0x9E3779B97F4A7C15)
    x = (x ^ ((x >> 30) & ((1 << 34) - 1))) * c(0xBF58476D1CE4E5B9)
    x = (x ^ ((x >> 27) & ((1 << 37) - 1))) * c(0x94D049BB133111EB)
    return x ^ ((x >> 31) & ((1 << 33) - 1))


def substitution_flags(torch, read_idx, read_len, err, seed):
    """Substitution errors as a pure function of (seed, read index, position in the read as sequenced): returns
    (flag bool[R, L], shift int32[R, L] in 1..3).  Recomputable for any subset of reads, so a check can ask for
    the errors of exactly the reads that cover a position without the generator having kept 150 flags per read."""
    j = torch.arange(read_len, device=read_idx.device, dtype=torch.int64)
    u = _mix64(torch, (read_idx.to(torch.int64)[:, None] * read_len + j[None, :]) ^ (seed * 0x632BE59BD9B4E019 % (1 << 62)))
    thr = int(err * (1 << 40))
    flag = ((u >> 13) & ((1 << 40) - 1)) < thr
    shift = (((u >> 3) & 0x3FF) % 3 + 1).to(torch.int32)
    return flag, shift


class DeviceSample(DeviceReads):
    """Reads of MANY genomes (a batch of isolates pooled, or a metagenome) in HBM."""
    def __init__(self):
        super().__init__()
        self.genome_off = None   # int64[n_genomes+1] offsets into the concatenated genome codes (self.genome, int8)
        self.gid = None          # int32[n_reads] source genome of each read
        self.read_index0 = 0     # global index of this share's first read (errors are a function of the global index)
        self.err = 0.0
        self.err_seed = 0
        self.read_len = 0
        self.weights = None      # float64[n_genomes] read-sampling probability of each genome


def device_genomes(torch, dev, lengths, seed):
    """Concatenated uniform i.i.d. genomes (int8 codes) + offsets; the same on every rank for one seed."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    off = np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64))])
    total = int(off[-1])
    codes = torch.empty(total, dtype=torch.int8, device=dev)
    step = 1 << 28
    for b0 in range(0, total, step):
        n = min(step, total - b0)
        codes[b0:b0 + n] = torch.randint(0, 4, (n,), generator=g, device=dev, dtype=torch.int8)
    return codes, torch.from_numpy(off).to(dev)


def device_sample_reads(torch, dev, genomes, genome_off, weights, n_reads, read_len, k, seed, err=0.0,
                        read_index0=0, chunk=1 << 18, circular=False):
    """n_reads reads (global indices read_index0 ...) drawn from the genomes with probabilities `weights`
    (numpy float64, sums to 1): uniform start, random strand, substitution errors at rate `err` that are a pure
    function of (seed, global read index, position).  Every read is one segment (no quality masking).
    Every rank of a sharded run calls this with its own read_index0 and gets its share of ONE sample."""
    out = DeviceSample()
    out.genome, out.genome_off = genomes, genome_off
    out.n_reads = out.n_seg = int(n_reads)
    out.n_bases = out.n_input_bases = int(n_reads) * read_len
    out.read_index0, out.err, out.err_seed, out.read_len = int(read_index0), float(err), int(seed), read_len
    out.weights = np.asarray(weights, dtype=np.float64)
    out.instances = int(n_reads) * (read_len - k + 1)
    lens = (genome_off[1:] - genome_off[:-1])
    cdf = torch.from_numpy(np.cumsum(out.weights)).to(dev)
    ar = torch.arange(read_len, device=dev)
    n_words = (out.n_bases + 15) // 16 + 1
    words = torch.zeros(n_words, dtype=torch.int32, device=dev)
    shifts = 2 * torch.arange(16, device=dev, dtype=torch.int32)
    gids, starts, strands = [], [], []
    assert (chunk * read_len) % 16 == 0
    for r0 in range(0, int(n_reads), chunk):
        R = min(chunk, int(n_reads) - r0)
        idx = torch.arange(read_index0 + r0, read_index0 + r0 + R, device=dev, dtype=torch.int64)
        # three independent uniforms per read from the read's global index: genome, start, strand
        u1 = _mix64(torch, idx ^ (seed * 0x2545F4914F6CDD1D % (1 << 62)))
        u2 = _mix64(torch, u1 ^ 0x5851F42D4C957F2D)
        u3 = _mix64(torch, u2 ^ 0x14057B7EF767814F)
        f1 = ((u1 >> 11) & ((1 << 53) - 1)).to(torch.float64) / float(1 << 53)
        gid = torch.clamp(torch.searchsorted(cdf, f1, right=True), max=cdf.numel() - 1)
        span = lens[gid] if circular else lens[gid] - read_len + 1        # circular replicons: a read may start anywhere
        f2 = ((u2 >> 11) & ((1 << 53) - 1)).to(torch.float64) / float(1 << 53)
        rel = torch.clamp((f2 * span.to(torch.float64)).to(torch.int64), max=span - 1)
        st = genome_off[gid] + rel
        strand = ((u3 >> 17) & 1).bool()
        if circular:
            codes = genomes[genome_off[gid][:, None] + (rel[:, None] + ar[None, :]) % lens[gid][:, None]].to(torch.int32)
        else:
            codes = genomes[st[:, None] + ar[None, :]].to(torch.int32)
        codes = torch.where(strand[:, None], (3 - codes).flip(1), codes)       # as sequenced
        if err > 0:
            flag, shift = substitution_flags(torch, idx, read_len, err, seed)
            codes = torch.where(flag, (codes + shift) & 3, codes)
        flat = codes.reshape(-1)
        pad = (-flat.numel()) % 16
        if pad:
            flat = torch.cat([flat, torch.zeros(pad, dtype=torch.int32, device=dev)])
        w = (flat.reshape(-1, 16) << shifts[None, :]).sum(dim=1, dtype=torch.int32)
        w0 = (r0 * read_len) // 16
        words[w0:w0 + w.numel()] = w
        gids.append(gid.to(torch.int32)); starts.append(st); strands.append(strand)
    out.words = words
    out.seg_off = (torch.arange(int(n_reads) + 1, device=dev, dtype=torch.int64) * read_len).to(torch.int32)
    out.gid, out.starts, out.strand = torch.cat(gids), torch.cat(starts), torch.cat(strands)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    return out


def metagenome_spec(n_genomes=2000, mean_len=3_000_000, sigma=1.0, seed=0xEC05):
    """SURVEY.md 8d cfg 5: genome lengths ~3 Mbp (uniform +-20 %), log-normal abundance (sigma 1).  A genome's
    share of the reads is proportional to abundance x length.  numpy only (identical on every rank)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lengths = rng.integers(int(mean_len * 0.8), int(mean_len * 1.2) + 1, size=n_genomes).astype(np.int64)
    abundance = np.exp(rng.normal(0.0, sigma, size=n_genomes))
    w = abundance * lengths
    return lengths, w / w.sum()


def isolate_batch_spec(n_isolates=96, lo=2_000_000, hi=7_000_000, seed=0xEC04):
    """SURVEY.md 8d cfg 4: 96 genomes with lengths uniform in [2 M, 7 M] bp; isolate i is generated from seed + i."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(lo, hi + 1, size=n_isolates).astype(np.int64)


def device_fastq_fixed(torch, codes, qual_char=ord("I"), low_mask=None, low_char=ord("+")):
    """FASTQ text built ON THE DEVICE for large inputs (bench.py's FASTQ leg): fixed-width records
    '@r%09d\\n<seq>\\n+\\n<qual>\\n' — the layout of to_fastq_fixed.  codes: int tensor [n, L] of base codes on the device;
    low_mask (optional, bool [n, L]): bases that get the low quality character.  Returns a uint8 device tensor."""
    n, L = codes.shape
    dev = codes.device
    rec = 11 + 1 + L + 1 + 2 + L + 1
    out = torch.empty((n, rec), dtype=torch.uint8, device=dev)
    out[:, 0] = ord("@"); out[:, 1] = ord("r")
    ids = torch.arange(n, device=dev, dtype=torch.int64)
    for d in range(9):
        out[:, 2 + d] = ((ids // (10 ** (8 - d))) % 10 + 48).to(torch.uint8)
    out[:, 11] = 10
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    out[:, 12:12 + L] = lut[codes.long()]
    out[:, 12 + L] = 10
    out[:, 13 + L] = ord("+"); out[:, 14 + L] = 10
    out[:, 15 + L:15 + 2 * L] = qual_char
    if low_mask is not None:
        out[:, 15 + L:15 + 2 * L][low_mask] = low_char
    out[:, 15 + 2 * L] = 10
    return out.reshape(-1)
