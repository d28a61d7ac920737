// writer_gpu.h — `assembly:saving` on the device for FRAGMENTED assemblies (SPEC S11; /root/reference/www/src/workers/
// Assembler.ts:7-13,127: get_assembly() -> {"outfasta","ncontigs","outdot","outgfa","outgfav2"}).
// (included by pipeline.hip inside namespace shk, after collapse.h; hipcub is included by pipeline.hip)
//
// An isolate leaves a handful of contigs and the host writer (outputs.cpp) turns them into text in 0.15 ms.  A metagenome
// leaves millions: ordering them, finding their links and writing a gigabyte of JSON took 0.75-1.0 s on the 16 host cores
// of one GPU's share against 0.6 s for everything on the device (configs[4] share, round 3).  Here the same bytes are made
// on the device: contigs ordered by a radix sort on (length, first 16 bases) with the rare ties settled by full comparison,
// links taken from the GRAPH (the last node of a contig knows its out-neighbours, and those are first nodes of chains) instead
// of hashing contig ends, every record's size computed from its numbers, exclusive scans for the offsets, records and
// sequences written in place — the host copies one buffer.
#pragma once

struct WContig { unsigned long long off, kc; uint32_t len /* bases */, slot /* chain record */; };

__device__ __forceinline__ uint32_t w_ndig(unsigned long long v) { uint32_t n = 1; while (v >= 10ull) { v /= 10ull; n++; } return n; }
// the two sinks of the host writer (outputs.cpp): one measures, one writes; NL / TAB / QUOTE are JSON-escaped (two characters)
struct WSize {
    unsigned long long n = 0;
    __device__ void seq(unsigned long long m) { n += m; }
    __device__ void nl() { n += 2; }
    __device__ void tab() { n += 2; }
    __device__ void quote() { n += 2; }
    __device__ void raw(const char *, uint32_t m) { n += m; }
    __device__ void ch(char) { n += 1; }
    __device__ void num(unsigned long long v) { n += w_ndig(v); }
};
struct WPtr {
    char *p;
    __device__ void seq(unsigned long long m) { p += m; }                  // (sequences are copied by k_w_copy_seqs)
    __device__ void nl() { *p++ = '\\'; *p++ = 'n'; }
    __device__ void tab() { *p++ = '\\'; *p++ = 't'; }
    __device__ void quote() { *p++ = '\\'; *p++ = '"'; }
    __device__ void raw(const char *q, uint32_t m) { for (uint32_t i = 0; i < m; i++) p[i] = q[i]; p += m; }
    __device__ void ch(char c) { *p++ = c; }
    __device__ void num(unsigned long long v) { const uint32_t m = w_ndig(v); for (uint32_t i = m; i-- > 0;) { p[i] = (char)('0' + (int)(v % 10ull)); v /= 10ull; } p += m; }
};
// the records, exactly as outputs.cpp spells them; *_pre = the part in front of the sequence
template <class S> __device__ void w_fasta_pre(S &w, unsigned long long i1, unsigned long long len, unsigned long long kc) {
    w.raw(">contig_", 8); w.num(i1); w.raw(" len=", 5); w.num(len); w.raw(" kc=", 4); w.num(kc); w.nl();
}
template <class S> __device__ void w_fasta(S &w, unsigned long long i1, unsigned long long len, unsigned long long kc) { w_fasta_pre(w, i1, len, kc); w.seq(len); w.nl(); }
template <class S> __device__ void w_dotn(S &w, unsigned long long i1, unsigned long long len, unsigned long long kc) {
    w.raw("  ", 2); w.quote(); w.num(i1); w.quote(); w.raw(" [label=", 8); w.quote(); w.num(i1); w.raw(" len=", 5); w.num(len); w.raw(" kc=", 4); w.num(kc);
    w.quote(); w.raw("];", 2); w.nl();
}
template <class S> __device__ void w_g1s_pre(S &w, unsigned long long i1) { w.ch('S'); w.tab(); w.num(i1); w.tab(); }
template <class S> __device__ void w_g1s(S &w, unsigned long long i1, unsigned long long len, unsigned long long kc) {
    w_g1s_pre(w, i1); w.seq(len); w.tab(); w.raw("LN:i:", 5); w.num(len); w.tab(); w.raw("KC:i:", 5); w.num(kc); w.nl();
}
template <class S> __device__ void w_g2s_pre(S &w, unsigned long long i1, unsigned long long len) { w.ch('S'); w.tab(); w.num(i1); w.tab(); w.num(len); w.tab(); }
template <class S> __device__ void w_g2s(S &w, unsigned long long i1, unsigned long long len, unsigned long long kc) {
    w_g2s_pre(w, i1, len); w.seq(len); w.tab(); w.raw("KC:i:", 5); w.num(kc); w.nl();
}
// a link: a:31 | ao:1 | b:31 | bo:1 (most significant first: sorting the word sorts the tuples), contigs numbered from 1
__device__ __forceinline__ unsigned long long w_link_pack(uint32_t a, uint32_t ao, uint32_t b, uint32_t bo) {
    return ((unsigned long long)a << 33) | ((unsigned long long)ao << 32) | ((unsigned long long)b << 1) | bo;
}
template <class S> __device__ void w_dotl(S &w, unsigned long long L) {
    const unsigned long long a = L >> 33, b = (L >> 1) & 0x7FFFFFFFull; const uint32_t ao = (uint32_t)(L >> 32) & 1u, bo = (uint32_t)L & 1u;
    w.raw("  ", 2); w.quote(); w.num(a); w.quote(); w.raw(" -> ", 4); w.quote(); w.num(b); w.quote(); w.raw(" [label=", 8); w.quote();
    w.ch(ao ? '-' : '+'); w.ch(bo ? '-' : '+'); w.quote(); w.raw("];", 2); w.nl();
}
template <class S> __device__ void w_g1l(S &w, unsigned long long L, uint32_t ov) {
    const unsigned long long a = L >> 33, b = (L >> 1) & 0x7FFFFFFFull; const uint32_t ao = (uint32_t)(L >> 32) & 1u, bo = (uint32_t)L & 1u;
    w.ch('L'); w.tab(); w.num(a); w.tab(); w.ch(ao ? '-' : '+'); w.tab(); w.num(b); w.tab(); w.ch(bo ? '-' : '+'); w.tab(); w.num(ov); w.ch('M'); w.nl();
}
template <class S> __device__ void w_g2l(S &w, unsigned long long L, uint32_t ov, unsigned long long la, unsigned long long lb) {
    const unsigned long long a = L >> 33, b = (L >> 1) & 0x7FFFFFFFull; const uint32_t ao = (uint32_t)(L >> 32) & 1u, bo = (uint32_t)L & 1u;
    w.ch('E'); w.tab(); w.ch('*'); w.tab(); w.num(a); w.ch(ao ? '-' : '+'); w.tab(); w.num(b); w.ch(bo ? '-' : '+'); w.tab();
    if (!ao) { w.num(la - ov); w.tab(); w.num(la); w.ch('$'); w.tab(); }
    else { w.ch('0'); w.tab(); w.num(ov); if ((unsigned long long)ov == la) w.ch('$'); w.tab(); }
    if (!bo) { w.ch('0'); w.tab(); w.num(ov); if ((unsigned long long)ov == lb) w.ch('$'); w.tab(); }
    else { w.num(lb - ov); w.tab(); w.num(lb); w.ch('$'); w.tab(); }
    w.num(ov); w.ch('M'); w.nl();
}

// ---- which chains are emitted and where (a fragmented assembly has millions of chain records: they stay on the device) ----
__global__ __launch_bounds__(256) void k_w_plan_sizes(const HeadRec *__restrict__ heads, uint32_t n_heads, uint32_t k,
                                                      unsigned long long *__restrict__ sz, unsigned long long *__restrict__ fl) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_heads; i += gridDim.x * blockDim.x) {
        const HeadRec h = heads[i];
        sz[i] = h.emit ? h.len + (unsigned long long)(k - 1u) : 0ull;
        fl[i] = h.emit ? 1ull : 0ull;
    }
}
__global__ __launch_bounds__(256) void k_w_plan_fill(const HeadRec *__restrict__ heads, uint32_t n_heads, uint32_t k,
                                                     const unsigned long long *__restrict__ off, const unsigned long long *__restrict__ idx,
                                                     EmitRec *__restrict__ head_off, WContig *__restrict__ c, uint32_t *__restrict__ flags) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_heads; i += gridDim.x * blockDim.x) {
        const HeadRec h = heads[i];
        EmitRec e; e.off = ~0ull; e.rot = h.rot; e.len = (uint32_t)h.len;
        if (h.emit) {
            e.off = off[i];
            const unsigned long long bases = h.len + (unsigned long long)(k - 1u);
            if (bases > 0xFFFFFFFFull) flags[0] = 9;
            WContig x; x.off = off[i]; x.kc = h.kc; x.len = (uint32_t)bases; x.slot = i;
            c[idx[i]] = x;
        }
        head_off[i] = e;
    }
}

__device__ __forceinline__ uint32_t w_code(char c) { return c == 'C' ? 1u : (c == 'G' ? 2u : (c == 'T' ? 3u : 0u)); }

// ---- order: (length descending, sequence ascending) ----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_w_keys(const char *__restrict__ text, const WContig *__restrict__ c, uint32_t n,
                                                unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const WContig x = c[i];
        uint32_t pf = 0;
        const uint32_t m = x.len < 16u ? x.len : 16u;
        for (uint32_t j = 0; j < m; j++) pf |= w_code(text[x.off + j]) << (2 * (15 - j));
        keys[i] = ((unsigned long long)(0xFFFFFFFFu - x.len) << 32) | pf;
        vals[i] = i;
    }
}
// contigs of one length that share their first 16 bases (a few in millions): the first thread of such a run orders it by
// the full spelling
__global__ __launch_bounds__(256) void k_w_ties(const char *__restrict__ text, const WContig *__restrict__ c, uint32_t n,
                                                const unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        if ((i > 0 && keys[i - 1] == k) || i + 1 >= n || keys[i + 1] != k) continue;
        uint32_t e = i + 1;
        while (e < n && keys[e] == k) e++;
        const uint32_t len = c[vals[i]].len;
        for (uint32_t a = i + 1; a < e; a++) {                             // insertion sort: runs are 2-3 long
            const uint32_t va = vals[a];
            uint32_t b = a;
            while (b > i) {
                const uint32_t vb = vals[b - 1];
                const char *pa = text + c[va].off, *pb = text + c[vb].off;
                int cmp = 0;
                for (uint32_t j = 16; j < len && cmp == 0; j++) cmp = (int)(unsigned char)pa[j] - (int)(unsigned char)pb[j];
                if (cmp == 0) cmp = va < vb ? -1 : 1;                      // (equal spellings cannot occur; keeps the order total)
                if (cmp >= 0) break;
                vals[b] = vb; b--;
            }
            vals[b] = va;
        }
    }
}
// sorted position -> {length, kc, source} and the inverse (chain record -> sorted position + 1, 0: not emitted)
__global__ __launch_bounds__(256) void k_w_rank(const WContig *__restrict__ c, uint32_t n, const uint32_t *__restrict__ vals,
                                                uint32_t *__restrict__ rank_of_slot, unsigned long long *__restrict__ slen,
                                                unsigned long long *__restrict__ skc, uint32_t *__restrict__ rank_of_contig) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const WContig x = c[vals[r]];
        rank_of_slot[x.slot] = r + 1u;
        rank_of_contig[vals[r]] = r;
        slen[r] = x.len; skc[r] = x.kc;
    }
}

// ---- links from the graph (SPEC S11): for a contig c and orientation o, every edge from its last oriented node to the
// first oriented node of (c', o'); a link and its mirror (c', !o', c, !o) are one, written as the smaller tuple
template <int W>
__global__ __launch_bounds__(256) void k_w_links(Graph<W> g, const HeadRec *__restrict__ heads, const uint2 *__restrict__ ol,
                                                 const WContig *__restrict__ c, uint32_t n, const uint32_t *__restrict__ vals,
                                                 const uint32_t *__restrict__ rank_of_slot, unsigned long long *__restrict__ links,
                                                 unsigned int *__restrict__ n_links, uint32_t cap, uint32_t *__restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const uint32_t total = 2u * n;
    const uint32_t n_round = (total + 63u) & ~63u;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_round; t += gridDim.x * blockDim.x) {
        unsigned long long found[4]; uint32_t nf = 0;
        if (t < total) {
            const uint32_t r = t >> 1, o = t & 1u;
            const HeadRec h = heads[c[vals[r]].slot];
            if (h.circ) { if (o == 0) found[nf++] = w_link_pack(r + 1u, 0u, r + 1u, 0u); }       // a ring closes on itself: its one link
            else {
                const uint32_t v = o ? (h.head_node ^ 1u) : h.tail_node;                      // the last oriented node of (c, o)
                const uint32_t om = g.outmask(v);
                for (uint32_t b = 0; b < 4; b++) {
                    if (!((om >> b) & 1u)) continue;
                    const uint32_t u = g.follow(v, b);
                    if (u == NIL) { flags[0] = 1; continue; }
                    const uint2 pos = ol[u];
                    if (pos.x == NIL || pos.y != 0u) { flags[0] = 2; continue; }                 // (an out-neighbour of a chain's last node starts a chain)
                    uint32_t rj = rank_of_slot[pos.x], oj = 0;
                    if (rj == 0u) {                                                            // that strand is not the emitted one: its mirror chain is
                        const uint2 mp = ol[heads[pos.x].tail_node ^ 1u];
                        if (mp.x == NIL) { flags[0] = 3; continue; }
                        rj = rank_of_slot[mp.x]; oj = 1;
                        if (rj == 0u) { flags[0] = 4; continue; }
                    }
                    const unsigned long long L = w_link_pack(r + 1u, o, rj, oj), M = w_link_pack(rj, oj ^ 1u, r + 1u, o ^ 1u);
                    found[nf++] = M < L ? M : L;
                }
            }
        }
        // wave-aggregated append
        uint32_t incl = nf;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, d); if (lane >= d) incl += x; }
        const uint32_t tot = (uint32_t)__shfl((int)incl, 63);
        uint32_t base = 0;
        if (lane == 63 && tot) base = atomicAdd(n_links, tot);
        base = (uint32_t)__shfl((int)base, 63);
        const uint32_t at = base + incl - nf;
        for (uint32_t q = 0; q < nf; q++) if (at + q < cap) links[at + q] = found[q];
    }
}

// ---- sizes of the records (offsets by exclusive scans), then the records themselves -----------------------------------
__global__ __launch_bounds__(256) void k_w_contig_sizes(uint32_t n, const unsigned long long *__restrict__ slen, const unsigned long long *__restrict__ skc,
                                                        unsigned long long *__restrict__ s_fasta, unsigned long long *__restrict__ s_dotn,
                                                        unsigned long long *__restrict__ s_g1s, unsigned long long *__restrict__ s_g2s) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const unsigned long long len = slen[r], kc = skc[r], i1 = (unsigned long long)r + 1ull;
        { WSize w; w_fasta(w, i1, len, kc); s_fasta[r] = w.n; }
        { WSize w; w_dotn(w, i1, len, kc); s_dotn[r] = w.n; }
        { WSize w; w_g1s(w, i1, len, kc); s_g1s[r] = w.n; }
        { WSize w; w_g2s(w, i1, len, kc); s_g2s[r] = w.n; }
    }
}
__global__ __launch_bounds__(256) void k_w_link_sizes(uint32_t n, const unsigned long long *__restrict__ links, uint32_t ov,
                                                      const unsigned long long *__restrict__ slen,
                                                      unsigned long long *__restrict__ s_dotl, unsigned long long *__restrict__ s_g1l,
                                                      unsigned long long *__restrict__ s_g2l) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const unsigned long long L = links[j];
        const unsigned long long la = slen[(L >> 33) - 1ull], lb = slen[((L >> 1) & 0x7FFFFFFFull) - 1ull];
        { WSize w; w_dotl(w, L); s_dotl[j] = w.n; }
        { WSize w; w_g1l(w, L, ov); s_g1l[j] = w.n; }
        { WSize w; w_g2l(w, L, ov, la, lb); s_g2l[j] = w.n; }
    }
}
struct WBases { unsigned long long fasta, dotn, dotl, g1s, g1l, g2s, g2l; };     // where each section's records start in the JSON
__global__ __launch_bounds__(256) void k_w_contig_recs(uint32_t n, const unsigned long long *__restrict__ slen, const unsigned long long *__restrict__ skc,
                                                       const unsigned long long *__restrict__ o_fasta, const unsigned long long *__restrict__ o_dotn,
                                                       const unsigned long long *__restrict__ o_g1s, const unsigned long long *__restrict__ o_g2s,
                                                       WBases B, char *__restrict__ js) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const unsigned long long len = slen[r], kc = skc[r], i1 = (unsigned long long)r + 1ull;
        { WPtr w{js + B.fasta + o_fasta[r]}; w_fasta(w, i1, len, kc); }
        { WPtr w{js + B.dotn + o_dotn[r]}; w_dotn(w, i1, len, kc); }
        { WPtr w{js + B.g1s + o_g1s[r]}; w_g1s(w, i1, len, kc); }
        { WPtr w{js + B.g2s + o_g2s[r]}; w_g2s(w, i1, len, kc); }
    }
}
__global__ __launch_bounds__(256) void k_w_link_recs(uint32_t n, const unsigned long long *__restrict__ links, uint32_t ov,
                                                     const unsigned long long *__restrict__ slen, const unsigned long long *__restrict__ o_dotl,
                                                     const unsigned long long *__restrict__ o_g1l, const unsigned long long *__restrict__ o_g2l,
                                                     WBases B, char *__restrict__ js) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const unsigned long long L = links[j];
        const unsigned long long la = slen[(L >> 33) - 1ull], lb = slen[((L >> 1) & 0x7FFFFFFFull) - 1ull];
        { WPtr w{js + B.dotl + o_dotl[j]}; w_dotl(w, L); }
        { WPtr w{js + B.g1l + o_g1l[j]}; w_g1l(w, L, ov); }
        { WPtr w{js + B.g2l + o_g2l[j]}; w_g2l(w, L, ov, la, lb); }
    }
}
// the sequences: every 8 bytes of the contig text go to their three places (FASTA, GFA1, GFA2).  c is in text order
// (offsets ascending): the contig of a byte is found by bisection, its record by its sorted position.
__global__ __launch_bounds__(256) void k_w_copy_seqs(const char *__restrict__ text, unsigned long long n_bytes, const WContig *__restrict__ c, uint32_t n,
                                                     const uint32_t *__restrict__ rank_of_contig, const unsigned long long *__restrict__ o_fasta,
                                                     const unsigned long long *__restrict__ o_g1s, const unsigned long long *__restrict__ o_g2s,
                                                     const unsigned long long *__restrict__ skc, WBases B, char *__restrict__ js) {
    const unsigned long long n_chunks = (n_bytes + 7ull) / 8ull;
    for (unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q < n_chunks; q += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long p = q * 8ull;
        const unsigned long long pe = p + 8ull < n_bytes ? p + 8ull : n_bytes;
        uint32_t lo = 0, hi = n;                                   // the contig holding byte p
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (c[mid].off <= p) lo = mid; else hi = mid; }
        while (p < pe) {
            const WContig x = c[lo];
            const uint32_t r = rank_of_contig[lo];
            const unsigned long long i1 = (unsigned long long)r + 1ull, kc = skc[r];
            WSize a; w_fasta_pre(a, i1, x.len, kc);
            WSize b; w_g1s_pre(b, i1);
            WSize d; w_g2s_pre(d, i1, x.len);
            char *df = js + B.fasta + o_fasta[r] + a.n, *d1 = js + B.g1s + o_g1s[r] + b.n, *d2 = js + B.g2s + o_g2s[r] + d.n;
            const unsigned long long end = x.off + x.len < pe ? x.off + x.len : pe;
            for (; p < end; p++) { const char ch = text[p]; const unsigned long long j = p - x.off; df[j] = ch; d1[j] = ch; d2[j] = ch; }
            lo++;
        }
    }
}
