// kmer.h — k-mer arithmetic shared by the HIP kernels and the host code (SPEC.md S3).
// A k-mer is the unsigned 2k-bit integer whose top 2 bits are its first base, held in W
// 64-bit words, w[0] least significant.  A=0 C=1 G=2 T=3.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SHK_HD __host__ __device__ __forceinline__
#else
#define SHK_HD inline
#endif

namespace shk {

template <int W> struct Kmer { uint64_t w[W]; };

template <int W> SHK_HD Kmer<W> km_zero() {
    Kmer<W> z;
#pragma unroll
    for (int i = 0; i < W; i++) z.w[i] = 0;
    return z;
}

template <int W> SHK_HD bool km_less(const Kmer<W> &a, const Kmer<W> &b) {
#pragma unroll
    for (int i = W - 1; i >= 0; i--) {
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
    }
    return false;
}

template <int W> SHK_HD bool km_eq(const Kmer<W> &a, const Kmer<W> &b) {
    bool e = true;
#pragma unroll
    for (int i = 0; i < W; i++) e = e && (a.w[i] == b.w[i]);
    return e;
}

// mask of the used bits of the top word
template <int W> SHK_HD uint64_t km_topmask(int k) {
    int used = 2 * k - 64 * (W - 1);            // 1..64
    return used >= 64 ? ~0ull : ((1ull << used) - 1ull);
}

// forward roll: drop the first base, append b at the end
template <int W> SHK_HD void km_push_back(Kmer<W> &x, uint32_t b, int k) {
#pragma unroll
    for (int i = W - 1; i > 0; i--) x.w[i] = (x.w[i] << 2) | (x.w[i - 1] >> 62);
    x.w[0] = (x.w[0] << 2) | (uint64_t)b;
    x.w[W - 1] &= km_topmask<W>(k);
}

// reverse roll: drop the last base, prepend b at the front
template <int W> SHK_HD void km_push_front(Kmer<W> &x, uint32_t b, int k) {
#pragma unroll
    for (int i = 0; i < W - 1; i++) x.w[i] = (x.w[i] >> 2) | (x.w[i + 1] << 62);
    x.w[W - 1] >>= 2;
    // W = ceil(2k/64) => the first base always lives in the top word (no runtime word index:
    // a runtime-indexed register array would be demoted to scratch)
    x.w[W - 1] |= (uint64_t)b << ((2 * (k - 1)) & 63);
}

// the 2-bit group at bit offset `bit` (select chain instead of a runtime array index)
template <int W> SHK_HD uint32_t km_bits2(const Kmer<W> &x, int bit) {
    uint64_t word = x.w[0];
#pragma unroll
    for (int i = 1; i < W; i++) word = ((bit >> 6) == i) ? x.w[i] : word;
    return (uint32_t)(word >> (bit & 63)) & 3u;
}

template <int W> SHK_HD uint32_t km_first_base(const Kmer<W> &x, int k) {
    return (uint32_t)(x.w[W - 1] >> ((2 * (k - 1)) & 63)) & 3u;
}
template <int W> SHK_HD uint32_t km_last_base(const Kmer<W> &x) { return (uint32_t)x.w[0] & 3u; }

// reverse the order of the 32 2-bit groups of a word and complement them
SHK_HD uint64_t rc64(uint64_t v) {
    v = ((v >> 2) & 0x3333333333333333ull) | ((v & 0x3333333333333333ull) << 2);
    v = ((v >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((v & 0x0F0F0F0F0F0F0F0Full) << 4);
    v = ((v >> 8) & 0x00FF00FF00FF00FFull) | ((v & 0x00FF00FF00FF00FFull) << 8);
    v = ((v >> 16) & 0x0000FFFF0000FFFFull) | ((v & 0x0000FFFF0000FFFFull) << 16);
    v = (v >> 32) | (v << 32);
    return ~v;
}

template <int W> SHK_HD Kmer<W> km_revcomp(const Kmer<W> &x, int k) {
    // full 64W-bit reverse-complement, then shift right by the unused (64W - 2k) bits
    Kmer<W> t;
#pragma unroll
    for (int i = 0; i < W; i++) t.w[i] = rc64(x.w[W - 1 - i]);
    int sh = 64 * W - 2 * k;                    // 2..62 (k odd => never 0 or 64)
    Kmer<W> r;
#pragma unroll
    for (int i = 0; i < W; i++) {
        uint64_t lo = t.w[i] >> sh;
        uint64_t hi = (i + 1 < W) ? (t.w[i + 1] << (64 - sh)) : 0ull;
        r.w[i] = lo | hi;
    }
    return r;
}

template <int W> SHK_HD Kmer<W> km_canonical(const Kmer<W> &x, int k, int &orient) {
    Kmer<W> r = km_revcomp<W>(x, k);
    if (km_less<W>(r, x)) { orient = 1; return r; }
    orient = 0;
    return x;
}

// placement hash for tables keyed by a full k-mer value (murmur3 fmix64 chain)
SHK_HD uint64_t fmix64(uint64_t h) {
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}
template <int W> SHK_HD uint64_t km_hash(const Kmer<W> &x) {
    uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
    for (int i = 0; i < W; i++) h = fmix64(h ^ x.w[i]);
    return h;
}

// ---- ntHash (Mohamadi et al. 2016), SPEC S3 -------------------------------------------------
SHK_HD uint64_t rol64(uint64_t v, unsigned s) { s &= 63; return s ? (v << s) | (v >> (64 - s)) : v; }
SHK_HD uint64_t ror64(uint64_t v, unsigned s) { s &= 63; return s ? (v >> s) | (v << (64 - s)) : v; }

#define SHK_NT_A 0x3c8bfbb395c60474ull
#define SHK_NT_C 0x3193c18562a02b4cull
#define SHK_NT_G 0x20323ed082572324ull
#define SHK_NT_T 0x295549f54be24456ull

SHK_HD uint64_t nt_seed(uint32_t b) {
    uint64_t lo = (b & 1) ? SHK_NT_C : SHK_NT_A;
    uint64_t hi = (b & 1) ? SHK_NT_T : SHK_NT_G;
    return (b & 2) ? hi : lo;
}

struct NtState { uint64_t fh, rh; };

// feed base number i (0-based) of a window being filled (i < m)
SHK_HD void nt_init_step(NtState &s, uint32_t b, unsigned i) {
    s.fh = rol64(s.fh, 1) ^ nt_seed(b);
    s.rh ^= rol64(nt_seed(3 - b), i);
}
// roll a full m-window: `out` leaves at the front, `in` enters at the back
SHK_HD void nt_roll(NtState &s, uint32_t out, uint32_t in, unsigned m) {
    s.fh = rol64(s.fh, 1) ^ rol64(nt_seed(out), m) ^ nt_seed(in);
    s.rh = ror64(s.rh, 1) ^ ror64(nt_seed(3 - out), 1) ^ rol64(nt_seed(3 - in), m - 1);
}
SHK_HD uint64_t nt_canonical(const NtState &s) { return s.fh < s.rh ? s.fh : s.rh; }

// ---- ntHash with a 32-bit state (minimiser ordering in pass 1 of the counting step) ----------
// The same construction (Mohamadi et al. 2016) over 32-bit words: seeds = the high halves of the
// published 64-bit seeds, rotations are rol32/ror32.  CDNA's VALU is 32 bits wide: a 64-bit
// rotate-xor step costs 4-5 instructions per strand, the 32-bit one 2 (v_alignbit + v_xor).
// For m > 32 bases 32 positions apart share a rotation (as positions 64 apart do in the 64-bit
// form): still a strand-symmetric hash of the m-mer, which is all a minimiser order needs.
SHK_HD uint32_t rol32(uint32_t v, unsigned s) { s &= 31; return s ? (v << s) | (v >> (32 - s)) : v; }
SHK_HD uint32_t ror32(uint32_t v, unsigned s) { s &= 31; return s ? (v >> s) | (v << (32 - s)) : v; }
SHK_HD uint32_t nt32_seed(uint32_t b) { return (uint32_t)(nt_seed(b) >> 32); }
struct Nt32State { uint32_t fh, rh; };
SHK_HD void nt32_init_step(Nt32State &s, uint32_t b, unsigned i) {
    s.fh = rol32(s.fh, 1) ^ nt32_seed(b);
    s.rh ^= rol32(nt32_seed(3 - b), i);
}
SHK_HD void nt32_roll(Nt32State &s, uint32_t out, uint32_t in, unsigned m) {
    s.fh = rol32(s.fh, 1) ^ rol32(nt32_seed(out), m) ^ nt32_seed(in);
    s.rh = ror32(s.rh, 1) ^ ror32(nt32_seed(3 - out), 1) ^ rol32(nt32_seed(3 - in), m - 1);
}
SHK_HD uint32_t nt32_canonical(const Nt32State &s) { return s.fh < s.rh ? s.fh : s.rh; }

}  // namespace shk
