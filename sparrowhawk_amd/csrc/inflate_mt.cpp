// inflate_mt.cpp — a single gzip member inflated by many host threads.
//
// The reference's real input is a `.fastq.gz` pair read through flate2's MultiGzDecoder
// (/root/reference/rust/orphos-bridge/src/fastx_wasm.rs:53-70; /root/reference/docs/src/assembly.md:28).  A plain gzip
// member is one deflate stream: serial by construction for zlib (~0.4 GB/s), 2.6 s in front of a 3.4 ms assembly for the
// bench isolate.  This is the two-pass scheme of pugz / rapidgzip (Kerbiriou & Chikhi 2019; Knespel & Brunst 2023),
// restated from the papers' description:
//   1. the compressed stream is cut into one chunk per thread; every thread but the first SEARCHES, from its cut, for
//      the next bit position where a dynamic-Huffman block really starts (BFINAL = 0, BTYPE = 2, a complete code-length
//      code, complete literal/length and distance codes, an end-of-block symbol, and — the input is FASTQ text —
//      a block whose literals are all text bytes);
//   2. every thread inflates its chunk from there to the next chunk's start WITHOUT knowing the 32 KiB window in front
//      of it: output symbols are 16 bits wide, a back-reference that reaches into the unknown window is kept as a
//      MARKER (256 + position in that window), and markers are copied around like literals;
//   3. the windows are resolved front to back (32 KiB per chunk: the only serial part), then every chunk replaces its
//      markers and narrows to bytes in parallel; CRC-32 per chunk, combined, checked against the member's trailer.
// Anything unexpected — no block start found, a chunk that runs past its neighbour's start, a CRC mismatch — makes the
// caller fall back to zlib: the bytes handed on are always the bytes zlib would produce.
#include "inflate_mt.h"

#include <string.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>

namespace shk {
namespace {

using SymVec = std::vector<uint16_t, NoInitAlloc<uint16_t>>;
constexpr uint32_t WSIZE = 32768;
constexpr uint16_t MARK = 256;                 // symbol >= MARK: byte (symbol - MARK) of the unknown window

struct Bits {
    const uint8_t *in; size_t n;               // the whole member's deflate data
    size_t byte = 0; uint64_t buf = 0; unsigned cnt = 0;
    void seek(uint64_t bitpos) { byte = (size_t)(bitpos >> 3); buf = 0; cnt = 0; refill(); drop((unsigned)(bitpos & 7)); }
    void refill() { while (cnt <= 56 && byte < n) { buf |= (uint64_t)in[byte++] << cnt; cnt += 8; } }
    uint32_t peek(unsigned b) const { return (uint32_t)(buf & ((1ull << b) - 1ull)); }
    void drop(unsigned b) { buf >>= b; cnt -= b; }
    uint32_t get(unsigned b) { if (cnt < b) refill(); const uint32_t v = peek(b); drop(b); return v; }
    uint64_t pos() const { return (uint64_t)byte * 8 - cnt; }
    bool past_end() const { return byte >= n && cnt == 0; }
    bool overrun() const { return cnt > 64; }  // (drop below zero wrapped)
};

// canonical Huffman decoder: one table of 2^PB entries (symbol << 4 | length) for the short codes, a linear walk over
// the lengths above PB (rare in FASTQ streams)
template <unsigned PB> struct Huff {
    uint16_t tab[1u << PB];
    uint16_t count[16], symbol[288];
    unsigned maxlen = 0;
    // returns false unless the lengths form a complete prefix code (or, allow_single, exactly one code of length 1)
    bool build(const uint8_t *len, unsigned n, bool allow_single) {
        memset(count, 0, sizeof count);
        for (unsigned i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        unsigned used = 0; maxlen = 0;
        for (unsigned l = 1; l < 16; l++) { used += count[l]; if (count[l]) maxlen = l; }
        if (!used) return false;
        int left = 1;
        for (unsigned l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; }
        if (left > 0 && !(allow_single && used == 1 && count[1] == 1)) return false;
        uint16_t offs[16]; offs[1] = 0;
        for (unsigned l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
        for (unsigned i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        // table of the codes up to PB bits (deflate codes are packed LSB first: reversed bit order)
        memset(tab, 0, sizeof tab);
        unsigned code = 0, idx = 0;
        for (unsigned l = 1; l <= maxlen; l++) {
            for (unsigned c = 0; c < count[l]; c++, idx++, code++) {
                if (l > PB) continue;
                unsigned rev = 0;
                for (unsigned b = 0; b < l; b++) rev |= ((code >> b) & 1u) << (l - 1 - b);
                for (unsigned f = rev; f < (1u << PB); f += 1u << l) tab[f] = (uint16_t)((symbol[idx] << 4) | l);
            }
            code <<= 1;
        }
        return true;
    }
    // -1: no such code (corrupt, or a false block start)
    int decode(Bits &b) const {
        if (b.cnt < 15) b.refill();
        const uint16_t e = tab[b.peek(PB)];
        if (e & 15u) { b.drop(e & 15u); return e >> 4; }
        // long code: canonical walk bit by bit
        unsigned code = 0, first = 0, index = 0;
        uint64_t v = b.buf;
        for (unsigned l = 1; l <= maxlen; l++) {
            code |= (unsigned)(v & 1u); v >>= 1;
            const unsigned c = count[l];
            if (code < first + c) { if (l > b.cnt) return -1; b.drop(l); return symbol[index + (code - first)]; }
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Codes { Huff<11> lit; Huff<9> dist; };

// reads a dynamic block's code definitions; strict = the checks of a block-start CANDIDATE (complete codes everywhere)
bool read_dynamic(Bits &b, Codes &c) {
    const unsigned hlit = b.get(5) + 257, hdist = b.get(5) + 1, hclen = b.get(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (unsigned i = 0; i < hclen; i++) cl[CL_ORDER[i]] = (uint8_t)b.get(3);
    Huff<7> clh;
    if (!clh.build(cl, 19, true)) return false;
    uint8_t len[320];
    unsigned i = 0;
    while (i < hlit + hdist) {
        const int s = clh.decode(b);
        if (s < 0 || b.overrun()) return false;
        if (s < 16) { len[i++] = (uint8_t)s; continue; }
        unsigned rep, val = 0;
        if (s == 16) { if (i == 0) return false; val = len[i - 1]; rep = 3 + b.get(2); }
        else if (s == 17) rep = 3 + b.get(3);
        else rep = 11 + b.get(7);
        if (i + rep > hlit + hdist) return false;
        while (rep--) len[i++] = (uint8_t)val;
    }
    if (len[256] == 0) return false;                                // no end-of-block code
    if (!c.lit.build(len, hlit, false)) return false;
    if (!c.dist.build(len + hlit, hdist, true)) {
        // (a block without any distance code is legal: all its lengths are zero)
        bool none = true; for (unsigned d = 0; d < hdist; d++) none = none && len[hlit + d] == 0;
        if (!none) return false;
        c.dist.maxlen = 0; memset(c.dist.tab, 0, sizeof c.dist.tab); memset(c.dist.count, 0, sizeof c.dist.count);
    }
    return true;
}
void fixed_codes(Codes &c) {
    uint8_t len[288];
    for (int i = 0; i < 144; i++) len[i] = 8;
    for (int i = 144; i < 256; i++) len[i] = 9;
    for (int i = 256; i < 280; i++) len[i] = 7;
    for (int i = 280; i < 288; i++) len[i] = 8;
    c.lit.build(len, 288, false);
    uint8_t dl[30]; for (int i = 0; i < 30; i++) dl[i] = 5;
    c.dist.build(dl, 30, true);                                      // (30 codes of 5 bits: incomplete by design — built by hand below)
    // the fixed distance code is incomplete (30 of 32): fill the table directly
    memset(c.dist.tab, 0, sizeof c.dist.tab);
    for (unsigned s = 0; s < 30; s++) {
        unsigned rev = 0; for (unsigned bb = 0; bb < 5; bb++) rev |= ((s >> bb) & 1u) << (4 - bb);
        for (unsigned f = rev; f < (1u << 9); f += 32) c.dist.tab[f] = (uint16_t)((s << 4) | 5);
    }
    c.dist.maxlen = 5;
}

static inline bool text_byte(unsigned c) { return (c >= 0x20 && c < 0x7F) || c == '\n' || c == '\r' || c == '\t'; }

// Inflates blocks from the reader's position.  known_window: the output starts the member (references before it are
// errors); otherwise they become markers.  Stops at the end of the final block (final = true) or, after a block, when
// the position has reached stop_at (exactly: ok; beyond it without hitting it: overshoot).  probe: stop after the first
// block and require text literals (block-start candidates).
enum class Stop { Final, AtStop, Overshoot, Corrupt };
Stop inflate_blocks(Bits &b, SymVec &out, bool known_window, uint64_t stop_at, bool probe, uint64_t *end_pos) {
    Codes codes;
    for (;;) {
        if (!probe && b.pos() == stop_at) { *end_pos = b.pos(); return Stop::AtStop; }
        if (!probe && b.pos() > stop_at) return Stop::Overshoot;
        const unsigned bfinal = b.get(1), btype = b.get(2);
        if (b.overrun()) return Stop::Corrupt;
        if (btype == 3) return Stop::Corrupt;
        if (btype == 0) {
            b.drop(b.cnt & 7u);                                      // to the byte boundary
            const unsigned len = b.get(16), nlen = b.get(16);
            if ((len ^ nlen) != 0xFFFFu || b.overrun()) return Stop::Corrupt;
            for (unsigned i = 0; i < len; i++) { if (b.past_end()) return Stop::Corrupt; out.push_back((uint16_t)b.get(8)); }
        } else {
            if (btype == 1) fixed_codes(codes);
            else if (!read_dynamic(b, codes)) return Stop::Corrupt;
            for (;;) {
                const int s = codes.lit.decode(b);
                if (s < 0 || b.overrun()) return Stop::Corrupt;
                if (s < 256) { if (probe && !text_byte((unsigned)s)) return Stop::Corrupt; out.push_back((uint16_t)s); continue; }
                if (s == 256) break;
                if (s > 285) return Stop::Corrupt;
                const unsigned len = LEN_BASE[s - 257] + b.get(LEN_EXTRA[s - 257]);
                const int ds = codes.dist.decode(b);
                if (ds < 0 || ds > 29 || b.overrun()) return Stop::Corrupt;
                const unsigned dist = DIST_BASE[ds] + b.get(DIST_EXTRA[ds]);
                const size_t pos = out.size();
                if (dist > pos && known_window) return Stop::Corrupt;
                if (dist > pos + WSIZE) return Stop::Corrupt;
                out.resize(pos + len);
                uint16_t *o = out.data();
                for (unsigned i = 0; i < len; i++) {
                    const size_t p = pos + i;
                    o[p] = p >= dist ? o[p - dist] : (uint16_t)(MARK + (WSIZE - (dist - p)));
                }
            }
        }
        if (probe) { *end_pos = b.pos(); return bfinal ? Stop::Final : Stop::AtStop; }
        if (bfinal) { *end_pos = b.pos(); return Stop::Final; }
    }
}

// the first bit position >= from (< limit) where a non-final dynamic block of text starts, followed by a sane block header
bool find_block_start(const uint8_t *in, size_t n, uint64_t from, uint64_t limit, uint64_t &found) {
    SymVec scratch;
    Bits b{in, n};
    for (uint64_t p = from; p < limit; p++) {
        // cheap filters first: BFINAL = 0, BTYPE = 2 (bits: 0, then 0 1 LSB first -> value 0b100 = 4 over three bits)
        const size_t byte = (size_t)(p >> 3);
        if (byte + 8 >= n) return false;
        uint32_t w; memcpy(&w, in + byte, 4);
        const uint32_t h = w >> (p & 7);
        if ((h & 7u) != 4u) continue;
        if (((h >> 3) & 31u) > 29u || ((h >> 8) & 31u) > 29u) continue;      // HLIT <= 286, HDIST <= 30
        b.seek(p);
        scratch.clear();
        uint64_t end = 0;
        if (inflate_blocks(b, scratch, false, 0, true, &end) != Stop::AtStop) continue;
        if (scratch.size() < 64) continue;                                   // (a real block of a FASTQ stream holds thousands of symbols)
        // the next header must make sense too
        const unsigned nb = b.get(3);
        if ((nb >> 1) == 3 || b.overrun()) continue;
        found = p;
        return true;
    }
    return false;
}

std::atomic<uint64_t> g_members{0};

// fn(t) for t in [0, T) on T threads.  No exception leaves a worker (std::terminate) or this function with threads still
// joinable: whatever is thrown — std::bad_alloc from a marker buffer on a host short of memory, a gzip bomb — is caught,
// every thread is joined, and false comes back: the caller gives the member to zlib.
template <typename F> bool run_threads(unsigned T, F &&fn) {
    std::atomic<bool> failed{false};
    auto safe = [&fn, &failed](unsigned t) { try { fn(t); } catch (...) { failed.store(true); } };
    std::vector<std::thread> ts;
    try {
        ts.reserve(T);
        for (unsigned t = 1; t < T; t++) ts.emplace_back(safe, t);
    } catch (...) {                                           // (a thread could not be started: its share stays undone)
        failed.store(true);
    }
    safe(0u);
    for (auto &t : ts) t.join();
    return !failed.load();
}

}  // namespace

static int inflate_member_parallel_body(const uint8_t *in, size_t n, ByteVec &out, size_t out_at, size_t &consumed, unsigned threads);
// 0 = the member is in `out`; 1 = not taken (the caller gives it to zlib: every unexpected turn ends here, out of host
// memory included — no exception leaves this function)
int inflate_member_parallel(const uint8_t *in, size_t n, ByteVec &out, size_t out_at, size_t &consumed, unsigned threads) {
    try { return inflate_member_parallel_body(in, n, out, out_at, consumed, threads); }
    catch (...) {
        try { if (out.size() > out_at) out.resize(out_at); } catch (...) {}
        return 1;
    }
}
static int inflate_member_parallel_body(const uint8_t *in, size_t n, ByteVec &out, size_t out_at, size_t &consumed, unsigned threads) {
    // ---- gzip member header (RFC 1952)
    if (n < 18 || in[0] != 0x1F || in[1] != 0x8B || in[2] != 8) return 1;
    const unsigned flg = in[3];
    size_t p = 10;
    if (flg & 4) { if (p + 2 > n) return 1; p += 2 + (in[p] | ((size_t)in[p + 1] << 8)); }
    if (flg & 8) { while (p < n && in[p]) p++; p++; }
    if (flg & 16) { while (p < n && in[p]) p++; p++; }
    if (flg & 2) p += 2;
    if (p + 8 >= n) return 1;
    const uint8_t *def = in + p;
    const size_t dn = n - p;                                   // deflate data (+ trailer, + whatever follows the member)
    if (threads < 2 || dn < ((size_t)1 << 20)) return 1;
    unsigned C = (unsigned)std::min<size_t>(threads, dn / ((size_t)512 << 10));     // >= 512 KiB of compressed data per chunk
    if (C < 2) return 1;
    const bool dbg = getenv("SHK_GUNZIP_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // ---- 1. block starts
    std::vector<uint64_t> start(C, 0);
    std::vector<uint8_t> ok(C, 1);
    if (!run_threads(C, [&](unsigned c) {
        if (c == 0) return;
        const uint64_t from = (uint64_t)(dn / C) * c * 8, limit = std::min<uint64_t>((uint64_t)(dn / C) * (c + 1) * 8, (uint64_t)dn * 8);
        uint64_t f = 0;
        if (find_block_start(def, dn, from, limit, f)) start[c] = f; else ok[c] = 0;
    })) return 1;
    // (a chunk without a start is merged into its predecessor)
    std::vector<uint64_t> st; st.push_back(0);
    for (unsigned c = 1; c < C; c++) if (ok[c] && start[c] > st.back()) st.push_back(start[c]);
    C = (unsigned)st.size();
    if (C < 2) return 1;
    const double t1 = now();
    // ---- 2. every chunk, with markers for what lies in front of it
    std::vector<SymVec> sym(C);
    std::vector<Stop> how(C, Stop::Corrupt);
    std::vector<uint64_t> endpos(C, 0);
    if (!run_threads(C, [&](unsigned c) {
        Bits b{def, dn};
        b.seek(st[c]);
        const uint64_t stop_at = c + 1 < C ? st[c + 1] : ~0ull;
        sym[c].reserve((size_t)((c + 1 < C ? st[c + 1] : (uint64_t)dn * 8) - st[c]) / 8 * 6 + 65536);   // (FASTQ text deflates 3-5x; a vector that runs out grows)
        how[c] = inflate_blocks(b, sym[c], c == 0, stop_at, false, &endpos[c]);
    })) return 1;
    for (unsigned c = 0; c < C; c++) if (how[c] != (c + 1 < C ? Stop::AtStop : Stop::Final)) return 1;
    const double t2 = now();
    // ---- trailer
    const size_t tail = (size_t)((endpos[C - 1] + 7) >> 3);
    if (tail + 8 > dn) return 1;
    const uint32_t want_crc = def[tail] | ((uint32_t)def[tail + 1] << 8) | ((uint32_t)def[tail + 2] << 16) | ((uint32_t)def[tail + 3] << 24);
    const uint32_t want_len = def[tail + 4] | ((uint32_t)def[tail + 5] << 8) | ((uint32_t)def[tail + 6] << 16) | ((uint32_t)def[tail + 7] << 24);
    std::vector<size_t> off(C + 1, 0);
    for (unsigned c = 0; c < C; c++) off[c + 1] = off[c] + sym[c].size();
    if ((uint32_t)off[C] != want_len) return 1;
    // ---- 3. windows front to back (the serial part: 32 KiB per chunk), then markers -> bytes in parallel
    std::vector<std::vector<uint8_t>> win(C);                   // win[c]: the 32 KiB in front of chunk c (c >= 1)
    for (unsigned c = 1; c < C; c++) {
        win[c].assign(WSIZE, 0);
        const SymVec &s = sym[c - 1];
        const size_t have = std::min<size_t>(s.size(), WSIZE);
        // what the previous chunk does not cover comes from ITS window
        if (have < WSIZE && c >= 2) memcpy(win[c].data(), win[c - 1].data() + have, WSIZE - have);
        for (size_t i = 0; i < have; i++) {
            const uint16_t v = s[s.size() - have + i];
            if (v >= MARK) { if (c < 2) return 1; win[c][WSIZE - have + i] = win[c - 1][v - MARK]; }
            else win[c][WSIZE - have + i] = (uint8_t)v;
        }
    }
    const double t3 = now();
    out.resize(out_at + off[C]);
    const double t4 = now();
    std::vector<uint32_t> crc(C, 0);
    std::atomic<int> bad{0};
    if (!run_threads(C, [&](unsigned c) {
        uint8_t *o = out.data() + out_at + off[c];
        const SymVec &s = sym[c];
        const uint8_t *w = c ? win[c].data() : nullptr;
        for (size_t i = 0; i < s.size(); i++) {
            const uint16_t v = s[i];
            if (v >= MARK) { if (!w) { bad = 1; return; } o[i] = w[v - MARK]; } else o[i] = (uint8_t)v;
        }
        uint32_t cr = (uint32_t)crc32(0L, Z_NULL, 0);
        for (size_t a = 0; a < s.size(); a += (size_t)1 << 30) cr = (uint32_t)crc32(cr, o + a, (uInt)std::min<size_t>(s.size() - a, (size_t)1 << 30));
        crc[c] = cr;
        SymVec().swap(sym[c]);
    })) { out.resize(out_at); return 1; }
    if (bad.load()) { out.resize(out_at); return 1; }
    uint32_t total_crc = crc[0];
    for (unsigned c = 1; c < C; c++) total_crc = (uint32_t)crc32_combine(total_crc, crc[c], (z_off_t)(off[c + 1] - off[c]));
    if (total_crc != want_crc) { out.resize(out_at); return 1; }
    consumed = p + tail + 8;
    if (dbg) fprintf(stderr, "[inflate_mt] chunks %u: search %.3f s, decode %.3f s, windows %.3f s, resize %.3f s, resolve+crc %.3f s\n", C, t1 - t0, t2 - t1, t3 - t2, t4 - t3, now() - t4);
    g_members.fetch_add(1);
    return 0;
}
uint64_t inflate_mt_members() { return g_members.load(); }

}  // namespace shk
