// inflate_gpu.hip — see inflate_gpu.h.  One gzip member (one deflate stream) inflated on gfx950.
//
// The reference's real input is a `.fastq.gz` pair read through flate2's MultiGzDecoder
// (/root/reference/rust/orphos-bridge/src/fastx_wasm.rs:53-70; /root/reference/docs/src/assembly.md:25-28).  A deflate
// stream is serial by construction; the two-pass scheme of pugz / rapidgzip (restated for host threads in inflate_mt.cpp)
// makes it parallel, and thousands of chunks make it a GPU job:
//   k_gz_find_starts   one wave per cut of the compressed stream: 64 bit positions are tested at once (one per lane) for
//                      "a non-final dynamic-Huffman block starts here" — header fields, a complete code-length code, complete
//                      literal/length and distance codes with an end-of-block symbol, all from registers and a 128-byte
//                      per-lane table in LDS; a position that passes is probed by the whole wave (real tables, a few
//                      thousand symbols of the block: text bytes only, distances in range)
//   k_gz_decode        one wave per chunk, from its start to the next chunk's start, WITHOUT the 32 KiB window in front
//                      of it: 16-bit symbols, a back-reference into the unknown window is a MARKER (256 + position in that
//                      window).  The Huffman state is wave-uniform (every lane decodes the same symbol: tables in LDS,
//                      broadcast reads), the copy of a match is done by the 64 lanes together out of an LDS ring that holds
//                      the last 4096 symbols (older ones from HBM); output leaves in coalesced 1 KB flushes
//   k_gz_windows       the 32 KiB windows in front of the chunks, front to back (the one serial step: one workgroup, the
//                      window of chunk c from that of c - 1 through LDS)
//   k_gz_resolve       markers -> bytes, 16 -> 8 bits, coalesced;  k_gz_crc  CRC-32 of 64 KiB slices (combined on the host)
// Everything that does not look as expected makes gpu_inflate_member return 1 and the caller inflates on the host: the
// bytes handed on are always the bytes zlib would produce (CRC-32 and ISIZE of the trailer are checked here too).
#include <hip/hip_runtime.h>
#include <zlib.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "inflate_gpu.h"
#include "pipeline.h"

namespace shk {
namespace {

constexpr uint32_t GZ_WSIZE = 32768;
constexpr uint32_t GZ_MARK = 256;
constexpr int LIT_PB = 10, DIST_PB = 9, CL_PB = 7;
constexpr uint32_t RING = 2048, RING_MASK = RING - 1, NEAR = RING - 768, FLUSH = 256;
constexpr uint32_t PROBE_SYMS = 1024;

// ---- bit reader over the deflate data (32-bit words, zero-padded behind the end) ------------------------------------
struct DBits {
    const uint32_t *w; uint64_t nbytes;
    uint64_t byte; uint64_t buf; uint32_t cnt;
    __device__ __forceinline__ uint32_t load32(uint64_t at) const {
        const uint64_t i = at >> 2;
        const uint32_t a = w[i], b = w[i + 1], sh = (uint32_t)(at & 3u) * 8u;
        return sh ? (a >> sh) | (b << (32u - sh)) : a;
    }
    __device__ __forceinline__ void refill() { if (cnt <= 32u) { buf |= (uint64_t)load32(byte) << cnt; byte += 4; cnt += 32u; } }
    __device__ __forceinline__ void seek(uint64_t bitpos) { byte = bitpos >> 3; buf = 0; cnt = 0; refill(); refill(); drop((uint32_t)(bitpos & 7u)); }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    __device__ __forceinline__ void drop(uint32_t n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t get(uint32_t n) { refill(); const uint32_t v = peek(n); drop(n); return v; }     // n <= 32
    __device__ __forceinline__ uint64_t pos() const { return byte * 8ull - cnt; }
    __device__ __forceinline__ bool overrun() const { return pos() > nbytes * 8ull; }
};

// The same stream read by a WAVE whose lanes all decode the same symbol (k_gz_decode, the probe): the next 64 dwords of the
// input sit one per lane in a register (fetched with one coalesced load, the 64 behind them already on their way), and
// a refill is a v_readlane — no memory latency on the symbol-to-symbol chain.
struct UBits {
    const uint32_t *w; uint64_t nbytes;
    uint32_t wbase, di;                       // dword index of the window's first word / of the next word to consume (input < 16 GiB)
    uint32_t lanew, lanew_next;
    uint64_t buf; uint32_t cnt;
    __device__ __forceinline__ uint32_t next_word() {
        uint32_t k = di - wbase;
        if (k >= 64u) { lanew = lanew_next; wbase += 64u; lanew_next = w[(size_t)wbase + 64u + threadIdx.x]; k -= 64u; }
        di++;
        return (uint32_t)__builtin_amdgcn_readlane((int)lanew, __builtin_amdgcn_readfirstlane((int)k));
    }
    __device__ __forceinline__ void refill() { if (cnt <= 32u) { buf |= (uint64_t)next_word() << cnt; cnt += 32u; } }
    __device__ __forceinline__ void seek(uint64_t bitpos) {
        wbase = (uint32_t)(bitpos >> 5); di = wbase; buf = 0; cnt = 0;
        lanew = w[(size_t)wbase + threadIdx.x]; lanew_next = w[(size_t)wbase + 64u + threadIdx.x];
        refill(); refill(); drop((uint32_t)(bitpos & 31u));
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }              // n < 32
    __device__ __forceinline__ void drop(uint32_t n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t get(uint32_t n) { refill(); const uint32_t v = peek(n); drop(n); return v; }     // n < 32
    __device__ __forceinline__ uint64_t pos() const { return (uint64_t)di * 32ull - cnt; }
    __device__ __forceinline__ bool overrun() const { return pos() > nbytes * 8ull; }
};
// length / distance codes -> base value and extra bits, by arithmetic (RFC 1951 3.2.5; a table in constant memory costs a
// scalar load — hundreds of cycles — per lookup on the symbol-to-symbol chain)
__device__ __forceinline__ void len_code(uint32_t c, uint32_t &base, uint32_t &extra) {       // c = symbol - 257, 0 .. 28
    if (c < 8u) { base = 3u + c; extra = 0; }
    else if (c == 28u) { base = 258u; extra = 0; }
    else { extra = (c >> 2) - 1u; base = 3u + ((4u + (c & 3u)) << extra); }
}
__device__ __forceinline__ void dist_code(uint32_t d, uint32_t &base, uint32_t &extra) {      // d = 0 .. 29
    if (d < 4u) { base = 1u + d; extra = 0; }
    else { extra = (d >> 1) - 1u; base = 1u + ((2u + (d & 1u)) << extra); }
}
// the order in which a dynamic block lists the lengths of its code-length code (16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15), 5 bits each
__device__ __forceinline__ uint32_t cl_order(uint32_t i) {
    const unsigned long long lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
    const unsigned long long hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
    return i < 12u ? (uint32_t)(lo >> (5u * i)) & 31u : (uint32_t)(hi >> (5u * (i - 12u))) & 31u;
}

// ---- canonical Huffman tables of one wave in LDS ---------------------------------------------------------------------
struct HuffLds {
    uint16_t lit_tab[1 << LIT_PB];           // symbol << 4 | code length; 0: a code longer than the table's index
    uint16_t dist_tab[1 << DIST_PB];
    uint16_t cl_tab[1 << CL_PB];
    uint16_t lit_sym[288], dist_sym[32], cl_sym[20];
    uint32_t lit_count[16], dist_count[16], cl_count[16];
    uint32_t lit_maxlen, dist_maxlen, cl_maxlen;
    uint8_t len[328];                        // code lengths of the block being set up
};
struct WaveLds {
    HuffLds h;
    uint16_t ring[RING];                     // the last RING symbols of the chunk's output
};

// Builds the decoding table of one code from len[0..n) (wave-cooperative; one wave per workgroup).  Returns false unless
// the lengths form a complete prefix code (or, allow_single, exactly one code of length 1).
__device__ bool build_code(const uint8_t *len, uint32_t n, bool allow_single, uint16_t *tab, int PB, uint32_t *count,
                           uint16_t *symbol, uint32_t *maxlen_out) {
    const int lane = threadIdx.x;
    if (lane < 16) count[lane] = 0;
    for (uint32_t f = lane; f < (1u << PB); f += 64) tab[f] = 0;
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64) { const uint32_t l = len[i]; if (l) atomicAdd(&count[l], 1u); }
    __syncthreads();
    uint32_t used = 0, maxlen = 0; int left = 1; bool over = false;
    for (uint32_t l = 1; l < 16; l++) { const uint32_t cl = count[l]; used += cl; if (cl) maxlen = l; left = left * 2 - (int)cl; if (left < 0) over = true; }
    if (!used || over) return false;
    if (left > 0 && !(allow_single && used == 1 && count[1] == 1)) return false;
    // symbols in canonical order: by length, then by value
    uint32_t off = 0;
    for (uint32_t l = 1; l <= maxlen; l++) {
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t i = base + lane;
            const bool has = i < n && len[i] == l;
            const unsigned long long m = __ballot(has);
            if (has) symbol[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
            off += (uint32_t)__popcll(m);
        }
    }
    __syncthreads();
    // table of the codes up to PB bits (deflate codes are packed LSB first: reversed bit order)
    for (uint32_t idx = lane; idx < used; idx += 64) {
        uint32_t l = 1, start = 0, code = 0;
        for (uint32_t q = 1; q < 15; q++) { const uint32_t cq = count[q]; if (idx < start + cq) break; start += cq; code = (code + cq) << 1; l = q + 1; }
        if (l > (uint32_t)PB) continue;
        const uint32_t cw = code + (idx - start);
        const uint32_t rev = __brev(cw) >> (32u - l);
        const uint16_t e = (uint16_t)((symbol[idx] << 4) | l);
        for (uint32_t f = rev; f < (1u << PB); f += 1u << l) tab[f] = e;
    }
    if (lane == 0) *maxlen_out = maxlen;
    __syncthreads();
    return true;
}

// one symbol (wave-uniform: every lane runs this on the same state); -1: no such code
__device__ __forceinline__ int decode_sym(UBits &b, const uint16_t *tab, int PB, const uint32_t *count, const uint16_t *symbol, uint32_t maxlen) {
    // (every lane reads the same entry: readfirstlane tells the compiler so, and the Huffman state — bit buffer, counters,
    // positions — lives in scalar registers and is worked on by the scalar unit, not 64 times over by the vector unit)
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[b.peek((uint32_t)PB)]);
    if (e & 15u) { b.drop(e & 15u); return (int)(e >> 4); }
    uint32_t code = 0, first = 0, index = 0;
    uint64_t v = b.buf;
    for (uint32_t l = 1; l <= maxlen; l++) {
        code |= (uint32_t)(v & 1u); v >>= 1;
        const uint32_t cc = (uint32_t)__builtin_amdgcn_readfirstlane((int)count[l]);
        if (code < first + cc) { if (l > b.cnt) return -1; b.drop(l); return __builtin_amdgcn_readfirstlane((int)symbol[index + (code - first)]); }
        index += cc; first += cc; first <<= 1; code <<= 1;
    }
    return -1;
}

// a dynamic block's code definitions at the reader's position -> tables (wave-uniform); false: not a valid header
__device__ bool read_dynamic(UBits &b, HuffLds &h) {
    const int lane = threadIdx.x;
    const uint32_t hlit = b.get(5) + 257, hdist = b.get(5) + 1, hclen = b.get(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    if (lane < 19) h.len[lane] = 0;
    __syncthreads();
    for (uint32_t i = 0; i < hclen; i++) { const uint32_t v = b.get(3); if (lane == 0) h.len[cl_order(i)] = (uint8_t)v; }
    __syncthreads();
    if (!build_code(h.len, 19, true, h.cl_tab, CL_PB, h.cl_count, h.cl_sym, &h.cl_maxlen)) return false;
    const uint32_t cl_maxlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)h.cl_maxlen);
    __syncthreads();                                           // (h.len is rewritten below)
    uint32_t i = 0, prev = 0;
    const uint32_t total = hlit + hdist;
    while (i < total) {
        b.refill();
        const int s = decode_sym(b, h.cl_tab, CL_PB, h.cl_count, h.cl_sym, cl_maxlen);
        if (s < 0 || b.overrun()) return false;
        if (s < 16) { if (lane == 0) h.len[i] = (uint8_t)s; prev = (uint32_t)s; i++; continue; }
        uint32_t rep, val = 0;
        if (s == 16) { if (i == 0) return false; val = prev; rep = 3 + b.get(2); }
        else if (s == 17) rep = 3 + b.get(3);
        else rep = 11 + b.get(7);
        if (i + rep > total) return false;
        for (uint32_t q = lane; q < rep; q += 64) h.len[i + q] = (uint8_t)val;
        i += rep; prev = val;
    }
    __syncthreads();
    if (h.len[256] == 0) return false;                          // no end-of-block code
    if (!build_code(h.len, hlit, false, h.lit_tab, LIT_PB, h.lit_count, h.lit_sym, &h.lit_maxlen)) return false;
    if (!build_code(h.len + hlit, hdist, true, h.dist_tab, DIST_PB, h.dist_count, h.dist_sym, &h.dist_maxlen)) {
        // (a block without any distance code is legal: all its lengths are zero)
        bool none = true;
        for (uint32_t d = 0; d < hdist; d++) none = none && h.len[hlit + d] == 0;
        if (!none) return false;
        for (uint32_t f = lane; f < (1u << DIST_PB); f += 64) h.dist_tab[f] = 0;
        if (lane < 16) h.dist_count[lane] = 0;
        if (lane == 0) h.dist_maxlen = 0;
        __syncthreads();
    }
    return true;
}
__device__ void fixed_codes(HuffLds &h) {
    const int lane = threadIdx.x;
    for (uint32_t i = lane; i < 288; i += 64) h.len[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
    __syncthreads();
    (void)build_code(h.len, 288, false, h.lit_tab, LIT_PB, h.lit_count, h.lit_sym, &h.lit_maxlen);
    // the fixed distance code is incomplete (30 of 32 five-bit codes): the table is filled directly
    for (uint32_t f = lane; f < (1u << DIST_PB); f += 64) {
        const uint32_t s = __brev(f & 31u) >> 27;
        h.dist_tab[f] = s < 30 ? (uint16_t)((s << 4) | 5u) : (uint16_t)0;
    }
    if (lane < 16) h.dist_count[lane] = 0;
    if (lane == 0) h.dist_maxlen = 0;
    __syncthreads();
}

__device__ __forceinline__ bool text_byte(uint32_t c) { return (c >= 0x20 && c < 0x7F) || c == '\n' || c == '\r' || c == '\t'; }

// ---- the wave probes ONE candidate position: a real dynamic block of text?  (k_gz_find_starts) ------------------------
struct ProbeLds { HuffLds h; };
__device__ bool probe_block(const uint32_t *w, uint64_t nbytes, uint64_t p, ProbeLds &L) {
    UBits b; b.w = w; b.nbytes = nbytes;
    b.seek(p);
    const uint32_t bfinal = b.get(1), btype = b.get(2);
    if (bfinal != 0 || btype != 2) return false;
    if (!read_dynamic(b, L.h)) return false;
    const uint32_t lit_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.h.lit_maxlen), dist_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.h.dist_maxlen);
    uint32_t produced = 0, nsym = 0;
    for (;;) {
        b.refill();
        const int s = decode_sym(b, L.h.lit_tab, LIT_PB, L.h.lit_count, L.h.lit_sym, lit_max);
        if (s < 0 || b.overrun()) return false;
        nsym++;
        if (s < 256) { if (!text_byte((uint32_t)s)) return false; produced++; }
        else if (s == 256) {
            if (nsym < 64) return false;                        // (a real block of a FASTQ stream holds thousands of symbols)
            const uint32_t nb = b.get(3);                       // the next header must make sense too
            return (nb >> 1) != 3 && !b.overrun();
        } else {
            if (s > 285) return false;
            uint32_t lb, le; len_code((uint32_t)s - 257u, lb, le);
            const uint32_t len = lb + b.get(le);
            b.refill();
            const int ds = decode_sym(b, L.h.dist_tab, DIST_PB, L.h.dist_count, L.h.dist_sym, dist_max);
            if (ds < 0 || ds > 29) return false;
            uint32_t db, de; dist_code((uint32_t)ds, db, de);
            const uint32_t dist = db + b.get(de);
            if (b.overrun() || dist > produced + GZ_WSIZE) return false;
            produced += len;
        }
        if (nsym >= PROBE_SYMS) return true;
    }
}

// per lane: does a dynamic block header at bit position p hold complete codes?  (registers + a 128-byte table of the
// code-length code per lane: cl_tab[entry][lane])
__device__ bool header_plausible(const uint32_t *w, uint64_t nbytes, uint64_t p, uint8_t (*cl_tab)[64]) {
    const int lane = threadIdx.x;
    if ((p >> 3) + 16 >= nbytes) return false;
    // 96 bits from p: the 17 header bits and up to 19 x 3 bits of code-length code lengths
    const uint64_t wi = p >> 5; const uint32_t sh = (uint32_t)(p & 31u);
    const uint32_t d0 = w[wi], d1 = w[wi + 1], d2 = w[wi + 2], d3 = w[wi + 3];
    const uint32_t e0 = sh ? (d0 >> sh) | (d1 << (32u - sh)) : d0, e1 = sh ? (d1 >> sh) | (d2 << (32u - sh)) : d1, e2 = sh ? (d2 >> sh) | (d3 << (32u - sh)) : d2;
    // BFINAL = 0, BTYPE = 2 (bits 0, then 0 1 LSB first: value 4 over three bits), HLIT <= 286, HDIST <= 30
    if ((e0 & 7u) != 4u || ((e0 >> 3) & 31u) > 29u || ((e0 >> 8) & 31u) > 29u) return false;
    const uint32_t hlit = ((e0 >> 3) & 31u) + 257u, hdist = ((e0 >> 8) & 31u) + 1u, hclen = ((e0 >> 13) & 15u) + 4u;
    // the lengths of the code-length code, in the order of the stream: is the code complete?  (the order does not matter for that)
    const uint64_t raw = (((uint64_t)e1 << 32 | e0) >> 17) | ((uint64_t)e2 << 47);
    uint32_t kraft = 0, used = 0;
    for (uint32_t i = 0; i < hclen; i++) { const uint32_t l = (uint32_t)(raw >> (3u * i)) & 7u; if (l) { kraft += 128u >> l; used++; } }
    if (kraft != 128u || used < 2) return false;                // complete (a single code of length 1 is left to the chunk before)
    uint64_t clv = 0;                                           // ... now by symbol: 19 lengths of 3 bits in one register
    for (uint32_t i = 0; i < hclen; i++) clv |= (uint64_t)((uint32_t)(raw >> (3u * i)) & 7u) << (3u * cl_order(i));
    DBits b{w, nbytes, 0, 0, 0};
    b.seek(p + 17u + 3u * hclen);
    // canonical codes -> the lane's 7-bit table (a complete code fills every entry)
    uint64_t next = 0;                                          // next code of length l in byte l
    {
        uint32_t code = 0;
        for (uint32_t l = 1; l <= 7; l++) {
            uint32_t cnt = 0;
            for (uint32_t s = 0; s < 19; s++) cnt += ((uint32_t)(clv >> (3u * s)) & 7u) == l;
            next |= (uint64_t)code << (8u * l);
            code = (code + cnt) << 1;
        }
    }
    for (uint32_t s = 0; s < 19; s++) {
        const uint32_t l = (uint32_t)(clv >> (3u * s)) & 7u;
        if (!l) continue;
        const uint32_t code = (uint32_t)(next >> (8u * l)) & 0xFFu;
        next += 1ull << (8u * l);
        const uint32_t rev = __brev(code) >> (32u - l);
        for (uint32_t f = rev; f < 128u; f += 1u << l) cl_tab[f][lane] = (uint8_t)((s << 3) | l);
    }
    // the literal/length and distance code lengths: only their Kraft sums, the end-of-block code and the counts are kept
    uint32_t i = 0, prev = 0, kl = 0, kd = 0, nd = 0, d_single = 0;
    bool has256 = false;
    const uint32_t total = hlit + hdist;
    auto put = [&](uint32_t at, uint32_t l) {
        if (!l) return;
        if (at < hlit) { kl += 32768u >> l; if (at == 256) has256 = true; }
        else { kd += 32768u >> l; nd++; d_single = l; }
    };
    // (random bits over-subscribe a code within a few dozen lengths: the walk over ~300 lengths ends there)
    while (i < total) {
        b.refill();
        const uint32_t e = cl_tab[b.peek(7)][lane];
        const uint32_t s = e >> 3;
        b.drop(e & 7u);
        if (b.overrun()) return false;
        if (s < 16) { put(i, s); prev = s; i++; if (kl > 32768u || kd > 32768u) return false; continue; }
        uint32_t rep, val = 0;
        if (s == 16) { if (i == 0) return false; val = prev; rep = 3 + b.get(2); }
        else if (s == 17) rep = 3 + b.get(3);
        else rep = 11 + b.get(7);
        if (i + rep > total) return false;
        for (uint32_t q = 0; q < rep; q++) put(i + q, val);
        i += rep; prev = val;
        if (kl > 32768u || kd > 32768u) return false;
    }
    if (!has256 || kl != 32768u) return false;
    if (!(kd == 32768u || nd == 0 || (nd == 1 && d_single == 1))) return false;
    return true;
}

// start[c] (c >= 1): the first bit position >= cut[c] (< cut[c + 1]) where a non-final dynamic block of text starts; ~0: none
__global__ __launch_bounds__(64) void k_gz_find_starts(const uint32_t *__restrict__ w, uint64_t nbytes, uint64_t chunk_bytes, uint32_t n_chunks,
                                                      unsigned long long *__restrict__ start) {
    __shared__ ProbeLds L;
    __shared__ uint8_t cl_tab[128][64];
    const uint32_t c = blockIdx.x + 1;
    if (c >= n_chunks) return;
    const int lane = threadIdx.x;
    const uint64_t from = (uint64_t)c * chunk_bytes * 8ull, limit = min((uint64_t)(c + 1) * chunk_bytes * 8ull, nbytes * 8ull);
    unsigned long long found = ~0ull;
    for (uint64_t base = from; base < limit && found == ~0ull; base += 64) {
        const uint64_t p = base + (uint64_t)lane;
        const bool ok = p < limit && header_plausible(w, nbytes, p, cl_tab);
        unsigned long long m = __ballot(ok);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (probe_block(w, nbytes, base + (uint64_t)l, L)) { found = base + (uint64_t)l; break; }
            __syncthreads();
        }
    }
    if (lane == 0) start[c] = found;
}

// ---- decode one chunk ------------------------------------------------------------------------------------------------
enum : uint32_t { GZ_FINAL = 1, GZ_ATSTOP = 2, GZ_OVERSHOOT = 3, GZ_CORRUPT = 4, GZ_NO_ROOM = 5 };
struct ChunkOut { unsigned long long n_sym, end_pos; uint32_t status, pad; };

__global__ __launch_bounds__(64) void k_gz_decode(const uint32_t *__restrict__ w, uint64_t nbytes, uint32_t n_chunks,
                                                 const unsigned long long *__restrict__ start, const unsigned long long *__restrict__ sym_off,
                                                 uint16_t *__restrict__ syms, ChunkOut *__restrict__ res) {
    __shared__ WaveLds L;
    const uint32_t c = blockIdx.x;
    if (c >= n_chunks) return;
    const uint32_t lane = threadIdx.x;
    const uint64_t stop_at = c + 1 < n_chunks ? start[c + 1] : ~0ull;
    uint16_t *out = syms + sym_off[c];
    const uint64_t room64 = sym_off[c + 1] - sym_off[c];
    const uint32_t room = (uint32_t)(room64 > 0xFFFF0000ull ? 0xFFFF0000ull : room64);
    const bool known_window = c == 0;
    UBits b; b.w = w; b.nbytes = nbytes;
    b.seek(start[c]);
    uint32_t pos = 0, flushed = 0;                               // symbols produced / already in HBM (a multiple of FLUSH)
    uint32_t status = 0;
    // whenever FLUSH symbols have piled up: out they go (coalesced), and the two checks that need not run per symbol are made:
    // room for what may come before the next flush, and a reader that has not left the input (it is zero-padded far enough
    // for what can be read in between)
    auto flush_full = [&]() -> bool {
        while (pos - flushed >= FLUSH) {
            const uint32_t *r32 = reinterpret_cast<const uint32_t *>(L.ring);
            uint32_t *o32 = reinterpret_cast<uint32_t *>(out + flushed);
            const uint32_t at = (flushed & RING_MASK) >> 1;
#pragma unroll
            for (uint32_t t = 0; t < FLUSH / 2 / 64; t++) o32[t * 64 + lane] = r32[at + t * 64 + lane];
            flushed += FLUSH;
            // (symbols this far back are read from HBM again by far matches — by lanes of this same wave: the stores have to be
            // complete, nothing has to leave the L2)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }
        if (pos + 2048u > room) { status = GZ_NO_ROOM; return false; }
        if (b.overrun()) { status = GZ_CORRUPT; return false; }
        return true;
    };
    for (;;) {
        if (b.pos() == stop_at) { status = GZ_ATSTOP; break; }
        if (b.pos() > stop_at) { status = GZ_OVERSHOOT; break; }
        const uint32_t bfinal = b.get(1), btype = b.get(2);
        if (b.overrun() || btype == 3) { status = GZ_CORRUPT; break; }
        if (btype == 0) {
            b.drop(b.cnt & 7u);                                  // to the byte boundary
            const uint32_t len = b.get(16), nlen = b.get(16);
            if ((len ^ nlen) != 0xFFFFu || b.overrun()) { status = GZ_CORRUPT; break; }
            bool ok = true;
            for (uint32_t i = 0; i < len && ok; i++) {
                const uint32_t v = b.get(8);
                L.ring[pos & RING_MASK] = (uint16_t)v;
                pos++;
                if (pos - flushed >= FLUSH) ok = flush_full();
            }
            if (!ok) break;
            if (b.overrun()) { status = GZ_CORRUPT; break; }
        } else {
            if (btype == 1) fixed_codes(L.h);
            else if (!read_dynamic(b, L.h)) { status = GZ_CORRUPT; break; }
            const uint32_t lit_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.h.lit_maxlen), dist_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.h.dist_maxlen);
            bool bad = false;
            for (;;) {
                b.refill();
                const int s = decode_sym(b, L.h.lit_tab, LIT_PB, L.h.lit_count, L.h.lit_sym, lit_max);
                if (s < 0) { bad = true; break; }
                if (s < 256) {
                    L.ring[pos & RING_MASK] = (uint16_t)s;
                    pos++;
                    if (pos - flushed >= FLUSH) { if (!flush_full()) break; }
                    continue;
                }
                if (s == 256) break;
                if (s > 285) { bad = true; break; }
                uint32_t lb, le; len_code((uint32_t)s - 257u, lb, le);
                const uint32_t len = lb + b.get(le);
                b.refill();
                const int ds = decode_sym(b, L.h.dist_tab, DIST_PB, L.h.dist_count, L.h.dist_sym, dist_max);
                if (ds < 0 || ds > 29) { bad = true; break; }
                uint32_t db, de; dist_code((uint32_t)ds, db, de);
                const uint32_t dist = db + b.get(de);
                if ((dist > pos && known_window) || dist > pos + GZ_WSIZE) { bad = true; break; }
                // the copy, by all lanes: position pos + i takes what stands dist back — periodic when the match overlaps itself,
                // so every source lies in front of pos and the lanes do not depend on one another
                for (uint32_t i = lane; i < len; i += 64) {
                    const uint32_t j = i < dist ? i : i % dist;
                    const int src = (int)pos - (int)dist + (int)j;
                    uint16_t v;
                    if (src < 0) v = (uint16_t)(GZ_MARK + (GZ_WSIZE + src));
                    else if (pos - (uint32_t)src <= NEAR) v = L.ring[(uint32_t)src & RING_MASK];
                    else v = out[src];
                    L.ring[(pos + i) & RING_MASK] = v;
                }
                pos += len;
                if (pos - flushed >= FLUSH) { if (!flush_full()) break; }
            }
            if (status) break;
            if (bad) { status = GZ_CORRUPT; break; }
        }
        if (bfinal) { status = GZ_FINAL; break; }
    }
    // what is left in the ring
    for (uint32_t p = flushed + lane; p < pos; p += 64) out[p] = L.ring[p & RING_MASK];
    if (lane == 0) { ChunkOut r; r.n_sym = pos; r.end_pos = b.pos(); r.status = status; r.pad = 0; res[c] = r; }
}

// ---- the windows in front of the chunks ----------------------------------------------------------------------------------
// M[c][j] (c >= 1, j < 32768): byte j of the 32 KiB in front of chunk c (= the last 32768 bytes of everything before it), or
// — not known yet — 256 + a position in the window that many chunks FURTHER FRONT.  k_gz_win_init takes it from the tail
// of chunk c - 1 (a marker there, or a chunk shorter than a window, points into the window in front of c - 1); round r of
// k_gz_win_round replaces every pointer into the window of chunk c - 2^r by what stands there after round r - 1: a byte, or a
// pointer into the window of chunk c - 2^(r+1).  In FASTQ text a marker lives for ever (a quality line is a copy of the one
// before, back to the stream's first), so the chain of dependencies runs through ALL chunks: walked front to back by one
// workgroup it cost 6 us per chunk (24 ms for 4030 chunks); ceil(log2(chunks)) rounds over all windows at once cost ~3.
__global__ __launch_bounds__(1024) void k_gz_win_init(const uint16_t *__restrict__ syms, const unsigned long long *__restrict__ sym_off,
                                                     const ChunkOut *__restrict__ res, uint32_t n_chunks, uint16_t *__restrict__ M) {
    const uint32_t c = blockIdx.x + 1;
    if (c >= n_chunks) return;
    const uint16_t *S = syms + sym_off[c - 1];
    const long long n = (long long)res[c - 1].n_sym;
    uint16_t *W = M + (size_t)c * GZ_WSIZE;
    for (uint32_t j = threadIdx.x; j < GZ_WSIZE; j += 1024) {
        const long long i = n - (long long)GZ_WSIZE + (long long)j;
        W[j] = i >= 0 ? S[i] : (uint16_t)(GZ_MARK + (uint32_t)((long long)j + n));      // the chunk is shorter than a window: the rest comes from ITS window
    }
}
__global__ __launch_bounds__(1024) void k_gz_win_round(const uint16_t *__restrict__ Min, uint16_t *__restrict__ Mout, uint32_t n_chunks, uint32_t step,
                                                      uint32_t *__restrict__ bad) {
    const uint32_t c = blockIdx.x + 1;
    if (c >= n_chunks) return;
    const uint16_t *A = Min + (size_t)c * GZ_WSIZE;
    uint16_t *O = Mout + (size_t)c * GZ_WSIZE;
    const bool have_src = c > step;                            // window c - step exists (there is none in front of chunk 0)
    const uint16_t *B = Min + (size_t)(have_src ? c - step : 0) * GZ_WSIZE;
    for (uint32_t j = threadIdx.x; j < GZ_WSIZE; j += 1024) {
        uint32_t e = A[j];
        if (e >= GZ_MARK) {
            if (have_src) e = B[(e - GZ_MARK) & (GZ_WSIZE - 1)];
            else { atomicOr(bad, 1u); e = 0; }                   // a reference in front of the stream's first byte
        }
        O[j] = (uint16_t)e;
    }
}

// markers -> bytes: text[out_off[c] + i] = symbol < 256 ? symbol : window c [symbol - 256]   (grid: x over a chunk, y = chunk)
__global__ __launch_bounds__(256) void k_gz_resolve(const uint16_t *__restrict__ syms, const unsigned long long *__restrict__ sym_off,
                                                   const ChunkOut *__restrict__ res, const unsigned long long *__restrict__ out_off,
                                                   const uint16_t *__restrict__ M, uint8_t *__restrict__ text, uint32_t *__restrict__ bad) {
    const uint32_t c = blockIdx.y;
    const uint16_t *S = syms + sym_off[c];
    const unsigned long long n = res[c].n_sym;
    const uint16_t *W = M + (size_t)c * GZ_WSIZE;
    uint8_t *T = text + out_off[c];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256u) {
        uint32_t s = S[i];
        if (s >= GZ_MARK) {
            if (c == 0) { atomicOr(bad, 2u); s = 0; }
            else { s = W[(s - GZ_MARK) & (GZ_WSIZE - 1)]; if (s >= GZ_MARK) { atomicOr(bad, 4u); s = 0; } }      // (a window byte nobody resolved: never)
        }
        T[i] = (uint8_t)s;
    }
}

// CRC-32 (the gzip polynomial, reflected) of slices of `slice` bytes: one thread per slice, byte-wise table in LDS
__global__ __launch_bounds__(256) void k_gz_crc(const uint8_t *__restrict__ text, unsigned long long n, uint32_t slice, uint32_t n_slices,
                                               uint32_t *__restrict__ crc_out) {
    __shared__ uint32_t tab[256];
    {
        uint32_t cc = threadIdx.x;
        for (int k = 0; k < 8; k++) cc = (cc & 1u) ? 0xEDB88320u ^ (cc >> 1) : cc >> 1;
        tab[threadIdx.x] = cc;
    }
    __syncthreads();
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n_slices) return;
    const unsigned long long a = (unsigned long long)s * slice, e = min(n, a + slice);
    uint32_t crc = 0xFFFFFFFFu;
    unsigned long long p = a;
    // 16 bytes per load (slices start 16-byte aligned: `slice` is a multiple of 16 and the text block is aligned)
    for (; p + 16 <= e; p += 16) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + p);
        const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t x = ws[q];
#pragma unroll
            for (int r = 0; r < 4; r++) { crc = tab[(crc ^ x) & 0xFFu] ^ (crc >> 8); x >>= 8; }
        }
    }
    for (; p < e; p++) crc = tab[(crc ^ text[p]) & 0xFFu] ^ (crc >> 8);
    crc_out[s] = crc ^ 0xFFFFFFFFu;
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// CRC-32 of A || B from crc(A), crc(B): crc(A) advanced over len(B) zero bytes (a GF(2) matrix), XOR crc(B).  The slices
// have one length, so the matrix is built once (zlib 1.2.11 has no crc32_combine_gen).
struct Gf2 { uint32_t m[32]; };
uint32_t gf2_times(const Gf2 &a, uint32_t v) { uint32_t s = 0; for (int i = 0; v; v >>= 1, i++) if (v & 1u) s ^= a.m[i]; return s; }
Gf2 gf2_square(const Gf2 &a) { Gf2 r; for (int i = 0; i < 32; i++) r.m[i] = gf2_times(a, a.m[i]); return r; }
Gf2 crc_shift_matrix(uint64_t len_bytes) {              // operator "append len_bytes zero bytes"
    Gf2 odd; odd.m[0] = 0xEDB88320u; { uint32_t row = 1; for (int i = 1; i < 32; i++) { odd.m[i] = row; row <<= 1; } }   // one zero BIT
    Gf2 even = gf2_square(odd);                           // two bits
    odd = gf2_square(even);                               // four bits
    Gf2 result; for (int i = 0; i < 32; i++) result.m[i] = 1u << i;      // identity
    Gf2 cur = gf2_square(odd);                            // eight bits = one byte
    for (uint64_t n = len_bytes; n; n >>= 1) {
        if (n & 1u) { Gf2 r; for (int i = 0; i < 32; i++) r.m[i] = gf2_times(cur, result.m[i]); result = r; }
        cur = gf2_square(cur);
    }
    return result;
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Blk {                                               // a device block of the process-wide pool
    void *p = nullptr; size_t bytes = 0;
    ~Blk() { if (p) device_pool_release(p, bytes); }
    bool get(size_t b) { bytes = b ? b : 8; p = device_pool_alloc(bytes); return p != nullptr; }
    void *take() { void *q = p; p = nullptr; return q; }
};

}  // namespace

#define GZCHK(call)                                                                         \
    do {                                                                                    \
        const hipError_t _e = (call);                                                       \
        if (_e != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(_e); (void)hipStreamSynchronize(st); return -5; } \
    } while (0)

int gpu_inflate_member(const uint8_t *gz, size_t n, int device, void *stream, GpuText &out, std::string &err, GpuInflateStats *stats, bool raw) {
    GpuInflateStats local; GpuInflateStats &S = stats ? *stats : local;
    S = GpuInflateStats();
    out = GpuText();
    auto not_taken = [&](const char *why) { S.why_not = why; return 1; };
    // ---- gzip member header (RFC 1952); the trailer is taken to be the last 8 bytes (one member, nothing behind it:
    // checked against where the final block really ends)
    if (n < 18 || gz[0] != 0x1F || gz[1] != 0x8B || gz[2] != 8) return not_taken("not a gzip member");
    const unsigned flg = gz[3];
    size_t p = 10;
    if (flg & 4) {
        if (p + 2 > n) return not_taken("truncated header");
        const size_t xlen = gz[p] | ((size_t)gz[p + 1] << 8);
        // (BGZF: an extra field 'BC' — many small members: the host reader inflates them block-parallel)
        if (xlen >= 6 && p + 2 + xlen <= n && gz[p + 2] == 'B' && gz[p + 3] == 'C') return not_taken("BGZF");
        p += 2 + xlen;
    }
    if (flg & 8) { while (p < n && gz[p]) p++; p++; }
    if (flg & 16) { while (p < n && gz[p]) p++; p++; }
    if (flg & 2) p += 2;
    if (p + 8 >= n) return not_taken("truncated header");
    const uint8_t *def = gz + p;
    const size_t dn = n - p - 8;                                  // deflate data if this is the only member
    const char *mv = getenv("SHK_GUNZIP_DEVICE_MIN");
    const size_t min_bytes = (mv && *mv) ? (size_t)strtoull(mv, nullptr, 10) : ((size_t)4 << 20);
    if (dn < min_bytes) return not_taken("small member");
    const uint8_t *tr = gz + n - 8;
    const uint32_t want_crc = tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
    const uint32_t want_len = tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
    if ((uint64_t)dn * 64 < want_len) return not_taken("ISIZE out of proportion");      // (text deflates 3-6x; more than 64x is not FASTQ)
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); err = "hipSetDevice failed"; return -5; }
    hipStream_t st = (hipStream_t)stream;
    const double t_begin = now_ms();
    // ---- chunks (SHK_GUNZIP_DEVICE_CHUNK: bytes of compressed data each), at most 16384
    const char *cv = getenv("SHK_GUNZIP_DEVICE_CHUNK");
    // (one wave per chunk, 18 waves to a CU by their LDS: ~4600 chunks are resident at once on 256 CUs — a member of the
    // bench isolate's size is cut so that all its chunks run in one round; smaller members into 32 KiB chunks)
    uint64_t chunk_bytes = (cv && *cv) ? strtoull(cv, nullptr, 10) : std::max<uint64_t>(32u << 10, ((uint64_t)dn / 4200u + 4095u) & ~4095ull);
    if (chunk_bytes < 4096) chunk_bytes = 4096;
    while ((dn + chunk_bytes - 1) / chunk_bytes > 16384) chunk_bytes *= 2;
    const uint32_t C0 = (uint32_t)((dn + chunk_bytes - 1) / chunk_bytes);
    if (C0 < 2) return not_taken("small member");
    // ---- upload (the compressed bytes are all that crosses PCIe), zero padding behind
    Blk d_in, d_start;
    const size_t in_words = (dn + 3) / 4 + 2048 + 256;      // (8 KB of zero padding: the decoder checks for the end of the input once per 512 symbols; + the 128 dwords its reader keeps in flight)
    if (!d_in.get(in_words * 4) || !d_start.get((size_t)(C0 + 1) * 8)) { err = "out of device memory (gzip input)"; return -4; }
    GZCHK(hipMemsetAsync((char *)d_in.p + (dn & ~(size_t)3), 0, in_words * 4 - (dn & ~(size_t)3), st));
    GZCHK(hipMemcpyAsync(d_in.p, def, dn, hipMemcpyHostToDevice, st));
    GZCHK(hipMemsetAsync(d_start.p, 0, (size_t)(C0 + 1) * 8, st));
    GZCHK(hipStreamSynchronize(st));
    S.h2d_ms = now_ms() - t_begin;
    // ---- 1. block starts
    double t1 = now_ms();
    hipLaunchKernelGGL(k_gz_find_starts, dim3(C0 - 1), dim3(64), 0, st, (const uint32_t *)d_in.p, (uint64_t)dn, chunk_bytes, C0, (unsigned long long *)d_start.p);
    GZCHK(hipGetLastError());
    std::vector<unsigned long long> start(C0 + 1, 0);
    GZCHK(hipMemcpyAsync(start.data(), d_start.p, (size_t)C0 * 8, hipMemcpyDeviceToHost, st));
    GZCHK(hipStreamSynchronize(st));
    S.search_ms = now_ms() - t1;
    // (a chunk without a start is merged into its predecessor)
    std::vector<unsigned long long> stv; stv.push_back(0);
    for (uint32_t c = 1; c < C0; c++) if (start[c] != ~0ull && start[c] > stv.back()) stv.push_back(start[c]);
    const uint32_t C = (uint32_t)stv.size();
    if (C < 2 || C * 4u < C0) return not_taken("block starts not found");      // (binary data, stored blocks, fixed codes)
    // ---- 2. every chunk, with markers for what lies in front of it.  Room per chunk: 10 symbols per compressed byte
    // (SHK_GUNZIP_DEVICE_RATIO; FASTQ text deflates 3-6x) + a margin; a chunk that needs more makes the member fall back
    const char *rv = getenv("SHK_GUNZIP_DEVICE_RATIO");
    const uint64_t ratio = (rv && *rv) ? std::max<uint64_t>(2, strtoull(rv, nullptr, 10)) : 10;
    std::vector<unsigned long long> sym_off(C + 1, 0);
    for (uint32_t c = 0; c < C; c++) {
        const uint64_t comp = ((c + 1 < C ? stv[c + 1] : (uint64_t)dn * 8) - stv[c] + 7) / 8;
        const uint64_t room = (comp * ratio + 8192 + 511) & ~511ull;
        sym_off[c + 1] = sym_off[c] + room;
    }
    Blk d_syms, d_soff, d_res, d_bad;
    if (!d_syms.get((size_t)sym_off[C] * 2 + 64) || !d_soff.get((size_t)(C + 1) * 8) || !d_res.get((size_t)C * sizeof(ChunkOut)) || !d_bad.get(64)) {
        err = "out of device memory (gzip symbols)"; return -4;
    }
    t1 = now_ms();
    GZCHK(hipMemcpyAsync(d_start.p, stv.data(), (size_t)C * 8, hipMemcpyHostToDevice, st));
    GZCHK(hipMemcpyAsync(d_soff.p, sym_off.data(), (size_t)(C + 1) * 8, hipMemcpyHostToDevice, st));
    GZCHK(hipMemsetAsync(d_bad.p, 0, 64, st));
    hipLaunchKernelGGL(k_gz_decode, dim3(C), dim3(64), 0, st, (const uint32_t *)d_in.p, (uint64_t)dn, C, (const unsigned long long *)d_start.p,
                       (const unsigned long long *)d_soff.p, (uint16_t *)d_syms.p, (ChunkOut *)d_res.p);
    GZCHK(hipGetLastError());
    std::vector<ChunkOut> res(C);
    GZCHK(hipMemcpyAsync(res.data(), d_res.p, (size_t)C * sizeof(ChunkOut), hipMemcpyDeviceToHost, st));
    GZCHK(hipStreamSynchronize(st));
    S.decode_ms = now_ms() - t1;
    std::vector<unsigned long long> out_off(C + 1, 0);
    for (uint32_t c = 0; c < C; c++) {
        const uint32_t want = c + 1 < C ? GZ_ATSTOP : GZ_FINAL;
        if (res[c].status != want) return not_taken(res[c].status == GZ_NO_ROOM ? "a chunk inflates beyond its room" : "a chunk does not end where the next begins");
        if (c + 1 < C && res[c].end_pos != stv[c + 1]) return not_taken("a chunk does not end where the next begins");
        out_off[c + 1] = out_off[c] + res[c].n_sym;
    }
    const uint64_t total = out_off[C];
    // the final block must end right in front of the trailer (one member, nothing behind it)
    if ((res[C - 1].end_pos + 7) / 8 != (uint64_t)dn) return not_taken("more than one member, or data behind the member");
    if ((uint32_t)total != want_len) return not_taken("ISIZE mismatch");
    if (total == 0 || total >= ((uint64_t)1 << 32)) return not_taken("empty or beyond 4 GiB");
    // ---- 3. windows front to back, 4. markers -> bytes, CRC
    Blk d_ma, d_mb, d_ooff, d_text, d_crc;
    const uint32_t SLICE = 65536;
    const uint32_t n_slices = (uint32_t)((total + SLICE - 1) / SLICE);
    if (!d_ma.get((size_t)C * GZ_WSIZE * 2) || !d_mb.get((size_t)C * GZ_WSIZE * 2) || !d_ooff.get((size_t)(C + 1) * 8) || !d_text.get((size_t)total + 64) ||
        !d_crc.get((size_t)n_slices * 4)) {
        err = "out of device memory (inflated text)"; return -4;
    }
    t1 = now_ms();
    GZCHK(hipMemcpyAsync(d_ooff.p, out_off.data(), (size_t)(C + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gz_win_init, dim3(C - 1), dim3(1024), 0, st, (const uint16_t *)d_syms.p, (const unsigned long long *)d_soff.p, (const ChunkOut *)d_res.p, C,
                       (uint16_t *)d_ma.p);
    uint16_t *Mi = (uint16_t *)d_ma.p, *Mo = (uint16_t *)d_mb.p;
    for (uint32_t step = 1; step < C; step *= 2) {
        hipLaunchKernelGGL(k_gz_win_round, dim3(C - 1), dim3(1024), 0, st, (const uint16_t *)Mi, Mo, C, step, (uint32_t *)d_bad.p);
        std::swap(Mi, Mo);
    }
    GZCHK(hipGetLastError());
    if (getenv("SHK_GUNZIP_DEBUG")) { GZCHK(hipStreamSynchronize(st)); S.windows_ms = now_ms() - t1; }
    hipLaunchKernelGGL(k_gz_resolve, dim3(16, C), dim3(256), 0, st, (const uint16_t *)d_syms.p, (const unsigned long long *)d_soff.p, (const ChunkOut *)d_res.p,
                       (const unsigned long long *)d_ooff.p, (const uint16_t *)Mi, (uint8_t *)d_text.p, (uint32_t *)d_bad.p);
    GZCHK(hipMemsetAsync((char *)d_text.p + total, 0, 64, st));
    hipLaunchKernelGGL(k_gz_crc, dim3((n_slices + 255) / 256), dim3(256), 0, st, (const uint8_t *)d_text.p, (unsigned long long)total, SLICE, n_slices, (uint32_t *)d_crc.p);
    GZCHK(hipGetLastError());
    std::vector<uint32_t> crcs(n_slices);
    uint32_t h_bad[2] = {0, 0};
    char tail[4096];
    const size_t tail_n = (size_t)std::min<uint64_t>(total, sizeof tail);
    GZCHK(hipMemcpyAsync(crcs.data(), d_crc.p, (size_t)n_slices * 4, hipMemcpyDeviceToHost, st));
    GZCHK(hipMemcpyAsync(h_bad, d_bad.p, 8, hipMemcpyDeviceToHost, st));
    GZCHK(hipMemcpyAsync(tail, (char *)d_text.p + (total - tail_n), tail_n, hipMemcpyDeviceToHost, st));
    GZCHK(hipStreamSynchronize(st));
    S.resolve_ms = now_ms() - t1;
    if (h_bad[0]) return not_taken("a back-reference in front of the stream");
    // combine the slice checksums
    uint32_t crc = crcs[0];
    if (n_slices > 1) {
        const Gf2 M = crc_shift_matrix(SLICE);
        for (uint32_t s = 1; s + 1 < n_slices; s++) crc = gf2_times(M, crc) ^ crcs[s];
        const uint64_t last = total - (uint64_t)(n_slices - 1) * SLICE;
        crc = (uint32_t)crc32_combine(crc, crcs[n_slices - 1], (z_off_t)last);
    }
    if (crc != want_crc) return not_taken("CRC-32 mismatch");
    // ---- the text as gpu_upload_text hands it on: trailing blank lines cut off, 32 zero bytes behind it
    // (fastq_gpu.hip: trimmed_len — "\n\n" and "\n\r\n" at the end lose their last line end, repeatedly)
    size_t te = tail_n;
    for (;;) {
        if (te >= 2 && tail[te - 1] == '\n' && tail[te - 2] == '\n') { te -= 1; continue; }
        if (te >= 3 && tail[te - 1] == '\n' && tail[te - 2] == '\r' && tail[te - 3] == '\n') { te -= 2; continue; }
        break;
    }
    if (raw) te = tail_n;
    if (te < 8 && total > tail_n) return not_taken("kilobytes of blank lines at the end");
    size_t e = (size_t)(total - tail_n) + te;
    const bool unterminated = te == 0 || tail[te - 1] != '\n';
    if (e == 0) return not_taken("empty text");
    if (!raw) GZCHK(hipMemsetAsync((char *)d_text.p + e, 0, std::min<size_t>(32, (size_t)total + 64 - e), st));
    GZCHK(hipStreamSynchronize(st));
    out.pool_bytes = d_text.bytes; out.d = (uint8_t *)d_text.take();
    out.e = e; out.unterminated = unterminated; out.h2d_ms = S.h2d_ms;
    S.chunks = C; S.text_bytes = total; S.total_ms = now_ms() - t_begin;
    if (getenv("SHK_GUNZIP_DEBUG"))
        fprintf(stderr, "[inflate_gpu] %u chunks (%u cuts): upload %.2f ms, block starts %.2f, decode %.2f, windows (%.2f) + resolve + crc %.2f, total %.2f ms for %.3f GB of text\n",
                C, C0, S.h2d_ms, S.search_ms, S.decode_ms, S.windows_ms, S.resolve_ms, S.total_ms, (double)total / 1e9);
    return 0;
}

}  // namespace shk
