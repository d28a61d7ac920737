// count_part.h — minimiser-partitioned k-mer counting (the production counting path).
//
// Why: counting straight into one HBM hash table costs one scattered atomic per k-mer instance
// (~400 M at 100x coverage of a 5 Mbp isolate) and runs at the chip's scattered-atomic rate
// (measured 38 ms, profiles/r01_baseline_global_atomics).  Here identical k-mers are brought
// together first, so that counting happens in the 160 KB LDS of one CU:
//
//   pass 1  k_partition         reads -> super-k-mer records, scattered by minimiser partition
//   pass 2  k_count_partitions  one partition at a time per (persistent) workgroup: LDS record table +
//                               LDS k-mer table -> histogram + (k-mer, count) rows with count > T
//           k_ovf_scatter / k_count_buckets   partitions whose distinct k-mers exceed the LDS table
//                               (error-rich reads): split once more by k-mer hash, counted per bucket
//
// Minimiser = the m-mer (m = k - w + 1, w = 8, 16 or 18 m-mers per k-mer) of a k-mer with the smallest canonical ntHash
// (SPEC S3); partition = low bits of that hash.  Both strands of a k-mer share it, so every
// instance of a canonical k-mer lands in one partition.  A record is a run of consecutive
// k-mers of one read segment with the same partition: (n + k - 1) bases, 2-bit, little-endian,
// in RW = 2W 64-bit words, n-1 in the top 6 bits.  Results never depend on the partitioning.
//
// No global atomics in pass 1: workgroup g owns slice [p][g] of every partition buffer and keeps
// its (up to 16384) write cursors in LDS; the layout is a deterministic function of the input.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "kmer.h"

// Timing experiments (parts of a kernel switched off: wrong results) exist only in `make ABLATE=1` builds,
// which produce libshk_hip_ablate.so; in the shipped library every SHK_DBG(...) is the constant 0 and the
// branches behind it are compiled out (they also cost registers in kernels that run one workgroup per CU).
#ifndef SHK_ABLATE
#define SHK_ABLATE 0
#endif
#define SHK_DBG(x) (SHK_ABLATE ? (uint32_t)(x) : 0u)

namespace shk {

static constexpr int PART_THREADS = 1024;            // pass 1: 4 waves per SIMD
static constexpr int COUNT_THREADS = 1024;           // pass 2: 4 waves per SIMD hide the LDS latency
static constexpr int PART_MAX_P = 16384;             // LDS cursors: 64 KB

struct PartParams {
    uint32_t P, G, slice_cap;
    int k, m;
    uint32_t max_n;
    uint32_t dbg_nostore;        // timing experiments only (SHK_DEBUG_NOSTORE): 1 = every record to slot 0 of its slice, 2 = no flush
    uint32_t dbg_flush_at;       // timing experiments only (SHK_DEBUG_P1FLUSH, ABLATE builds): descriptors a lane may hold before its wave flushes
    unsigned long long *dbg_clk; // timing experiments only (SHK_DEBUG_P1CLK, ABLATE builds): wave-cycles per phase of k_partition, summed over the waves
};

template <int RW> struct Rec { uint64_t w[RW]; };

// 4-way select kept as compare/select on scalars (an indexable array would go to scratch)
__device__ __forceinline__ uint64_t sel4(uint32_t b, uint64_t t0, uint64_t t1, uint64_t t2, uint64_t t3) {
    uint64_t lo = (b & 1) ? t1 : t0;
    uint64_t hi = (b & 1) ? t3 : t2;
    return (b & 2) ? hi : lo;
}

// ---- pass 1 -------------------------------------------------------------------------------
// One 1024-thread workgroup per CU, one lane per read segment.  Every WAVE stages a tile of 64 segments
// of its own in LDS (coalesced dword loads; round 4 — the tile loop in k_partition says why); every lane
// then walks its segment base by base: rolling canonical
// ntHash (32-bit state, kmer.h) of the m-mers, sliding-window minimum in registers, run detection.
// Finished runs go to a LANE-PRIVATE descriptor list in LDS (slot i of lane l at [i][l]: conflict
// free, no ballot, no atomics); a wave turns its descriptors into records by itself whenever a lane's
// list fills up and at the end of the tile, so the walk needs no workgroup barrier (the first
// version shared one list and paid two 16-wave barriers per 16 bases: 68 % of the wave cycles were
// waits, profiles/r01_s2_start).
//
// The runs of a lane follow one another without a gap (a run starts at the k-mer where the one before
// it ended, the first at k-mer 0 of the segment), so a descriptor is just {partition, length}: the
// flush adds the lengths up from the position it has reached (`fpos`).  Closing a run costs the walk
// three vector instructions and one LDS store (round 3: eight and two).
//
// The window of a k-mer is NBLK blocks of WBLK m-mers (m = k - NBLK*WBLK + 1).  With one block the
// minimum of the window is min(suffix of the block before, prefix of this block); with two it is
// min3(suffix of the block before the last, the whole last block, prefix of this block) — the same
// instruction count per base; k = 31 uses two blocks of 9 (18 m-mers instead of 16: runs of 9.7 k-mers
// on average instead of 8.3, a seventh fewer records to build, store and fetch again).
static constexpr int PART_WAVES = PART_THREADS / 64;
static constexpr uint32_t LDESC_CAP = 12;            // descriptors per lane
static constexpr int STAGE_PF = 12;                  // prefetch registers per lane: ceil(WSTAGE / 64)

static constexpr uint32_t WSTAGE = 724;              // packed words of a wave's own tile (11 584 bases: 64 reads of 181)
static constexpr uint32_t LONG_NK = 2048;            // a segment with more k-mers is walked in pieces, a lane per piece
static constexpr uint32_t PIECE_K = 160;             // k-mers per piece (64 pieces and their k-1 bases of overlap fit a stage)
struct PartShared {
    uint2 nt_lut[16];                 // [out<<2|in]: x = rol(seed[out],m)^seed[in], y = ror(seed[~out],1)^rol(seed[~in],m-1); first: the address fits the instruction's offset field
    uint32_t stage[PART_WAVES * WSTAGE + 24]; // one tile per WAVE: the next one is fetched into registers while this one is walked
    uint32_t desc[PART_WAVES][LDESC_CAP][64]; // partition | (n-1) << 14
    uint32_t cursor[PART_MAX_P];
    uint32_t red[PART_WAVES];
};
// + for records of up to four words: the bit mask of a record of n k-mers (one 16-byte LDS read per two words instead
// of two 64-bit shifts, two selects and two compares per word)
template <int RW> struct PartSharedT : PartShared {
    static constexpr bool HAS_MASKS = RW <= 4;
    __attribute__((aligned(16))) uint32_t rmask[HAS_MASKS ? 64 * 2 * RW : 4];
};

// build one record from the staged tile: bases [off, off + n + k - 1), n = n1 + 1; rmask = the mask row of n1
template <int RW>
__device__ __forceinline__ void part_build_record(const uint32_t *stage, const uint32_t *rmask, uint32_t off, uint32_t n1, int k,
                                                  uint64_t (&out)[RW]) {
    const uint32_t wi = off >> 4, s = 2u * off;                // (the funnel shift takes the low five bits of s)
    uint32_t w[2 * RW + 1];
#pragma unroll
    for (int o = 0; o < 2 * RW + 1; o++) w[o] = stage[wi + o];
    if constexpr (PartSharedT<RW>::HAS_MASKS) {
        uint32_t mk[2 * RW];
#pragma unroll
        for (int o = 0; o < 2 * RW; o += 4) {
            const uint4 q = *reinterpret_cast<const uint4 *>(rmask + o);
            mk[o] = q.x; mk[o + 1] = q.y; mk[o + 2] = q.z; mk[o + 3] = q.w;
        }
#pragma unroll
        for (int o = 0; o < RW; o++) {
            const uint32_t lo = __builtin_amdgcn_alignbit(w[2 * o + 1], w[2 * o], s) & mk[2 * o];
            const uint32_t hi = __builtin_amdgcn_alignbit(w[2 * o + 2], w[2 * o + 1], s) & mk[2 * o + 1];
            out[o] = (uint64_t)lo | ((uint64_t)hi << 32);
        }
    } else {
        const uint32_t nbits = 2u * (n1 + (uint32_t)k);
#pragma unroll
        for (int o = 0; o < RW; o++) {
            const uint32_t lo = __builtin_amdgcn_alignbit(w[2 * o + 1], w[2 * o], s);
            const uint32_t hi = __builtin_amdgcn_alignbit(w[2 * o + 2], w[2 * o + 1], s);
            uint64_t v = (uint64_t)lo | ((uint64_t)hi << 32);
            const int rem = (int)nbits - 64 * o;            // bits of this word that belong to the run
            if (rem <= 0) v = 0;
            else if (rem < 64) v &= (1ull << rem) - 1ull;
            out[o] = v;
        }
    }
    out[RW - 1] |= (uint64_t)n1 << 58;
}
// A record and its reverse complement describe the same k-mers (pass 2 canonicalises every k-mer it expands),
// and at high coverage every locus is read on both strands: the smaller of the two spellings is stored, so
// that pass 2's record table sees one record per locus instead of two (half the expansions).
template <int RW>
__device__ __forceinline__ void rec_canonicalise(uint64_t (&r)[RW], uint32_t n, int k) {
    const uint32_t L = n + (uint32_t)k - 1u;                 // bases in the record (2L <= 64 RW - 6 bits)
    uint64_t d[RW], rc[RW];
#pragma unroll
    for (int o = 0; o < RW; o++) d[o] = r[o];
    d[RW - 1] &= (1ull << 58) - 1ull;                        // without the length field
    // reverse the order of all 32*RW two-bit groups and complement them ...
#pragma unroll
    for (int o = 0; o < RW; o++) {
        uint64_t y = __builtin_bitreverse64(d[RW - 1 - o]);
        y = ((y & 0x5555555555555555ull) << 1) | ((y >> 1) & 0x5555555555555555ull);
        rc[o] = ~y;
    }
    // ... and move base 0 of the reversed string down to bit 0: shift right by 64 RW - 2L bits
    const uint32_t sh = 64u * RW - 2u * L, ws = sh >> 6, bs = sh & 63u;
    uint64_t t[RW];
#pragma unroll
    for (int o = 0; o < RW; o++) {
        uint64_t lo = 0, hi = 0;
#pragma unroll
        for (int q = 0; q < RW; q++) {                         // (register arrays: selects, no dynamic indexing)
            if ((uint32_t)q == (uint32_t)o + ws) lo = rc[q];
            if ((uint32_t)q == (uint32_t)o + ws + 1u) hi = rc[q];
        }
        t[o] = bs ? (lo >> bs) | (hi << (64u - bs)) : lo;
    }
    // keep the low 2L bits
#pragma unroll
    for (int o = 0; o < RW; o++) {
        const int rem = (int)(2u * L) - 64 * o;
        if (rem <= 0) t[o] = 0;
        else if (rem < 64) t[o] &= (1ull << rem) - 1ull;
    }
    bool smaller = false, decided = false;                   // t < d as multiword integers?
#pragma unroll
    for (int o = RW - 1; o >= 0; o--) {
        if (!decided && t[o] != d[o]) { smaller = t[o] < d[o]; decided = true; }
    }
    if (smaller) {
#pragma unroll
        for (int o = 0; o < RW; o++) r[o] = t[o];
        r[RW - 1] |= (uint64_t)(n - 1u) << 58;
    }
}
template <int RW>
__device__ __forceinline__ void part_store_record(const uint64_t (&out)[RW], uint64_t *__restrict__ dst) {
#pragma unroll
    for (int o = 0; o < RW; o += 2) {
        ulonglong2 v2; v2.x = out[o]; v2.y = out[o + 1];
        *reinterpret_cast<ulonglong2 *>(dst + o) = v2;
    }
}

// (LDS pointers keep their address space across the call: with generic pointers the compiler falls
// back to flat loads with 64-bit address arithmetic for every stage / descriptor access)
#define SHK_LDS __attribute__((address_space(3)))
#define SHK_GLOBAL __attribute__((address_space(1)))
// The wave's descriptor lists -> records.  `cnt` descriptors in this lane's list, `fpos` = tile-relative base position of
// its first unflushed run; returns the position behind the last one.
// (Inlined: as a function, everything the walk keeps across the call had to sit in the callee-saved half of the
// registers — the next tile's words, three rounds of offsets, the block's hashes and suffix minima — and what did not fit
// was spilled around EVERY load, a wait for memory each.)
template <int RW>
__device__ __forceinline__ uint32_t wave_flush(SHK_LDS PartSharedT<RW> *sh_l, uint32_t wave, uint32_t cnt, uint32_t fpos,
                                            int k, uint32_t G, uint32_t slice_cap, uint32_t g, SHK_GLOBAL uint64_t *recs_g,
                                            uint32_t dbg_arg = 0) {
    const uint32_t dbg = SHK_DBG(dbg_arg);
    PartSharedT<RW> *sh = (PartSharedT<RW> *)sh_l;       // address space is inferred from the cast
    uint64_t *recs = (uint64_t *)recs_g;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t mx = 0;                                     // longest list in the wave (cnt <= LDESC_CAP): ballots, no shuffles
#pragma unroll
    for (uint32_t c = 1; c <= LDESC_CAP; c++) if (__ballot(cnt >= c)) mx = c;
    // One LDS round trip per record on the critical path: the next descriptor is fetched while this one is worked on, and
    // the bases and the mask of a record are requested together with its slot (a record whose slice is full is built
    // and dropped: the host repeats the pass with more room).
    uint32_t ds = sh->desc[wave][0][lane];
    for (uint32_t d = 0; d < mx; d++) {
        const uint32_t ds_next = sh->desc[wave][min(d + 1u, LDESC_CAP - 1u)][lane];
        if (d < cnt) {
            const uint32_t p = ds & 0x3FFFu, n1 = ds >> 14;
            const uint32_t off = fpos;
            fpos += n1 + 1u;
            if (dbg != 2) {                                      // (2: timing experiment, the walk alone)
                // LDS cursor of slice [p][g].  No branch on the room: a record behind the end of its slice goes to the slice's last
                // slot (the cursor keeps counting, the host sees the overflow and repeats both passes with the exact room, so what
                // pass 2 read from that slot meanwhile is thrown away) — with a branch the compiler sinks the loads of the bases
                // behind the wait for the cursor, one more LDS round trip per record
                // (timing experiments 3 .. 6: no cursor atomic / no bases and mask / no store / no mask)
                const uint32_t idx = dbg == 3 ? (fpos & 15u) : min(atomicAdd(&sh->cursor[p], 1u), slice_cap - 1u);
                uint64_t r[RW];
                if (dbg == 4) {
#pragma unroll
                    for (int o = 0; o < RW; o++) r[o] = ((uint64_t)off << 20) | n1;
                } else part_build_record<RW>(sh->stage + wave * WSTAGE, dbg == 6 ? sh->rmask : sh->rmask + n1 * (2 * RW), off, n1, k, r);
                const uint32_t slice = p * G + g;                            // (P * G <= 2^22)
                if (dbg == 7) {                                              // (timing experiment: every record twice, side by side — twice the requests, the same lines)
                    const uint32_t i2 = min(2u * idx, slice_cap - 2u);
                    part_store_record<RW>(r, recs + ((uint64_t)slice * slice_cap + i2) * RW);
                    part_store_record<RW>(r, recs + ((uint64_t)slice * slice_cap + i2 + 1u) * RW);
                } else
                if (dbg != 5) part_store_record<RW>(r, recs + ((uint64_t)slice * slice_cap + (dbg == 1 ? 0u : idx)) * RW);
                else if (r[0] == 0x1234567ull) recs[0] = r[1];               // (keeps the record alive)
            }
        }
        ds = ds_next;
    }
    return fpos;
}

// W: key words (records have RW = 2W words); the window of a k-mer = NBLK blocks of WBLK m-mers
template <int W, int WBLK, int NBLK>
__global__ __launch_bounds__(PART_THREADS) void k_partition(const uint32_t *__restrict__ bases,
                                                            const uint32_t *__restrict__ seg_off,
                                                            uint32_t n_seg, PartParams pp,
                                                            uint64_t *__restrict__ recs,
                                                            uint32_t *__restrict__ fill,
                                                            uint32_t *__restrict__ flags) {
    constexpr int RW = 2 * W;
    static_assert(WBLK <= 16, "a block of m-mers must fit one 32-bit window");
    static_assert(NBLK == 1 || NBLK == 2, "one or two blocks per window");
    constexpr int DESC_CHECK = WBLK % 3 == 0 ? 3 : WBLK % 4 == 0 ? 4 : 2;   // steps between room checks (one new descriptor per lane and step at most)
    static_assert(WBLK % DESC_CHECK == 0, "checks at equal distances across blocks");
    __shared__ PartSharedT<RW> sh;
    const uint32_t g = blockIdx.x;
    const int lane = threadIdx.x & 63;
    // (told to the compiler as wave-uniform: every wave iterates over tiles of its own, and with a "divergent" wave number
    // the whole tile loop was compiled as predicated vector code)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = pp.k, m = pp.m;
    const uint32_t pmask = pp.P - 1u;
    const uint32_t max_n = pp.max_n;
    // cursors continue where an earlier batch left this workgroup's slices
    for (uint32_t p = threadIdx.x; p < pp.P; p += PART_THREADS) sh.cursor[p] = fill[(uint64_t)p * pp.G + g];
    // ntHash roll terms (SPEC S3) for every (outgoing, incoming) base pair: 16 x 8 B in LDS, one
    // ds_read_b64 per step instead of three 4-way register selects
    if (threadIdx.x < 16) {
        const uint32_t out = threadIdx.x >> 2, in = threadIdx.x & 3u;
        uint2 v;
        v.x = rol32(nt32_seed(out), (unsigned)m) ^ nt32_seed(in);
        v.y = ror32(nt32_seed(3u - out), 1) ^ rol32(nt32_seed(3u - in), (unsigned)(m - 1));
        sh.nt_lut[threadIdx.x] = v;
    }
    if constexpr (PartSharedT<RW>::HAS_MASKS) {
        for (uint32_t e = threadIdx.x; e < 64u * 2u * RW; e += PART_THREADS) {
            const uint32_t n1 = e / (2u * RW), o = e % (2u * RW);
            const int rem = (int)(2u * (n1 + (uint32_t)k)) - 32 * (int)o;        // bits of dword o that belong to a record of n1 + 1 k-mers
            sh.rmask[e] = rem <= 0 ? 0u : rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u);
        }
    }
    __syncthreads();

    // ---- tiles.  Every WAVE has a tile of its own (round 4; before, the 16 waves of the workgroup shared one tile and met
    // at two barriers per tile: measured with s_memtime, a wave spent 30 % of its cycles between its last step and the next
    // tile's first — waiting for the slowest wave, the oldest wave of a SIMD being served first — and 12 % finding the next
    // tile).  Wave v of the grid takes the 64 segments [64 T, 64 T + 64), T = v, v + waves, ...; their offsets are requested
    // one round ahead, the packed words of the round's first tile travel in registers while the tile before is walked.  A
    // round is one tile when its 64 segments fit the stage (150-base reads: always); otherwise as many consecutive segments
    // as fit, and a segment of more than LONG_NK k-mers is walked in pieces of PIECE_K k-mers, a lane per piece (a piece is
    // a segment of its own to the walk: runs end at piece edges, the k-mers are the same).  No barrier until the end.
    static_assert((WSTAGE + 63) / 64 <= STAGE_PF, "prefetch registers");
    static_assert(64u * PIECE_K + 256u + 32u <= 16u * (WSTAGE - 1u), "64 pieces fit a stage");
    const uint32_t n_super = (n_seg + 63u) >> 6, n_waves = gridDim.x * PART_WAVES;
    SHK_LDS uint32_t *const stage_l = (SHK_LDS uint32_t *)&sh.stage[wave * WSTAGE];
    const uint32_t *const stage = (const uint32_t *)stage_l;
    SHK_LDS uint32_t *const dbase = (SHK_LDS uint32_t *)&sh.desc[wave][0][lane];
#if SHK_ABLATE
    unsigned long long clk_pre = 0, clk_walk = 0, clk_flush = 0, clk_tail = 0, clk_t0 = 0;
#define SHK_CLK(x) x
#else
#define SHK_CLK(x)
#endif
    auto load_seg = [&](uint32_t T, uint32_t &s0, uint32_t &s1) {      // lanes behind the last segment get an empty one
        const uint32_t i = min(64u * T + (uint32_t)lane, n_seg);
        s0 = seg_off[i]; s1 = seg_off[min(i + 1u, n_seg)];
    };
    // a tile of whole segments [start, end) of the round, or (end == start) the long segment `start` alone
    struct Plan { uint32_t end, w0, nwords; };
    auto plan = [&](uint32_t s0, uint32_t s1, uint32_t start) -> Plan {
        const uint32_t Ls = s1 - s0;
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)s0, start) >> 4;     // (the builtin is typed int: beyond 2^31 bases an arithmetic shift put w0 out of range, nothing "fitted" and every segment was walked alone, in pieces — correct and 20 x slower: configs[4]'s share)
        const bool is_long = Ls >= (uint32_t)k + LONG_NK;
        const bool fits = ((s1 + 15u) >> 4) - w0 <= WSTAGE - 1u;
        const unsigned long long stop = __ballot((uint32_t)lane >= start && (is_long || !fits));
        const uint32_t end = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
        Plan pl; pl.end = end; pl.w0 = w0;
        pl.nwords = end > start ? (((uint32_t)__builtin_amdgcn_readlane((int)s1, end - 1u) + 15u) >> 4) - w0 : 0u;
        return pl;
    };
    uint32_t T = g * PART_WAVES + wave;                        // this wave's round
    uint32_t s0c = 0, s1c = 0, s0n = 0, s1n = 0, s0nn = 0, s1nn = 0;     // offsets of this round, the next, the one after it
    if (T < n_super) load_seg(T, s0c, s1c);
    if (T + n_waves < n_super) load_seg(T + n_waves, s0n, s1n);
    bool prefetched = false;                                   // pf[] holds the first tile of the round (plan pl_pf)
    uint32_t pf[STAGE_PF];
    Plan pl_pf{0, 0, 0};
    for (; T < n_super; T += n_waves) {
        uint32_t start = 0, round = 0;                         // next segment of the round / next 64 pieces of a long segment
        const uint32_t n_here = min(64u, n_seg - 64u * T);
        bool first_tile = true;
        while (start < n_here) {
            SHK_CLK(clk_t0 = __builtin_amdgcn_s_memtime();)
            // ---- the tile: which lanes walk what
            const Plan pl = (first_tile && prefetched) ? pl_pf : plan(s0c, s1c, start);
            uint32_t L = 0, rel = 0, w0 = pl.w0, nwords = pl.nwords;
            uint32_t next_start = pl.end, next_round = 0;
            if (pl.end > start) {
                if ((uint32_t)lane >= start && (uint32_t)lane < pl.end) { L = s1c - s0c; rel = s0c - (w0 << 4); }
            } else {                                           // pieces 64 round .. 64 round + 63 of segment `start`
                const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)s0c, start), a1 = (uint32_t)__builtin_amdgcn_readlane((int)s1c, start);
                const uint32_t nk = a1 - a0 - (uint32_t)k + 1u;
                const uint32_t k0 = 64u * PIECE_K * round;             // first k-mer of this round
                const uint32_t kend = min(k0 + 64u * PIECE_K, nk);
                w0 = (a0 + k0) >> 4;
                nwords = ((a0 + kend + (uint32_t)k - 1u + 15u) >> 4) - w0;
                const uint32_t kp = k0 + PIECE_K * (uint32_t)lane;
                if (kp < kend) { L = min(PIECE_K, kend - kp) + (uint32_t)k - 1u; rel = a0 + kp - (w0 << 4); }
                if (kend < nk) { next_start = start; next_round = round + 1u; } else next_start = start + 1u;
            }
            // ---- its packed words: out of the prefetch registers, or fetched now (a round of several tiles)
            if (!(first_tile && prefetched)) {
#pragma unroll
                for (int i = 0; i < STAGE_PF; i++) {
                    const uint32_t idx = (uint32_t)lane + 64u * (uint32_t)i;
                    pf[i] = idx < nwords + 1u ? bases[w0 + idx] : 0u;          // +1: the spare word
                }
                prefetched = false;                            // (a later tile of the round: the registers held the next round's words)
            }
#pragma unroll
            for (int i = 0; i < STAGE_PF; i++) {
                const uint32_t idx = (uint32_t)lane + 64u * (uint32_t)i;
                if (idx < WSTAGE) stage_l[idx] = pf[i];
            }
            // ---- behind the round's first tile: the first tile of the next round into the registers, the offsets of the
            // round after it on their way
            // (one-word keys only: with wider records the flush needs the twelve registers — kept across the walk they were
            // spilled, and the spill traffic shared the memory pipeline with the record stores: k = 51 with its errors left in
            // took 1.54 ms where the shared-tile kernel had taken 1.18)
            if (first_tile) {
                prefetched = false;
                if (T + 2u * n_waves < n_super) load_seg(T + 2u * n_waves, s0nn, s1nn);       // (used a whole round from now)
                if (W == 1 && T + n_waves < n_super) {
                    pl_pf = plan(s0n, s1n, 0u);
                    if (pl_pf.end > 0u) {
                        prefetched = true;
#pragma unroll
                        for (int i = 0; i < STAGE_PF; i++) {
                            const uint32_t idx = (uint32_t)lane + 64u * (uint32_t)i;
                            pf[i] = idx < pl_pf.nwords + 1u ? bases[pl_pf.w0 + idx] : 0u;
                        }
                    }
                }
            }
            first_tile = false;
            // longest and shortest segment of the wave (reads of one length: two instructions)
            uint32_t maxL = __builtin_amdgcn_readfirstlane(L), minL = maxL;
            if (__ballot(L != maxL)) {
                maxL = L; minL = L;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    maxL = max(maxL, (uint32_t)__shfl_xor((int)maxL, o));
                    minL = min(minL, (uint32_t)__shfl_xor((int)minL, o));
                }
                maxL = __builtin_amdgcn_readfirstlane(maxL); minL = __builtin_amdgcn_readfirstlane(minL);
            }

            // ---- per-lane walk; every lane of the wave runs the same (block, t) schedule ----------
            // Bases come from 64-bit windows loaded once per block (a block of WBLK <= 16 bases spans
            // at most two packed words), never from per-lane reloads inside the step loop.
            auto window32 = [&](uint32_t pos) -> uint32_t {  // the 16 bases from stream position pos
                const uint32_t wi = min(pos >> 4, WSTAGE + 14u);
                return __builtin_amdgcn_alignbit(stage[wi + 1], stage[wi], 2u * (pos & 15u));
            };
            Nt32State nt{0, 0};
            // prologue: the first m-mer (bases 0 .. m-1)
            for (int jb = 0; jb < m; jb += 16) {
                const uint32_t wv = window32(rel + (uint32_t)jb);
                for (int t = 0; t < 16 && jb + t < m; t++)
                    nt32_init_step(nt, (wv >> (2 * t)) & 3u, (unsigned)(jb + t));
            }
            // suffix minima of the blocks before: block b reads those of block b - NBLK from sr[b % NBLK] and leaves its own there
            uint32_t hreg[WBLK], sr0[WBLK], sr1[NBLK == 2 ? WBLK : 1];
#pragma unroll
            for (int t = 0; t < WBLK; t++) { hreg[t] = 0xFFFFFFFFu; sr0[t] = 0xFFFFFFFFu; if constexpr (NBLK == 2) sr1[t] = 0xFFFFFFFFu; }
            // the current run: partition run_p, started at k-mer (run_lim >> 14) - max_n; NO_RUN before the first k-mer.  The
            // k-mer index i of a step is the same in every lane, so run lengths are scalar-minus-vector; run_lim is kept
            // shifted to where the descriptor holds the length (k-mer indices stay below 2^15).
            constexpr uint32_t NO_RUN = 0xFFFFFFFFu;
            uint32_t run_lim = 0, run_p = NO_RUN;
            SHK_LDS uint32_t *dptr = dbase;                  // this lane's next descriptor slot
            uint32_t fpos = rel;                             // where this lane's first unflushed run starts
            // the waves of one SIMD flush at different fill levels: a flush is a chain of LDS round trips,
            // it overlaps with the VALU-bound walk of the others only if they do not all flush together
            SHK_LDS uint32_t *const flush_at = dbase + 64u * (SHK_DBG(pp.dbg_flush_at) ? min(SHK_DBG(pp.dbg_flush_at), LDESC_CAP - DESC_CHECK) - min(wave >> 2, SHK_DBG(pp.dbg_flush_at) - 1u)
                                                                                          : LDESC_CAP - DESC_CHECK - (wave >> 2));
            auto flush = [&]() {
                SHK_CLK(const unsigned long long f0 = __builtin_amdgcn_s_memtime();)
                fpos = wave_flush<RW>((SHK_LDS PartSharedT<RW> *)&sh, wave, ((uint32_t)(uintptr_t)dptr - (uint32_t)(uintptr_t)dbase) >> 8, fpos, k, pp.G, pp.slice_cap, g,
                                      (SHK_GLOBAL uint64_t *)recs, SHK_DBG(pp.dbg_nostore));
                dptr = dbase;
                SHK_CLK(clk_flush += __builtin_amdgcn_s_memtime() - f0;)
            };
            const uint32_t n_mmers_max = maxL >= (uint32_t)m ? maxL - (uint32_t)m + 1u : 0u;
            const uint32_t n_blocks = (n_mmers_max + WBLK - 1) / WBLK;
            // blocks whose WBLK m-mers exist in every lane of the wave (and are behind the first k-mer): no bounds logic
            const uint32_t n_full = minL >= (uint32_t)m ? (minL - (uint32_t)m + 1u) / WBLK : 0u;
            auto emit = [&](uint32_t c_end) {                // close the run at k-mer i_end: c_end = (i_end - 1 + max_n) << 14
                *dptr = (c_end - run_lim) | run_p;           // n - 1 = i_end - 1 - run_start
                dptr += 64;
            };
            auto do_block = [&](auto full_tag, uint32_t bq, uint32_t (&srs)[WBLK], const uint32_t mid) {
                constexpr bool FULL = decltype(full_tag)::value;
                const uint32_t q0 = bq * WBLK;                             // first m-mer of the block
                // the state holds m-mer q; after using it, base q leaves and base q+m enters.
                // zo/ze: nibble j = (outgoing << 2 | incoming) of step t = 2j+1 / 2j  -> LUT index
                const uint32_t X = window32(rel + q0), Y = window32(rel + q0 + (uint32_t)m);
                const uint32_t ze = ((X & 0x33333333u) << 2) | (Y & 0x33333333u);
                const uint32_t zo = (X & 0xCCCCCCCCu) | ((Y >> 2) & 0x33333333u);
                uint32_t pm = NBLK == 2 ? mid : 0xFFFFFFFFu;
#pragma unroll
                for (int t = 0; t < WBLK; t++) {
                    if (t % DESC_CHECK == 0) {
                        if (__ballot(dptr > flush_at)) flush();            // wave-uniform
                    }
                    const uint32_t q = q0 + (uint32_t)t;
                    const uint32_t j = q + (uint32_t)m - 1u;               // last base of m-mer q (wave-uniform)
                    const uint32_t i = j - (uint32_t)k + 1u;               // k-mer completed by m-mer q (wave-uniform)
                    bool have = true;
                    uint32_t h = nt32_canonical(nt);
                    if (!FULL) { have = j < L; h = have ? h : 0xFFFFFFFFu; }
                    // roll to m-mer q+1 (unconditionally: a state past the segment end is never used)
                    {
                        const uint32_t z = (t & 1) ? zo : ze;
                        const uint2 term = sh.nt_lut[(z >> (4 * (t >> 1))) & 15u];
                        nt.fh = __builtin_amdgcn_alignbit(nt.fh, nt.fh, 31) ^ term.x;   // rol 1
                        nt.rh = __builtin_amdgcn_alignbit(nt.rh, nt.rh, 1) ^ term.y;    // ror 1
                    }
                    hreg[t] = h;
                    pm = min(pm, h);
                    // k-mer i = q - NBLK * WBLK + 1 completes here; its window is suffix(block b - NBLK, t+1) [+ block b - 1] + prefix(this block, t)
                    uint32_t kmin = pm;
                    if (t < WBLK - 1) kmin = min(srs[(t + 1) % WBLK], kmin);
                    const bool valid = FULL ? true : (have && (bq > (uint32_t)(NBLK - 1) || (bq == (uint32_t)(NBLK - 1) && t == WBLK - 1)));
                    const uint32_t p = kmin & pmask;
                    // a new run starts here if the partition changes or the current run is full
                    if (valid && (p != run_p || run_lim <= (i << 14))) {
                        if (FULL || run_p != NO_RUN) emit((i - 1u + max_n) << 14);
                        run_lim = (i + max_n) << 14; run_p = p;
                    }
                }
                // suffix minima of this block for the block NBLK further on
                srs[WBLK - 1] = hreg[WBLK - 1];
#pragma unroll
                for (int t = WBLK - 2; t >= 0; t--) srs[t] = min(hreg[t], srs[t + 1]);
            };
            SHK_CLK({ const unsigned long long t1 = __builtin_amdgcn_s_memtime(); clk_pre += t1 - clk_t0; clk_t0 = t1; })
            for (uint32_t bq = 0; bq < n_blocks; bq++) {
                // The next tile's words (requested before the walk) are waited for HERE, a few blocks in and before this tile's
                // first record leaves: loads and stores share one in-order counter, and a wait placed where the words are used —
                // behind the tile's last flush — also sat out the completion of every record store of the tile.
                if (bq == 3u) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
                const bool full = bq >= (uint32_t)NBLK && bq < n_full;
                if constexpr (NBLK == 1) {
                    if (full) do_block(std::true_type{}, bq, sr0, 0u);
                    else do_block(std::false_type{}, bq, sr0, 0u);
                } else {
                    if (bq & 1u) {
                        if (full) do_block(std::true_type{}, bq, sr1, sr0[0]);
                        else do_block(std::false_type{}, bq, sr1, sr0[0]);
                    } else {
                        if (full) do_block(std::true_type{}, bq, sr0, sr1[0]);
                        else do_block(std::false_type{}, bq, sr0, sr1[0]);
                    }
                }
            }
            SHK_CLK({ const unsigned long long t1 = __builtin_amdgcn_s_memtime(); clk_walk += t1 - clk_t0; clk_t0 = t1; })
            // close the last run of every segment (it ends with the segment's last k-mer)
            if (__ballot(dptr >= dbase + 64u * LDESC_CAP)) flush();
            if (run_p != NO_RUN) emit((L - (uint32_t)k + max_n) << 14);
            if (__ballot(dptr != dbase)) flush();
            start = next_start; round = next_round;
            SHK_CLK(clk_tail += __builtin_amdgcn_s_memtime() - clk_t0;)
        }
        s0c = s0n; s1c = s1n; s0n = s0nn; s1n = s1nn;
    }
    SHK_CLK(if (pp.dbg_clk && lane == 0) { atomicAdd(pp.dbg_clk + 0, clk_pre); atomicAdd(pp.dbg_clk + 1, clk_walk); atomicAdd(pp.dbg_clk + 2, clk_flush); atomicAdd(pp.dbg_clk + 3, clk_tail); atomicAdd(pp.dbg_clk + 4, 1ull); })
    __syncthreads();
    // publish this workgroup's slice fills (may exceed slice_cap: the host then retries bigger) and
    // their maximum (one global atomic per workgroup)
    uint32_t mx = 0;
    for (uint32_t p = threadIdx.x; p < pp.P; p += PART_THREADS) {
        const uint32_t c = sh.cursor[p];
        fill[(uint64_t)p * pp.G + g] = c;
        mx = max(mx, c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    if (lane == 0) sh.red[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < PART_WAVES; w++) mx = max(mx, sh.red[w]);
        if (mx) atomicMax(&flags[2], mx);
    }
}

// ---- pass 2 -------------------------------------------------------------------------------
// Record-level dedupe (phase A of pass 2): whole super-k-mer records with a multiplicity.  At high
// coverage the same genomic run is seen in many reads; it is counted once here and expanded once.
// rst: 0 empty, 1 being written, >= 3 ready with multiplicity rst-2
template <int W, uint32_t SR_> struct RecTable {
    static constexpr uint32_t SR = SR_;
    __attribute__((aligned(16))) uint64_t w[SR][2 * W];   // one record = adjacent words: 16-byte LDS reads
    uint32_t rst[SR];
    uint16_t order[SR];                                 // occupied record slots, sorted by record length
    uint32_t nhist[64], nbase[64];
};

// the k-mer table alone (k_count_buckets needs nothing else: two workgroups share a CU's LDS there)
template <int W> struct KmerTable;
template <> struct KmerTable<1> {
    static constexpr uint32_t S = 5376;                 // k-mer table: 12 B / slot -> 63 KB
    static constexpr uint32_t NB = S / 4;               // buckets of 4 keys = two ds_read_b128
    __attribute__((aligned(16))) uint64_t key0[S];
    uint32_t cnt[S];
};
// W >= 2: one key array per word, a state word per slot (0 empty, 1 being written, 2 ready)
template <int W> struct KmerTable {
    static constexpr uint32_t S = (73728u / (8u * W + 8u)) & ~63u;            // 3072 / 2304 / 1792 slots for W = 2 / 3 / 4
    uint64_t key[W][S];
    uint32_t cnt[S];
    uint32_t state[S];
};
// pass 2 proper: the LDS is split between the k-mer table (63-72 KB) and the record table (70-88 KB)
template <int W> struct CountShared : KmerTable<W> {
    RecTable<W, (W == 1 ? 4096u : ((71680u / (16u * W + 6u)) & ~63u))> rt;   // 4096 / 1856 / 1280 / 1024 records
};

struct CountCtlCore {                                   // what the table, the emit and the residue rounds need
    uint32_t histo[500];
    uint32_t n_used, overflow, n_emit, emit_base_lo, emit_base_hi, sp, wave_cursor, rec_used, n_recs;
    // pending residue classes of the key hash: entry = classes res + j*step (j = next .. factor-1) modulo step*factor
    uint32_t st_res[16], st_step[16], st_factor[16], st_next[16];
    uint32_t prog_num, prog_den;                        // how far the round got when the table filled up
    unsigned long long n_inst, tried, part_inst;        // (n_inst: over the workgroup's lifetime; part_inst: of the partition being handed over)
};
struct CountCtl : CountCtlCore {                        // + the run table of a partition (k_count_partitions)
    uint32_t pre[257];                                  // exclusive prefix of slice fills (G <= 256)
    unsigned long long roff[256];                       // address/16 of every run's first record (copied once: no dependent global load per fetch)
    uint32_t next_pi;                                   // the partition after the current one (grabbed from the launch's work counter)
};

// Table placement hash: add/shift/xor only (Jenkins one-at-a-time finaliser); integer multiplies
// are quarter rate on CDNA and this runs once per k-mer instance.
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h += h << 10; h ^= h >> 6; h += h << 3; h ^= h >> 11; h += h << 15;
    return h;
}
template <int W> __device__ __forceinline__ uint32_t km_mix32(const Kmer<W> &x) {
    uint32_t h = 0x9E3779B9u;
#pragma unroll
    for (int j = 0; j < W; j++) {
        const uint32_t lo = (uint32_t)x.w[j], hi = (uint32_t)(x.w[j] >> 32);
        h = mix32(h ^ lo ^ __builtin_amdgcn_alignbit(hi, hi, 17));
    }
    return h;
}

// Counts are u32 and SATURATE at 0xFFFFFFFF (SPEC S4).  A count can only pass 2^32 when its partition holds
// that many k-mer instances, which the caller knows before it starts (records x k-mers per record): `sat` is
// uniform per partition and false for everything but such giants, so the common path keeps its no-return add.
// Saturating form: an add that wrapped pins the count to the top; every later add wraps again and repairs it,
// so whatever the interleaving the count reads 0xFFFFFFFF once the adds are done (the emit scan is behind a barrier).
__device__ __forceinline__ void cnt_add(uint32_t *c, uint32_t weight, bool sat) {
    if (!sat) { atomicAdd(c, weight); return; }
    const uint32_t old = atomicAdd(c, weight);
    if (old > 0xFFFFFFFFu - weight) atomicMax(c, 0xFFFFFFFFu);
}

// insert into the LDS table; returns false when the probe sequence is exhausted
// (COUNT_USED = false: the caller keeps ctl.n_used itself from the return value 2 = "new key")
template <int W, bool COUNT_USED = true>
__device__ __forceinline__ int lds_insert(KmerTable<W> &tb, CountCtlCore &ctl, const Kmer<W> &key, uint32_t h,
                                          uint32_t weight, bool sat = false) {
    constexpr uint32_t S = KmerTable<W>::S;
    if constexpr (W == 1) {
        // 4-way buckets: the whole bucket comes back from one pair of 16-byte LDS reads, so a hit
        // (the common case at >1x coverage) costs one LDS round trip instead of a serial probe chain.
        // A key lives in the first bucket of its probe sequence that had a free slot when it arrived;
        // slots never change once written, so "bucket has a free slot and no match" proves absence.
        constexpr uint32_t NB = KmerTable<1>::NB;
        uint32_t b = (uint32_t)(((uint64_t)h * NB) >> 32);
        const unsigned long long kk = key.w[0];
        for (uint32_t probes = 0; probes < 64u;) {          // a longer chain means the table is (locally) full: split
            const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&tb.key0[4 * b]);
            const ulonglong2 c = *reinterpret_cast<const ulonglong2 *>(&tb.key0[4 * b + 2]);
            int j = a.x == kk ? 0 : a.y == kk ? 1 : c.x == kk ? 2 : c.y == kk ? 3 : -1;
            if (j < 0) {
                const int e = a.x == ~0ull ? 0 : a.y == ~0ull ? 1 : c.x == ~0ull ? 2 : c.y == ~0ull ? 3 : -1;
                if (e < 0) { b = b + 1 == NB ? 0 : b + 1; probes++; continue; }      // bucket full
                const unsigned long long old = atomicCAS((unsigned long long *)&tb.key0[4 * b + e], ~0ull, kk);
                if (old == ~0ull) {
                    if constexpr (COUNT_USED) atomicAdd(&ctl.n_used, 1u);
                    cnt_add(&tb.cnt[4 * b + e], weight, sat);
                    return 2;
                }
                else if (old == kk) j = e;
                else continue;                              // lost the slot to another key: look again
            }
            cnt_add(&tb.cnt[4 * b + j], weight, sat);
            return 1;
        }
        return 0;
    } else {
        uint32_t slot = (uint32_t)(((uint64_t)h * S) >> 32);
        uint32_t probes = 0;
        for (;;) {
            uint32_t st = __hip_atomic_load(&tb.state[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            bool won = false;
            if (st == 0) { st = atomicCAS(&tb.state[slot], 0u, 1u); won = (st == 0); }
            if (won) {
#pragma unroll
                for (int j = 0; j < W; j++) tb.key[j][slot] = key.w[j];
                __hip_atomic_store(&tb.state[slot], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if constexpr (COUNT_USED) atomicAdd(&ctl.n_used, 1u);
                cnt_add(&tb.cnt[slot], weight, sat);
                return 2;
            }
            if (st == 1) continue;                          // owner is mid-write
            bool eq = true;
#pragma unroll
            for (int j = 0; j < W; j++) eq = eq && tb.key[j][slot] == key.w[j];
            if (eq) { cnt_add(&tb.cnt[slot], weight, sat); return 1; }
            slot = slot + 1 == S ? 0 : slot + 1;
            if (++probes >= 48u) return 0;              // table (locally) full: the caller splits the class
        }
    }
}

// phase A: count a whole record; false = table saturated (the caller then expands it directly)
template <int W, typename RT>
__device__ __forceinline__ bool rec_insert(RT &rt, CountCtlCore &ctl, const Rec<2 * W> &rec, uint32_t weight = 1u) {
    constexpr uint32_t SR = RT::SR;
    uint32_t h = 0x9E3779B9u;
#pragma unroll
    for (int o = 0; o < 2 * W; o++) {
        const uint32_t lo = (uint32_t)rec.w[o], hi = (uint32_t)(rec.w[o] >> 32);
        h = mix32(h ^ lo ^ __builtin_amdgcn_alignbit(hi, hi, 9 + 3 * o));
    }
    uint32_t slot = (uint32_t)(((uint64_t)h * SR) >> 32);
    // once the table is saturated (error-rich reads: most records are unique) only a short look for an
    // existing copy is worth it: walking 64 slots of a 7/8 full table per record cost 4.5 ms per launch
    const uint32_t max_probes = ctl.rec_used >= (SR / 8) * 7 ? 6u : 64u;
    for (uint32_t probes = 0; probes < max_probes;) {
        uint32_t st = __hip_atomic_load(&rt.rst[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (st == 0) {
            if (ctl.rec_used >= (SR / 8) * 7) return false;             // keep the probe chains short
            st = atomicCAS(&rt.rst[slot], 0u, 1u);
            if (st == 0) {
#pragma unroll
                for (int o = 0; o < 2 * W; o++) rt.w[slot][o] = rec.w[o];
                __hip_atomic_fetch_add(&rt.rst[slot], 1u + weight, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // 1 -> 2 + weight
                atomicAdd(&ctl.rec_used, 1u);
                return true;
            }
        }
        if (st == 1) continue;                                          // owner is mid-write
        bool eq = true;
#pragma unroll
        for (int o = 0; o < 2 * W; o++) eq = eq && rt.w[slot][o] == rec.w[o];
        if (eq) { atomicAdd(&rt.rst[slot], weight); return true; }
        slot = slot + 1 == SR ? 0 : slot + 1;
        probes++;
    }
    return false;
}

// empty table, zero round-local histogram (all threads; ends with a barrier)
template <int W> __device__ __forceinline__ void kmer_table_reset(KmerTable<W> &tb, CountCtlCore &ctl) {
    constexpr uint32_t S = KmerTable<W>::S;
    for (uint32_t s = threadIdx.x; s < S; s += COUNT_THREADS) {
        tb.cnt[s] = 0;
        if constexpr (W == 1) tb.key0[s] = ~0ull; else tb.state[s] = 0;
    }
    for (uint32_t b = threadIdx.x; b < 500; b += COUNT_THREADS) ctl.histo[b] = 0;
    __syncthreads();
}
template <int W> __device__ __forceinline__ void table_reset(CountShared<W> &tb, CountCtl &ctl) {
    for (uint32_t s = threadIdx.x; s < decltype(tb.rt)::SR; s += COUNT_THREADS) tb.rt.rst[s] = 0;
    kmer_table_reset<W>(tb, ctl);
}

// a finished round: histogram of the table's counts, rows with count > threshold appended to the output.
// One scan of the table feeds the round-local histogram and lists the slots to emit in LDS (list: room
// for S slot numbers); then one global atomic reserves the rows and they are written densely — one
// store instruction per 64 rows and array, whatever the density of solid rows in the table.
template <int W, typename LT>
__device__ __forceinline__ void table_emit(KmerTable<W> &tb, CountCtlCore &ctl, unsigned long long mine, uint32_t threshold,
                                           unsigned long long *__restrict__ histo, KeyArr<W> out_keys,
                                           uint32_t *__restrict__ out_cnt, unsigned long long out_cap,
                                           unsigned long long *__restrict__ out_cursor, LT *list, uint32_t dbg_arg = 0,
                                           uint32_t *hist_accum = nullptr /* LDS: the round's histogram is added here instead of to `histo` */,
                                           uint32_t bias = 0 /* Bloom mode: the occurrence that only set the filter's bits */) {
    const uint32_t dbg = SHK_DBG(dbg_arg);
    constexpr uint32_t S = KmerTable<W>::S;
    const int lane = threadIdx.x & 63;
    if (dbg == 10) return;                                  // (timing experiments 10, 11, 4: stop after successive stages)
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if (lane == 0 && mine) atomicAdd(&ctl.n_inst, mine);
    for (uint32_t s = threadIdx.x; s < S; s += COUNT_THREADS) {
        // (S is a multiple of 64: whole waves.)  Singletons — nearly every slot when the reads carry
        // errors — go in with one LDS atomic per wave instead of 64 serialised ones on one address
        uint32_t c = tb.cnt[s];
        if (bias && c) { c = c > 0xFFFFFFFFu - bias ? 0xFFFFFFFFu : c + bias; tb.cnt[s] = c; }     // (rows are written from tb.cnt below)
        const unsigned long long m1 = __ballot(c == 1u);
        if (lane == 0 && m1) atomicAdd(&ctl.histo[0], (uint32_t)__popcll(m1));
        if (c > 1u) atomicAdd(&ctl.histo[c >= 500 ? 499 : c - 1], 1u);
        const bool e = c > threshold;                       // (c == 0: an empty slot, never above a threshold >= 0)
        const unsigned long long em = __ballot(e);
        if (!em) continue;
        uint32_t wb = 0;
        if (lane == 0) wb = atomicAdd(&ctl.n_emit, (uint32_t)__popcll(em));
        wb = __shfl(wb, 0);
        if (e) list[wb + (uint32_t)__popcll(em & ((1ull << lane) - 1ull))] = (LT)s;
    }
    if (dbg == 11) return;
    __syncthreads();
    const uint32_t n_emit = ctl.n_emit;
    if (threadIdx.x == 0) {
        unsigned long long base = (n_emit && dbg != 5) ? atomicAdd(out_cursor, (unsigned long long)n_emit) : 0ull;
        ctl.emit_base_lo = (uint32_t)base; ctl.emit_base_hi = (uint32_t)(base >> 32);
    }
    if (hist_accum) {                                       // (thread b owns bin b: no atomics)
        for (uint32_t b = threadIdx.x; b < 500; b += COUNT_THREADS) hist_accum[b] += ctl.histo[b];
    } else if (dbg != 1) {                                  // (dbg 1: timing experiment, no global histogram flush)
        for (uint32_t b = threadIdx.x; b < 500; b += COUNT_THREADS)
            if (ctl.histo[b]) atomicAdd(&histo[b], (unsigned long long)ctl.histo[b]);
    }
    if (!n_emit || dbg == 4) return;                        // (uniform)
    __syncthreads();
    const unsigned long long gbase = ((unsigned long long)ctl.emit_base_hi << 32) | ctl.emit_base_lo;
    for (uint32_t i = threadIdx.x; i < n_emit; i += COUNT_THREADS) {
        const uint32_t sl = list[i];
        const unsigned long long o = gbase + i;
        if (o < out_cap) {
            Kmer<W> x;
            if constexpr (W == 1) x.w[0] = tb.key0[sl];
            else {
#pragma unroll
                for (int j = 0; j < W; j++) x.w[j] = tb.key[j][sl];
            }
            out_keys.store(o, x);
            out_cnt[o] = tb.cnt[sl];
        }
    }
}

// a partition whose distinct k-mers do not fit the LDS table (reported by k_count_partitions)
struct OvfRec { uint32_t p, est_distinct; unsigned long long instances; };
// ... and how it is repartitioned at k-mer level: F buckets of `cap` k-mers starting at k-mer index `base`
struct OvfItem { uint32_t p, F, cap, pad; unsigned long long base; };
static constexpr uint32_t OVF_MAX_F = 256;

// A partition's records arrive as S runs (local: one per producer workgroup; sharded: one per
// source rank; batched: one per batch): run j of partition p holds run_cnt[p*S+j] records starting at
// device address 16 * run_addr16[p*S+j] (absolute, so the runs of one partition may live in different
// allocations: every batch of reads keeps its own record buffer).
struct RunView {
    const unsigned long long *run_addr16;
    const uint32_t *run_cnt;
    uint32_t S;            // runs per partition (<= 256)
    int k;
    uint32_t dbg;          // timing experiments only (SHK_DEBUG_P2)
    // sharded counting with records deduplicated by their source rank (k_dedupe_partitions): the record at address
    // 16 * (rec_base16 + i * W) stands for weights[i] identical records.  nullptr: every record counts once.
    const uint32_t *weights = nullptr;
    unsigned long long rec_base16 = 0;
};

// run table of the local layout recs[p][g][slice_cap]
__global__ __launch_bounds__(256) void k_make_runs(const uint32_t *__restrict__ fill, PartParams pp,
                                                   const uint64_t *recs, uint32_t rec_words,
                                                   unsigned long long *__restrict__ run_addr16,
                                                   uint32_t *__restrict__ run_cnt) {
    const uint64_t n = (uint64_t)pp.P * pp.G;
    const unsigned long long base16 = (unsigned long long)recs >> 4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        run_addr16[i] = base16 + i * pp.slice_cap * (rec_words / 2u);
        run_cnt[i] = min(fill[i], pp.slice_cap);
    }
}

// records held per partition (sum over the producer slices)
__global__ __launch_bounds__(256) void k_part_totals(const uint32_t *__restrict__ fill, PartParams pp,
                                                     unsigned long long *__restrict__ totals) {
    const uint32_t p = blockIdx.x;
    unsigned long long t = 0;
    for (uint32_t g = threadIdx.x; g < pp.G; g += blockDim.x) t += min(fill[(uint64_t)p * pp.G + g], pp.slice_cap);
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) totals[p] = part[0] + part[1] + part[2] + part[3];
}

// dense copy of partition p's records to dst + base[p] records (send buffer of the shard layer)
template <int RW>
__global__ __launch_bounds__(256) void k_pack_partition(const uint64_t *__restrict__ recs,
                                                        const uint32_t *__restrict__ fill, PartParams pp,
                                                        const unsigned long long *__restrict__ base,
                                                        uint64_t *__restrict__ dst) {
    __shared__ uint32_t pre[257];
    const uint32_t p = blockIdx.x;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        uint32_t run = 0;
        for (uint32_t g0 = 0; g0 < pp.G; g0 += 64) {
            const uint32_t g = g0 + threadIdx.x;
            const uint32_t f = g < pp.G ? min(fill[(uint64_t)p * pp.G + g], pp.slice_cap) : 0u;
            uint32_t incl = f;
            for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
            if (g < pp.G) pre[g] = run + incl - f;
            run += __shfl(incl, 63);
        }
        if (threadIdx.x == 0) pre[pp.G] = run;
    }
    __syncthreads();
    const uint32_t R = pre[pp.G];
    const uint64_t *src_p = recs + (uint64_t)p * pp.G * pp.slice_cap * RW;
    uint64_t *dst_p = dst + base[p] * RW;
    for (uint32_t r = threadIdx.x; r < R; r += blockDim.x) {
        uint32_t lo = 0, hi = pp.G;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= r) lo = mid; else hi = mid; }
        const uint64_t *src = src_p + ((uint64_t)lo * pp.slice_cap + (r - pre[lo])) * RW;
#pragma unroll
        for (int o = 0; o < RW; o += 2)
            *reinterpret_cast<ulonglong2 *>(dst_p + (uint64_t)r * RW + o) = *reinterpret_cast<const ulonglong2 *>(src + o);
    }
}

// several packed batches become one: the S runs of partition p (one per batch) are copied back to back to
// dst[base[p] ..) — a handle may receive any number of batches, a run table holds 256 per partition
template <int RW>
__global__ __launch_bounds__(256) void k_merge_runs(RunView rvw, const unsigned long long *__restrict__ base,
                                                    uint64_t *__restrict__ dst) {
    __shared__ uint32_t pre[257];
    __shared__ unsigned long long roff[256];
    const uint32_t p = blockIdx.x, S_runs = rvw.S;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        uint32_t run = 0;
        for (uint32_t g0 = 0; g0 < S_runs; g0 += 64) {
            const uint32_t g = g0 + threadIdx.x;
            const uint32_t f = g < S_runs ? rvw.run_cnt[(uint64_t)p * S_runs + g] : 0u;
            if (g < S_runs) roff[g] = rvw.run_addr16[(uint64_t)p * S_runs + g];
            uint32_t incl = f;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
            if (g < S_runs) pre[g] = run + incl - f;
            run += __shfl(incl, 63);
        }
        if (threadIdx.x == 0) pre[S_runs] = run;
    }
    __syncthreads();
    const uint32_t R = pre[S_runs];
    uint64_t *dst_p = dst + base[p] * RW;
    for (uint32_t r = threadIdx.x; r < R; r += blockDim.x) {
        uint32_t lo = 0, hi = S_runs;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= r) lo = mid; else hi = mid; }
        const uint64_t *src = reinterpret_cast<const uint64_t *>(roff[lo] << 4) + (uint64_t)(r - pre[lo]) * RW;
#pragma unroll
        for (int o = 0; o < RW; o += 2)
            *reinterpret_cast<ulonglong2 *>(dst_p + (uint64_t)r * RW + o) = *reinterpret_cast<const ulonglong2 *>(src + o);
    }
}

template <int W, bool WEIGHTED = false /* sharded counting: records carry multiplicities (RunView::weights) */>
__global__ __launch_bounds__(COUNT_THREADS) void k_count_partitions(
    RunView rvw, uint32_t n_parts, uint32_t threshold,
    unsigned long long *__restrict__ histo, KeyArr<W> out_keys, uint32_t *__restrict__ out_cnt,
    unsigned long long out_cap, unsigned long long *__restrict__ out_cursor,
    unsigned long long *__restrict__ n_inst, uint32_t *__restrict__ flags,
    const uint32_t *__restrict__ part_list /* nullable: partitions to process */,
    OvfRec *__restrict__ ovf /* nullable: partitions that do not fit are listed here instead of being split */,
    uint32_t *__restrict__ ovf_n,
    uint32_t *__restrict__ work_counter /* zero at launch: partitions beyond a workgroup's first one are handed out in order
                            of demand (partition sizes follow the minimiser distribution: with a fixed share of 16 each the
                            slowest workgroup decides) */,
    uint32_t probe_blocks, uint32_t defer_after /* != 0: workgroups start in order, so the first ones act as a sample
                            of the partitions (partition = minimiser hash).  ovf_n[1] counts the partitions that were
                            really tried (low 16 bits) and those of them that overflowed (high 16 bits); a workgroup
                            numbered >= probe_blocks that finds >= defer_after overflows AND >= 3/4 of the tried ones
                            overflowed takes the input for error-rich and hands its partition to the k-mer-level
                            repartition without trying it.  Only the path taken depends on the timing, never the counts. */) {
    constexpr int RW = 2 * W;
    constexpr uint32_t S = CountShared<W>::S;
    __shared__ CountShared<W> tb;
    __shared__ CountCtl ctl;
    __shared__ uint32_t whist[500];                     // this workgroup's histogram over all its partitions
    __shared__ uint32_t wtot[4];
    const int lane = threadIdx.x & 63;
    const int k = rvw.k;
    const uint32_t S_runs = rvw.S;
    // PERSISTENT workgroups (one per CU: the tables take the whole LDS), workgroup g takes partitions
    // g, g + G, ...  The run table of the next partition travels in registers while the current one is
    // counted, the histogram stays in LDS until the workgroup is done: per partition there is no launch,
    // no wait for the run table and no histogram flush (what 8192 instead of 4096 partitions used to cost).
    for (uint32_t b = threadIdx.x; b < 500; b += COUNT_THREADS) whist[b] = 0;
    if (threadIdx.x == 0) ctl.n_inst = 0;
    auto load_runs = [&](uint32_t pi, uint32_t &cnt, unsigned long long &addr) {
        cnt = 0; addr = 0;
        if (pi < n_parts && threadIdx.x < S_runs) {
            const uint32_t pp = part_list ? part_list[pi] : pi;
            cnt = rvw.run_cnt[(uint64_t)pp * S_runs + threadIdx.x];
            addr = rvw.run_addr16[(uint64_t)pp * S_runs + threadIdx.x];
        }
    };
    uint32_t nx_cnt; unsigned long long nx_addr;
    load_runs(blockIdx.x, nx_cnt, nx_addr);
  uint32_t pi_next = 0;
  for (uint32_t pi = blockIdx.x; pi < n_parts; pi = pi_next) {
    const uint32_t p = part_list ? part_list[pi] : pi;
    if (threadIdx.x == 0) ctl.next_pi = gridDim.x + atomicAdd(work_counter, 1u);     // (read after the two barriers below)
    // exclusive prefix of the run lengths of this partition (S_runs <= 256: threads 0..255, one run each)
    {
        const uint32_t f = nx_cnt;
        uint32_t incl = f;
        if (threadIdx.x < 256) {
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
            if (lane == 63) wtot[threadIdx.x >> 6] = incl;
        }
        __syncthreads();                                 // (also: the previous partition is done with ctl / the tables)
        if (threadIdx.x < 256) {
            uint32_t off = 0;
            for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) off += wtot[w];
            if (threadIdx.x < S_runs) { ctl.pre[threadIdx.x] = off + incl - f; ctl.roff[threadIdx.x] = nx_addr; }
        }
        if (threadIdx.x == 0) {
            ctl.pre[S_runs] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            ctl.sp = 1; ctl.st_res[0] = 0; ctl.st_step[0] = 1; ctl.st_factor[0] = 1; ctl.st_next[0] = 0;
            ctl.part_inst = 0; ctl.n_used = 0;
            uint32_t force = 0;
            if (ovf && defer_after && pi >= probe_blocks) {
                const uint32_t x = __hip_atomic_load(&ovf_n[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t n_over = x >> 16, n_tried = x & 0xFFFFu;
                force = (n_over >= defer_after && 4u * n_over >= 3u * n_tried) ? 1u : 0u;
            }
            ctl.overflow = force;
        }
    }
    __syncthreads();
    pi_next = ctl.next_pi;                               // (into a register: thread 0 writes the word again at the top of the next turn)
    load_runs(pi_next, nx_cnt, nx_addr);                 // the next partition's run table: in flight during this one
   [&]() {                                              // one partition; `return` = done with it
    const uint32_t R = ctl.pre[S_runs];
    // a k-mer count can only reach 2^32 (SPEC S4: saturating) in a partition of >= 2^32 instances = R x (<= 64 per record)
    const bool sat = R >= (1u << 26) || WEIGHTED;
    if (SHK_DBG(rvw.dbg) == 5) return;                            // timing experiment: launch + run prefix only

    // hand the whole partition to the k-mer-level repartition (k_ovf_scatter / k_count_buckets): report the
    // estimated number of distinct k-mers (distinct / instance ratio of what was inserted before the table
    // filled up, times all instances) and the exact number of instances
    auto defer = [&](unsigned long long mine, bool was_tried) {
        if (threadIdx.x == 0) ctl.tried = 0;
        __syncthreads();
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
        if (lane == 0 && mine) atomicAdd(&ctl.tried, mine);
        unsigned long long inst = 0;
        for (uint32_t r = threadIdx.x; r < R; r += COUNT_THREADS) {
            uint32_t lo = 0, hi = S_runs;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ctl.pre[mid] <= r) lo = mid; else hi = mid; }
            const uint64_t *src = reinterpret_cast<const uint64_t *>(ctl.roff[lo] << 4) + (uint64_t)(r - ctl.pre[lo]) * RW;
            unsigned long long wgt = 1ull;
            if constexpr (WEIGHTED) wgt = rvw.weights[(ctl.roff[lo] - rvw.rec_base16) / W + (r - ctl.pre[lo])];
            inst += ((src[RW - 1] >> 58) + 1ull) * wgt;
        }
        for (int o = 32; o > 0; o >>= 1) inst += __shfl_down(inst, o);
        if (lane == 0 && inst) atomicAdd(&ctl.part_inst, inst);
        __syncthreads();
        if (threadIdx.x == 0 && R) {
            const double est = ctl.tried ? (double)ctl.n_used * (double)ctl.part_inst / (double)ctl.tried : (double)ctl.part_inst;
            const uint32_t slot = atomicAdd(ovf_n, 1u);
            OvfRec o; o.p = p; o.est_distinct = !was_tried ? 0u /* not tried */ : est > 4.0e9 ? 0xFFFFFFFFu : est < 1.0 ? 1u : (uint32_t)est; o.instances = ctl.part_inst;
            ovf[slot] = o;
        }
        if (threadIdx.x == 0 && was_tried && defer_after) atomicAdd(&ovf_n[1], 0x10001u);      // tried, and it overflowed
    };
    if (ctl.overflow) { defer(0ull, false); return; }            // (uniform: written before the barrier above)

    while (true) {
        __syncthreads();
        if (ctl.sp == 0) break;
        const uint32_t top = ctl.sp - 1;
        const uint32_t res = ctl.st_res[top] + ctl.st_next[top] * ctl.st_step[top], mod = ctl.st_step[top] * ctl.st_factor[top];
        __syncthreads();
        if (threadIdx.x == 0) {
            if (++ctl.st_next[top] == ctl.st_factor[top]) ctl.sp--;
            ctl.n_used = 0; ctl.overflow = 0; ctl.n_emit = 0; ctl.wave_cursor = 0; ctl.rec_used = 0;
            ctl.prog_num = 0; ctl.prog_den = 1;
        }
        table_reset<W>(tb, ctl);

        unsigned long long mine = 0;
        // the record of the NEXT batch is requested before the current one is expanded, so its
        // HBM/L2 latency hides behind ~n*60 instructions of work
        // Every wave takes a contiguous share of the partition's records and its lanes consecutive records,
        // so a lane's run index only ever moves forward by a step or two: no binary search per record
        // (eight dependent LDS reads — the longest link of the per-record latency chain).
        const uint32_t per_wave = (((R + (COUNT_THREADS / 64) - 1) / (COUNT_THREADS / 64)) + 63u) & ~63u;
        const uint32_t w_begin = min(R, (threadIdx.x >> 6) * per_wave), w_end = min(R, w_begin + per_wave);
        uint32_t run_lo = 0, run_next = 0;                  // run of the lane's last record, first record of the next run
        {
            uint32_t lo = 0, hi = S_runs;                   // run with pre[lo] <= r < pre[lo+1], once per round
            const uint32_t r = min(w_begin + (uint32_t)lane, R ? R - 1u : 0u);
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ctl.pre[mid] <= r) lo = mid; else hi = mid; }
            run_lo = lo; run_next = ctl.pre[lo + 1];
        }
        auto fetch = [&](uint32_t r, Rec<RW> &rec, uint32_t &wgt) -> uint32_t {
            wgt = 1u;
            if (r >= w_end) return 0u;
            while (r >= run_next) { run_lo++; run_next = ctl.pre[run_lo + 1]; }
            const uint64_t *src = reinterpret_cast<const uint64_t *>(ctl.roff[run_lo] << 4) + (uint64_t)(r - ctl.pre[run_lo]) * RW;
            if constexpr (WEIGHTED) wgt = rvw.weights[(ctl.roff[run_lo] - rvw.rec_base16) / W + (r - ctl.pre[run_lo])];
#pragma unroll
            for (int o = 0; o < RW; o += 2) {
                const ulonglong2 v2 = *reinterpret_cast<const ulonglong2 *>(src + o);
                rec.w[o] = v2.x; rec.w[o + 1] = v2.y;
            }
            return 1u;
        };
        // expand one record: window s of the record (little-endian 2-bit) read as an integer IS the
        // reverse-complement k-mer, complemented: rc = ~(rec >> 2s) & mask.  The forward k-mer is
        // its revcomp once, then rolls.  No priming over the first k-1 bases.  Each k-mer is
        // added with `weight` (the multiplicity of the record).
        auto expand = [&](Rec<RW> rec, uint32_t n, uint32_t weight) {
            Kmer<W> f = km_zero<W>();
            for (uint32_t s = 0; s < n; s++) {
                if (ctl.overflow || ctl.n_used > (S / 10) * 9) return;   // the round is lost: do not walk full-table probe chains
                if (s) {
#pragma unroll
                    for (int o = 0; o < RW - 1; o++) rec.w[o] = (rec.w[o] >> 2) | (rec.w[o + 1] << 62);
                    rec.w[RW - 1] >>= 2;
                }
                Kmer<W> rv;
#pragma unroll
                for (int j = 0; j < W; j++) rv.w[j] = ~rec.w[j];
                rv.w[W - 1] &= km_topmask<W>(k);
                if (s) km_push_back<W>(f, (uint32_t)(rec.w[W - 1] >> ((2 * (k - 1)) & 63)) & 3u, k);
                else f = km_revcomp<W>(rv, k);
                const bool use_r = km_less<W>(rv, f);
                Kmer<W> c;
#pragma unroll
                for (int j = 0; j < W; j++) c.w[j] = use_r ? rv.w[j] : f.w[j];
                const uint32_t h = km_mix32<W>(c);
                if (mod == 1 || ((h >> 20) & (mod - 1u)) == res) {         // sub-round filter: bits 20.. of h
                    if (!lds_insert<W>(tb, ctl, c, h ^ __builtin_amdgcn_alignbit(h, h, 19), weight, sat)) ctl.overflow = 1;
                    mine += weight;
                }
            }
        };
        // records are requested two batches ahead of their use (one workgroup per CU: nothing else hides HBM latency)
        Rec<RW> nxt, nxt2;
#pragma unroll
        for (int o = 0; o < RW; o++) { nxt.w[o] = 0; nxt2.w[o] = 0; }
        uint32_t w_nxt, w_nxt2;
        uint32_t have_nxt = fetch(w_begin + (uint32_t)lane, nxt, w_nxt);
        uint32_t have_nxt2 = fetch(w_begin + 64u + (uint32_t)lane, nxt2, w_nxt2);
        if (threadIdx.x == 0) ctl.prog_den = R ? R : 1u;
        for (uint32_t r0 = w_begin; r0 < (SHK_DBG(rvw.dbg) == 6 ? w_begin : w_end); r0 += 64) {
            // the table filled up: stop early (every insert into a full table walks a long probe chain);
            // how far this round got (all waves advance alike) sizes the split
            if (ctl.overflow || ctl.n_used > (S / 10) * 9) { atomicMax(&ctl.prog_num, min(R, (r0 - w_begin) * (COUNT_THREADS / 64) + 1u)); break; }
            Rec<RW> rec = nxt;
            const uint32_t n = have_nxt ? (uint32_t)(rec.w[RW - 1] >> 58) + 1u : 0u;
            // one spelling per locus (pass 1 is bound by its instruction issue, this pass by latencies: done here)
            if (n && SHK_DBG(rvw.dbg) != 11) rec_canonicalise<RW>(rec.w, n, k);
            const uint32_t wgt = w_nxt;
            nxt = nxt2; have_nxt = have_nxt2; w_nxt = w_nxt2;
            have_nxt2 = fetch(r0 + 128u + (uint32_t)lane, nxt2, w_nxt2);
            // phase A: identical records (the same genomic run seen in many reads) are counted
            // once here and expanded once, with their multiplicity, in phase B
            if (SHK_DBG(rvw.dbg) == 4) { if (n && rec.w[0] == 0x123456789ull) ctl.overflow = 1; }   // timing experiment: fetch only
            else if (SHK_DBG(rvw.dbg) == 2) { if (n) expand(rec, n, wgt); }
            else if (SHK_DBG(rvw.dbg) == 7) { if (n) (void)rec_insert<W>(tb.rt, ctl, rec, wgt); }      // timing experiment: dedupe only
            else if (n && !rec_insert<W>(tb.rt, ctl, rec, wgt)) {
                // the record table is saturated.  If that happens in the first half of the records, most of
                // them are unique (error-rich reads): their k-mers cannot fit the k-mer table either, so the
                // round is given up at once when the k-mer-level repartition can take over (always correct)
                if (ovf && mod == 1 && (r0 - w_begin) * 2u < (w_end - w_begin)) ctl.overflow = 1;
                else expand(rec, n, wgt);
            }
        }
        if (SHK_DBG(rvw.dbg) == 10) return;                          // timing experiment: phase A only, no phase B
        {
            // phase B: every distinct record once, weighted.  The occupied slots are first listed in
            // order of record length (counting sort in LDS) so that the 64 records a wave expands
            // together have (nearly) the same number of k-mers: no lanes idling behind the longest.
            constexpr uint32_t SRc = decltype(tb.rt)::SR;
            auto &rt = tb.rt;
            if (threadIdx.x < 64) rt.nhist[threadIdx.x] = 0;
            __syncthreads();
            if (SHK_DBG(rvw.dbg) != 3)
            for (uint32_t s = threadIdx.x; s < SRc; s += COUNT_THREADS)
                if (rt.rst[s] >= 3u) atomicAdd(&rt.nhist[(uint32_t)(rt.w[s][RW - 1] >> 58)], 1u);
            __syncthreads();
            if (threadIdx.x < 64) {
                const uint32_t v = rt.nhist[threadIdx.x];
                uint32_t incl = v;
                for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o); if (lane >= o) incl += u; }
                rt.nbase[threadIdx.x] = incl - v;
                if (threadIdx.x == 63) ctl.n_recs = incl;
            }
            __syncthreads();
            if (SHK_DBG(rvw.dbg) != 3)
            for (uint32_t s = threadIdx.x; s < SRc; s += COUNT_THREADS)
                if (rt.rst[s] >= 3u) rt.order[atomicAdd(&rt.nbase[(uint32_t)(rt.w[s][RW - 1] >> 58)], 1u)] = (uint16_t)s;
            __syncthreads();
            uint32_t n_recs = ctl.n_recs;
            if (SHK_DBG(rvw.dbg) == 3) n_recs = SRc;                 // timing experiment: slot order instead of length order
            // (a table that filled up in phase A keeps its progress mark; phase B then stops at once)
            if (threadIdx.x == 0 && ctl.prog_num == 0) ctl.prog_den = n_recs ? n_recs : 1u;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n_recs; i += COUNT_THREADS) {
                if (ctl.overflow || ctl.n_used > (S / 10) * 9) { atomicMax(&ctl.prog_num, i - threadIdx.x + 1u); break; }
                const uint32_t s = SHK_DBG(rvw.dbg) == 3 ? i : rt.order[i];
                if (SHK_DBG(rvw.dbg) == 3 && rt.rst[s] < 3u) continue;
                Rec<RW> rec;
#pragma unroll
                for (int o = 0; o < RW; o++) rec.w[o] = rt.w[s][o];
                if (SHK_DBG(rvw.dbg) != 1) expand(rec, (uint32_t)(rec.w[RW - 1] >> 58) + 1u, rt.rst[s] - 2u);
            }
        }
        __syncthreads();
        const bool over = ctl.overflow != 0 || ctl.n_used > (S / 10) * 9;
        if (SHK_DBG(rvw.dbg) == 8) return;                           // timing experiment: no defer / emit work at all
        if (over && ovf && mod == 1) {
            if (SHK_DBG(rvw.dbg) == 9) return;                       // timing experiment: overflow detected, nothing reported
            defer(mine, true);
            return;
        }
        if (over) {
            // split this residue class (results of other classes are unaffected).  The table held
            // n_used keys after prog_num of prog_den records: aim the children at ~60 % of the table.
            if (mod >= 4096 || ctl.sp + 1 > 16) { if (threadIdx.x == 0) flags[0] = 1; break; }
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t num = ctl.prog_num ? ctl.prog_num : ctl.prog_den;
                const double est = (double)ctl.n_used * (double)ctl.prog_den / (double)num;
                uint32_t factor = 2;
                while ((double)factor * (0.6 * S) < est && mod * factor < 4096u) factor <<= 1;
                ctl.st_res[ctl.sp] = res; ctl.st_step[ctl.sp] = mod; ctl.st_factor[ctl.sp] = factor; ctl.st_next[ctl.sp] = 0;
                ctl.sp += 1;
            }
            continue;
        }
        // (phase B is over: the record table's multiplicity words are free to hold the emit list)
        static_assert(2u * decltype(tb.rt)::SR >= S, "emit list");
        table_emit<W>(tb, ctl, mine, threshold, histo, out_keys, out_cnt, out_cap, out_cursor, reinterpret_cast<uint16_t *>(tb.rt.rst), 0u, whist);
    }
    if (threadIdx.x == 0 && ovf && defer_after) atomicAdd(&ovf_n[1], 1u);                          // tried, and it fitted
   }();
  }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < 500; b += COUNT_THREADS)
        if (whist[b]) atomicAdd(&histo[b], (unsigned long long)whist[b]);
    if (threadIdx.x == 0 && ctl.n_inst) atomicAdd(n_inst, ctl.n_inst);
}


// ---- records deduplicated in a kernel of their own ------------------------------------------------------------------
// (1) Sharded counting: at 100x an isolate's super-k-mer records recur ~50 times, and they are what the shard layer exchanges
// (16W bytes each: 6x the packed reads they were cut from) — phase A of pass 2 runs on the SENDER, before the fabric.
// (2) Single GPU, one- and two-word keys: phase A and the expansion (k_count_weighted) as two kernels that need half of a
// CU's LDS each instead of one that owns it all.
// Out go the distinct records of partition p (sorted by length) at out_recs[base[p] ..), their multiplicities at
// out_w[base[p] ..), their number at n_out[p]; base[] leaves every partition room for all its raw records.  A record
// that finds the table saturated is passed through with weight 1.  Persistent workgroups over the partitions
// [p_first, n_parts), handed out by a work counter.
// ovf != nullptr (single GPU): reads with errors do not deduplicate and their partitions do not fit the k-mer table of
// the counting kernel.  The partitions tried here are the sample: each reports "tried" and, if it ended with more distinct
// records than the k-mer table has room for keys (kmer_cap) or lost less than half of its records as duplicates, "handed
// over" (ovf_n[1]: tried | handed over << 16; n_out[p] = 0 and an entry in ovf[]: the k-mer-level repartition counts it
// from its raw records).  The report returns the tally: once >= defer_after were handed over and they are at least HALF of
// those tried, the workgroup only reads its further partitions for their k-mer counts and reports them "not tried".
// (HALF, not the 3/4 k_count_partitions asks of its sample of whole counts: the records' repeats are the weaker sign — with
// 1 % errors left in 63 % of the partitions show it, and every one of them overflows the k-mer table.)
// Which partitions are tried depends on timing; the counts never do.
// (rounds 2–3: a sample of 256 partitions went through k_count_partitions in front of this kernel, 74 us of a 0.74 ms pass 2
// on clean reads.  Measured on the GPU box, alternating runs of bench.py: pass 2 0.75 -> 0.68 ms, configs[2] with its errors
// left in 10.7 -> 10.5 ms.)
template <int W> struct DedupeShared {
    RecTable<W, (W == 1 ? 3072u : ((69632u / (16u * W + 6u)) & ~63u))> rt;   // 3072 / 1792 / 1280 / 960 records: two workgroups share a CU's LDS
};
template <int W>
__global__ __launch_bounds__(COUNT_THREADS, W <= 2 ? 8 : 4) void k_dedupe_partitions(   // one- and two-word keys: 8 waves per SIMD = two workgroups per CU, <= 64 VGPRs
                                                                     RunView rvw, uint32_t p_first, uint32_t n_parts,
                                                                     const unsigned long long *__restrict__ base,
                                                                     uint64_t *__restrict__ out_recs, uint32_t *__restrict__ out_w,
                                                                     uint32_t *__restrict__ n_out, uint32_t *__restrict__ work_counter,
                                                                     OvfRec *__restrict__ ovf /* nullable */, uint32_t *__restrict__ ovf_n,
                                                                     uint32_t defer_after, uint32_t kmer_cap,
                                                                     uint32_t *__restrict__ verdict_out /* nullable: k_count_weighted's tally word — the verdict is passed on */) {
    constexpr int RW = 2 * W;
    __shared__ DedupeShared<W> tb;
    __shared__ CountCtl ctl;
    __shared__ uint32_t wtot[4];
    constexpr uint32_t SR = decltype(tb.rt)::SR;
    const int lane = threadIdx.x & 63;
    const int k = rvw.k;
    const uint32_t S_runs = rvw.S;
    const uint32_t n_here = n_parts - p_first;
    __shared__ uint32_t forced_sh;
    if (threadIdx.x == 0) forced_sh = 0;
    __syncthreads();
    bool forced = false;
    uint32_t pi_next = 0;
    for (uint32_t pi = blockIdx.x; pi < n_here; pi = pi_next) {
        const uint32_t p = p_first + pi;
        uint32_t f = 0; unsigned long long addr = 0;
        if (threadIdx.x < S_runs) { f = rvw.run_cnt[(uint64_t)p * S_runs + threadIdx.x]; addr = rvw.run_addr16[(uint64_t)p * S_runs + threadIdx.x]; }
        forced = forced_sh != 0;                         // (written behind a barrier of the previous partition, see below)
        if (forced) {
            // error-rich reads: no dedupe — a thread per run adds up the k-mer counts of its records, the partition is handed over
            unsigned long long inst = 0;
            const uint64_t *src = reinterpret_cast<const uint64_t *>(addr << 4);
            for (uint32_t i = 0; i < f; i++) inst += (src[(uint64_t)i * RW + RW - 1] >> 58) + 1ull;
            __syncthreads();                             // (the previous partition's report is out)
            if (threadIdx.x == 0) { ctl.part_inst = 0; ctl.next_pi = gridDim.x + atomicAdd(work_counter, 1u); }
            __syncthreads();
            for (int o = 32; o > 0; o >>= 1) inst += __shfl_down(inst, o);
            if (lane == 0 && inst) atomicAdd(&ctl.part_inst, inst);
            __syncthreads();
            pi_next = ctl.next_pi;
            if (threadIdx.x == 0) {
                n_out[p] = 0;
                if (ctl.part_inst) { const uint32_t slot = atomicAdd(ovf_n, 1u); OvfRec o; o.p = p; o.est_distinct = 0u; o.instances = ctl.part_inst; ovf[slot] = o; }
            }
            continue;
        }
        uint32_t incl = f;
        if (threadIdx.x < 256) {
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
            if (lane == 63) wtot[threadIdx.x >> 6] = incl;
        }
        __syncthreads();                                 // (also: the previous partition is done with ctl / the table)
        if (threadIdx.x < 256) {
            uint32_t off = 0;
            for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) off += wtot[w];
            if (threadIdx.x < S_runs) { ctl.pre[threadIdx.x] = off + incl - f; ctl.roff[threadIdx.x] = addr; }
        }
        if (threadIdx.x == 0) {
            ctl.pre[S_runs] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            ctl.rec_used = 0; ctl.n_emit = 0; ctl.part_inst = 0;
            ctl.next_pi = gridDim.x + atomicAdd(work_counter, 1u);
        }
        for (uint32_t s = threadIdx.x; s < SR; s += COUNT_THREADS) tb.rt.rst[s] = 0;
        __syncthreads();
        pi_next = ctl.next_pi;
        const uint32_t R = ctl.pre[S_runs];
        // the partition goes to the k-mer-level repartition: its k-mer instances (one more pass over its records), n_out = 0
        auto hand_over = [&]() {
            unsigned long long inst = 0;
            // (every wave a contiguous share of the records, its run cursor only moves forward: one search per wave)
            const uint32_t pw = (((R + (COUNT_THREADS / 64) - 1) / (COUNT_THREADS / 64)) + 63u) & ~63u;
            const uint32_t wb = min(R, (threadIdx.x >> 6) * pw), we = min(R, wb + pw);
            uint32_t lo = 0, hi = S_runs;
            { const uint32_t r = min(wb + (uint32_t)lane, R ? R - 1u : 0u);
              while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ctl.pre[mid] <= r) lo = mid; else hi = mid; } }
            uint32_t nx = ctl.pre[lo + 1];
            for (uint32_t r = wb + (uint32_t)lane; r < we; r += 64) {
                while (r >= nx) { lo++; nx = ctl.pre[lo + 1]; }
                const uint64_t *src = reinterpret_cast<const uint64_t *>(ctl.roff[lo] << 4) + (uint64_t)(r - ctl.pre[lo]) * RW;
                inst += (src[RW - 1] >> 58) + 1ull;
            }
            for (int o = 32; o > 0; o >>= 1) inst += __shfl_down(inst, o);
            if (lane == 0 && inst) atomicAdd(&ctl.part_inst, inst);
            __syncthreads();
            if (threadIdx.x == 0) {
                n_out[p] = 0;
                if (R) { const uint32_t slot = atomicAdd(ovf_n, 1u); OvfRec o; o.p = p; o.est_distinct = 0u; o.instances = ctl.part_inst; ovf[slot] = o; }
            }
        };
        uint64_t *dst = out_recs + base[p] * RW;
        uint32_t *dst_w = out_w + base[p];
        const uint32_t per_wave = (((R + (COUNT_THREADS / 64) - 1) / (COUNT_THREADS / 64)) + 63u) & ~63u;
        const uint32_t w_begin = min(R, (threadIdx.x >> 6) * per_wave), w_end = min(R, w_begin + per_wave);
        uint32_t run_lo = 0, run_next = 0;
        {
            uint32_t lo = 0, hi = S_runs;
            const uint32_t r = min(w_begin + (uint32_t)lane, R ? R - 1u : 0u);
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ctl.pre[mid] <= r) lo = mid; else hi = mid; }
            run_lo = lo; run_next = ctl.pre[lo + 1];
        }
        auto fetch = [&](uint32_t r, Rec<RW> &rec) -> uint32_t {
            if (r >= w_end) return 0u;
            while (r >= run_next) { run_lo++; run_next = ctl.pre[run_lo + 1]; }
            const uint64_t *src = reinterpret_cast<const uint64_t *>(ctl.roff[run_lo] << 4) + (uint64_t)(r - ctl.pre[run_lo]) * RW;
#pragma unroll
            for (int o = 0; o < RW; o += 2) {
                const ulonglong2 v2 = *reinterpret_cast<const ulonglong2 *>(src + o);
                rec.w[o] = v2.x; rec.w[o + 1] = v2.y;
            }
            return 1u;
        };
        Rec<RW> nxt, nxt2;
#pragma unroll
        for (int o = 0; o < RW; o++) { nxt.w[o] = 0; nxt2.w[o] = 0; }
        uint32_t have_nxt = fetch(w_begin + (uint32_t)lane, nxt);
        uint32_t have_nxt2 = fetch(w_begin + 64u + (uint32_t)lane, nxt2);
        for (uint32_t r0 = w_begin; r0 < w_end; r0 += 64) {
            Rec<RW> rec = nxt;
            const uint32_t n = have_nxt ? (uint32_t)(rec.w[RW - 1] >> 58) + 1u : 0u;
            if (n && SHK_DBG(rvw.dbg) != 21) rec_canonicalise<RW>(rec.w, n, k);
            nxt = nxt2; have_nxt = have_nxt2;
            have_nxt2 = fetch(r0 + 128u + (uint32_t)lane, nxt2);
            if (SHK_DBG(rvw.dbg) == 21 || SHK_DBG(rvw.dbg) == 22) { if (n && rec.w[0] == 0x123456789ull) ctl.n_emit = 1; continue; }   // timing experiments: fetch (+ canonical spelling) only
            // (the table is saturated and holds no copy within a few probes: passed through, once)
            const bool pass = n && !rec_insert<W>(tb.rt, ctl, rec);
            const unsigned long long m = __ballot(pass);
            if (m) {
                uint32_t at = 0;
                if (lane == (int)__builtin_ctzll(m)) at = atomicAdd(&ctl.n_emit, (uint32_t)__popcll(m));
                at = __shfl(at, (int)__builtin_ctzll(m)) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (pass) { part_store_record<RW>(rec.w, dst + (uint64_t)at * RW); dst_w[at] = 1u; }
            }
        }
        // the table's records leave in order of their LENGTH (counting sort in LDS): the 64 records a wave of the counting
        // kernel expands together then have (nearly) the same number of k-mers — no lanes idling behind the longest
        auto &rt = tb.rt;
        if (threadIdx.x < 64) rt.nhist[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < SR; s += COUNT_THREADS)
            if (rt.rst[s] >= 3u) atomicAdd(&rt.nhist[(uint32_t)(rt.w[s][RW - 1] >> 58)], 1u);
        __syncthreads();
        const uint32_t n_pass = ctl.n_emit;                   // (records passed through above)
        if (threadIdx.x < 64) {
            const uint32_t v = rt.nhist[threadIdx.x];
            uint32_t incl2 = v;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl2, o); if (lane >= o) incl2 += u; }
            rt.nbase[threadIdx.x] = n_pass + incl2 - v;
            if (threadIdx.x == 63) ctl.n_recs = n_pass + incl2;
        }
        __syncthreads();
        // (fewer than half of the records were duplicates: the partition count gives a partition ~20 table sizes of k-mer
        // instances, so reads that repeat that little bring far more distinct k-mers than the table holds)
        const bool hand = ovf && (ctl.n_recs > kmer_cap || (R > kmer_cap / 4u && 2u * ctl.n_recs > R));     // (uniform)
        if (ovf && threadIdx.x == 0 && R) {
            // (the tally comes back with this partition's report: the verdict for this workgroup's next partition — read at the
            // top of the loop, behind the barrier in hand_over() or the one in front of n_out[p] below)
            const uint32_t add = hand ? 0x10001u : 1u;
            const uint32_t x = atomicAdd(&ovf_n[1], add) + add;
            const uint32_t n_over = x >> 16, n_tried = x & 0xFFFFu;
            if (defer_after && n_over >= defer_after && 2u * n_over >= n_tried) {
                forced_sh = 1u;
                // (k_count_weighted starts handing over at once: the partitions that did pass the dedupe before the verdict
                // would fail there one by one, each after filling the table — 1.1 ms of 11 on configs[2] with its errors left in)
                if (verdict_out) atomicOr(verdict_out, 0x80000000u);
            }
        }
        if (hand) { hand_over(); continue; }
        for (uint32_t s = threadIdx.x; s < (SHK_DBG(rvw.dbg) == 23 ? 0u : SR); s += COUNT_THREADS) {     // (23: timing experiment, nothing written)
            const uint32_t st = rt.rst[s];
            if (st < 3u) continue;
            const uint32_t at = atomicAdd(&rt.nbase[(uint32_t)(rt.w[s][RW - 1] >> 58)], 1u);
            uint64_t r[RW];
#pragma unroll
            for (int o = 0; o < RW; o++) r[o] = rt.w[s][o];
            part_store_record<RW>(r, dst + (uint64_t)at * RW);
            dst_w[at] = st - 2u;
        }
        __syncthreads();
        if (threadIdx.x == 0) n_out[p] = ctl.n_recs;
    }
}

// Pass 2 over DEDUPLICATED records (k_dedupe_partitions wrote them: distinct records of partition p at recs[base[p] ..),
// multiplicities beside them, sorted by length): every record expanded once, every k-mer added with the record's weight.
// Needs the k-mer table alone: TWO workgroups per CU (one-word and two-word keys), so the latencies of one — the fetch of
// its records, the barriers, the row reservation's round trip — hide behind the inserts of the other.  k_count_partitions
// does both steps in one workgroup that owns the whole LDS, and every latency in it is exposed (derived VALUBusy 36 %).
// A partition whose distinct k-mers do not fit is reported in ovf[] (exact instance count, estimated distinct k-mers)
// and counted from its RAW records by the k-mer-level repartition, exactly as k_count_partitions reports it; the same
// in-launch sample decides when later partitions are handed over untried.
template <int W>
__global__ __launch_bounds__(COUNT_THREADS, W <= 2 ? 8 : 4) void k_count_weighted(
    const uint64_t *__restrict__ recs, const uint32_t *__restrict__ weights, const unsigned long long *__restrict__ base,
    const uint32_t *__restrict__ n_recs, uint32_t p_first, uint32_t n_parts, uint32_t merge /* partitions counted together in one table (1..4) */, int k, uint32_t threshold,
    unsigned long long *__restrict__ histo, KeyArr<W> out_keys, uint32_t *__restrict__ out_cnt,
    unsigned long long out_cap, unsigned long long *__restrict__ out_cursor,
    unsigned long long *__restrict__ n_inst, OvfRec *__restrict__ ovf, uint32_t *__restrict__ ovf_n,
    unsigned long long *__restrict__ work_and_tally /* low word: groups handed out, zero at launch; high word: this kernel's tally, tried |
                                                       overflowed << 16, which k_dedupe_partitions starts at its verdict — ONE atomic fetches both */,
    uint32_t probe_groups, uint32_t defer_after, uint32_t dbg_arg /* timing experiments (ABLATE builds) */) {
    constexpr int RW = 2 * W;
    constexpr uint32_t S = KmerTable<W>::S;
    const uint32_t dbg = SHK_DBG(dbg_arg);
    __shared__ KmerTable<W> tb;
    __shared__ CountCtlCore ctl;
    constexpr uint32_t ELIST = W == 1 ? S : 64;         // emit list: W >= 2 reuses the state words
    __shared__ uint16_t elist[ELIST];
    __shared__ uint32_t whist[500];
    __shared__ uint32_t next_g_sh, force_sh;
    const int lane = threadIdx.x & 63;
    for (uint32_t b = threadIdx.x; b < 500; b += COUNT_THREADS) whist[b] = 0;
    if (threadIdx.x == 0) ctl.n_inst = 0;
    // (group g = partitions g, g + n_groups, g + 2 n_groups ... — NOT neighbours: the rows of a group leave the table mixed, and
    // the graph and collapse kernels rely on neighbouring rows sharing the low bits of their minimiser hash (k_row_starts);
    // partitions a multiple of n_groups apart do.  Members below p_first were counted by the sample kernel.)
    const uint32_t n_groups = (n_parts + merge - 1) / merge;
    // The distinct k-mers of a partition load the table to ~a quarter (the partition count is sized for the RECORD table
    // of the dedupe and for reads with errors), and every partition costs fixed latencies — table reset, the fetch of
    // its first records, barriers, the row reservation's round trip: `merge` neighbouring partitions are counted in one
    // table and emitted together (their rows stay side by side).  A group that does not fit is counted partition by
    // partition; a partition that does not fit alone goes to the k-mer-level repartition.
    unsigned long long mine = 0;
    // counts the records of the partitions pm[0 .. nm) into the (reset) table; true = they fitted
    auto count_range = [&](const uint32_t (&pm)[4], uint32_t nm) -> bool {
        uint32_t Rj[4]; unsigned long long bj[4]; uint32_t Rt = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool in = (uint32_t)j < nm;
            Rj[j] = in ? n_recs[pm[j]] : 0u; bj[j] = in ? base[pm[j]] : 0ull;
            Rt += Rj[j];
        }
        auto fetch = [&](uint32_t r, Rec<RW> &rec, uint32_t &wgt) {
            if (r < Rt) {
                unsigned long long at = bj[0] + r;
                uint32_t pre = Rj[0];
#pragma unroll
                for (int j = 1; j < 4; j++) { if (r >= pre) at = bj[j] + (r - pre); pre += Rj[j]; }
                const uint64_t *src = recs + at * RW;
#pragma unroll
                for (int o = 0; o < RW; o += 2) {
                    const ulonglong2 v2 = *reinterpret_cast<const ulonglong2 *>(src + o);
                    rec.w[o] = v2.x; rec.w[o + 1] = v2.y;
                }
                wgt = weights[at];
            }
        };
        Rec<RW> nxt; uint32_t w_nxt = 0;
#pragma unroll
        for (int o = 0; o < RW; o++) nxt.w[o] = 0;
        fetch(threadIdx.x, nxt, w_nxt);                                  // in flight during the reset below
        __syncthreads();                                                 // the previous table's rows are out
        if (threadIdx.x == 0) {
            ctl.n_used = 0; ctl.overflow = 0; ctl.n_emit = 0; ctl.wave_cursor = 0; ctl.rec_used = 0;
            ctl.prog_num = 0; ctl.prog_den = Rt ? Rt : 1u;
        }
        kmer_table_reset<W>(tb, ctl);                                    // (ends with a barrier)
        mine = 0;
        if (dbg != 31)
        for (uint32_t r0 = 0; r0 < Rt; r0 += COUNT_THREADS) {
            if (ctl.overflow || ctl.n_used > (S / 10) * 9) break;
            Rec<RW> rec = nxt; const uint32_t weight = w_nxt;
            const uint32_t r = r0 + threadIdx.x;
            fetch(r + COUNT_THREADS, nxt, w_nxt);
            if (r >= Rt) continue;
            const uint32_t n = (uint32_t)(rec.w[RW - 1] >> 58) + 1u;
            Kmer<W> f = km_zero<W>();
            for (uint32_t s = 0; s < n; s++) {
                if (ctl.overflow || ctl.n_used > (S / 10) * 9) break;    // the round is lost: do not walk full-table probe chains
                if (s) {
#pragma unroll
                    for (int o = 0; o < RW - 1; o++) rec.w[o] = (rec.w[o] >> 2) | (rec.w[o + 1] << 62);
                    rec.w[RW - 1] >>= 2;
                }
                Kmer<W> rv;
#pragma unroll
                for (int j = 0; j < W; j++) rv.w[j] = ~rec.w[j];
                rv.w[W - 1] &= km_topmask<W>(k);
                if (s) km_push_back<W>(f, (uint32_t)(rec.w[W - 1] >> ((2 * (k - 1)) & 63)) & 3u, k);
                else f = km_revcomp<W>(rv, k);
                const bool use_r = km_less<W>(rv, f);
                Kmer<W> c;
#pragma unroll
                for (int j = 0; j < W; j++) c.w[j] = use_r ? rv.w[j] : f.w[j];
                const uint32_t h = km_mix32<W>(c);
                if (dbg == 33) { if (h == 0x12345u && c.w[0] == 77ull) ctl.overflow = 1; }       // timing experiment: roll + hash, no insert
                else if (!lds_insert<W>(tb, ctl, c, h ^ __builtin_amdgcn_alignbit(h, h, 19), weight, true)) ctl.overflow = 1;
                mine += weight;
            }
        }
        __syncthreads();
        return !(ctl.overflow != 0 || ctl.n_used > (S / 10) * 9);
    };
    auto emit = [&]() {
        if (dbg == 32) return;                                           // timing experiment: no emit
        if constexpr (W == 1) table_emit<W>(tb, ctl, mine, threshold, histo, out_keys, out_cnt, out_cap, out_cursor, elist, 0u, whist);
        else table_emit<W>(tb, ctl, mine, threshold, histo, out_keys, out_cnt, out_cap, out_cursor, tb.state, 0u, whist);
    };
    // hand partition p to the k-mer-level repartition: exact instances, estimated distinct k-mers (of the attempt just made)
    uint32_t *const tally = reinterpret_cast<uint32_t *>(work_and_tally) + 1;
    bool in_sample = true;                               // (this workgroup's first group: see the loop below)
    auto defer = [&](uint32_t p, bool was_tried) {
        const uint32_t R = n_recs[p];
        const uint64_t *src = recs + base[p] * RW;
        const uint32_t *src_w = weights + base[p];
        __syncthreads();
        if (threadIdx.x == 0) { ctl.tried = 0; ctl.part_inst = 0; }
        __syncthreads();
        unsigned long long m = was_tried ? mine : 0ull;
        for (int o = 32; o > 0; o >>= 1) m += __shfl_down(m, o);
        if (lane == 0 && m) atomicAdd(&ctl.tried, m);
        unsigned long long inst = 0;
        for (uint32_t r = threadIdx.x; r < R; r += COUNT_THREADS)
            inst += ((src[(uint64_t)r * RW + RW - 1] >> 58) + 1ull) * (unsigned long long)src_w[r];
        for (int o = 32; o > 0; o >>= 1) inst += __shfl_down(inst, o);
        if (lane == 0 && inst) atomicAdd(&ctl.part_inst, inst);
        __syncthreads();
        if (threadIdx.x == 0 && R) {
            const double est = ctl.tried ? (double)ctl.n_used * (double)ctl.part_inst / (double)ctl.tried : (double)ctl.part_inst;
            const uint32_t slot = atomicAdd(ovf_n, 1u);
            OvfRec o; o.p = p; o.est_distinct = !was_tried ? 0u /* not tried */ : est > 4.0e9 ? 0xFFFFFFFFu : est < 1.0 ? 1u : (uint32_t)est; o.instances = ctl.part_inst;
            ovf[slot] = o;
        }
        if (threadIdx.x == 0 && was_tried && defer_after && in_sample) atomicAdd(tally, 0x10001u);         // tried, and it overflowed
    };
    // (the tally is kept by every workgroup's FIRST group only — they run side by side, a sample of 2 x CUs groups: an atomic per
    // group on one word, 8192 of them, cost 35 us of the 0.35 ms this kernel takes on clean reads: thread 0 waits for it to
    // retire before it sees its next group's number)
    uint32_t g_next = 0;
    for (uint32_t g = blockIdx.x; g < n_groups; in_sample = false, g = g_next) {
        uint32_t pm[4] = {0, 0, 0, 0}, nm = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t pj = g + j * n_groups;
            const bool v = j < merge && pj < n_parts && pj >= p_first;
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) if (v && nm == q) pm[q] = pj;             // (register array: selects, no dynamic indexing)
            nm += v ? 1u : 0u;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long wt = atomicAdd(work_and_tally, 1ull);       // (the next group's number and the tally in one round trip)
            const uint32_t x = (uint32_t)(wt >> 32), n_over = x >> 16, n_tried = x & 0xFFFFu;
            next_g_sh = gridDim.x + (uint32_t)wt;
            force_sh = (defer_after && g >= probe_groups && n_over >= defer_after && 4u * n_over >= 3u * n_tried) ? 1u : 0u;
        }
        __syncthreads();
        g_next = next_g_sh;
        if (nm == 0) continue;
        if (force_sh) {
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) if (j < nm) defer(pm[j], false);
            continue;
        }
        if (nm > 1 && count_range(pm, nm)) {
            emit();
            if (threadIdx.x == 0 && defer_after && in_sample) atomicAdd(tally, nm);                        // tried, and they fitted
            continue;
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            if (j >= nm) continue;
            const uint32_t one[4] = {pm[j], 0, 0, 0};
            if (count_range(one, 1u)) { emit(); if (threadIdx.x == 0 && defer_after && in_sample) atomicAdd(tally, 1u); }
            else defer(pm[j], true);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < 500; b += COUNT_THREADS)
        if (whist[b]) atomicAdd(&histo[b], (unsigned long long)whist[b]);
    if (threadIdx.x == 0 && ctl.n_inst) atomicAdd(n_inst, ctl.n_inst);
}

// the deduplicated records of partition p (one contiguous run at src[src_base[p] ..)) and their weights go to their place
// in the destination-major send buffers
template <int RW>
__global__ __launch_bounds__(256) void k_pack_dedup(const uint64_t *__restrict__ src, const uint32_t *__restrict__ src_w,
                                                    const unsigned long long *__restrict__ src_base, const uint32_t *__restrict__ n_out,
                                                    const unsigned long long *__restrict__ dst_base,
                                                    uint64_t *__restrict__ dst, uint32_t *__restrict__ dst_w) {
    const uint32_t p = blockIdx.x, R = n_out[p];
    const uint64_t *sp = src + src_base[p] * RW;
    uint64_t *dp = dst + dst_base[p] * RW;
    for (uint32_t r = threadIdx.x; r < R; r += blockDim.x) {
#pragma unroll
        for (int o = 0; o < RW; o += 2)
            *reinterpret_cast<ulonglong2 *>(dp + (uint64_t)r * RW + o) = *reinterpret_cast<const ulonglong2 *>(sp + (uint64_t)r * RW + o);
        dst_w[dst_base[p] + r] = src_w[src_base[p] + r];
    }
}

// k-mer-level repartition of one overflowed partition: every canonical k-mer of its records goes to
// bucket (hash bits 12..) of the item's region; cursors live in LDS, the fills are published at the end
// BLOOM (do_bloom, docs/src/assembly.md:18: "allowing for some degree of overcounting"): a blocked Bloom filter over
// the partition's k-mers lives in LDS — one 32-bit word per k-mer, two bits in it, set and tested by ONE atomic OR,
// so of several simultaneous first sightings of a k-mer exactly one sees "new".  A k-mer sighted as new only sets
// its bits; every later sighting goes to the buckets as before.  Singletons — most of what an error-rich read set
// holds — therefore never reach HBM or a table; the counts of the others lack their first sighting, which the
// emit adds back (table_emit's bias).  A false positive sends a first sighting to the buckets too: that k-mer is
// counted one too high — over, never under.  new_count[item] = sightings taken as new (distinct k-mers, less the
// false positives).
static constexpr uint32_t BLOOM_WORDS = 24576;           // 96 KB of LDS: 786 k bits per partition
template <int W, bool BLOOM>
__global__ __launch_bounds__(COUNT_THREADS) void k_ovf_scatter(RunView rvw, const OvfItem *__restrict__ items,
                                                               uint64_t *__restrict__ kmers,
                                                               uint32_t *__restrict__ bucket_fill,
                                                               uint32_t *__restrict__ new_count) {
    constexpr int RW = 2 * W;
    __shared__ uint32_t pre[257];
    __shared__ uint32_t cursor[OVF_MAX_F];
    __shared__ uint32_t bloom[BLOOM ? BLOOM_WORDS : 1];
    __shared__ uint32_t n_new;
    if constexpr (BLOOM) {
        for (uint32_t i = threadIdx.x; i < BLOOM_WORDS; i += COUNT_THREADS) bloom[i] = 0;
        if (threadIdx.x == 0) n_new = 0;
    }
    uint32_t my_new = 0;
    const OvfItem it = items[blockIdx.x];
    const uint32_t p = it.p, S_runs = rvw.S;
    const int lane = threadIdx.x & 63;
    const int k = rvw.k;
    if (threadIdx.x < 64) {
        uint32_t run = 0;
        for (uint32_t g0 = 0; g0 < S_runs; g0 += 64) {
            const uint32_t g = g0 + threadIdx.x;
            uint32_t f = g < S_runs ? rvw.run_cnt[(uint64_t)p * S_runs + g] : 0u;
            uint32_t incl = f;
            for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
            if (g < S_runs) pre[g] = run + incl - f;
            run += __shfl(incl, 63);
        }
        if (threadIdx.x == 0) pre[S_runs] = run;
    }
    for (uint32_t b = threadIdx.x; b < OVF_MAX_F; b += COUNT_THREADS) cursor[b] = 0;
    __syncthreads();
    const uint32_t R = pre[S_runs];
    const uint32_t F = it.F;
    for (uint32_t r = threadIdx.x; r < R; r += COUNT_THREADS) {
        uint32_t lo = 0, hi = S_runs;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= r) lo = mid; else hi = mid; }
        const uint64_t *src = reinterpret_cast<const uint64_t *>(rvw.run_addr16[(uint64_t)p * S_runs + lo] << 4) + (uint64_t)(r - pre[lo]) * RW;
        Rec<RW> rec;
#pragma unroll
        for (int o = 0; o < RW; o += 2) {
            const ulonglong2 v2 = *reinterpret_cast<const ulonglong2 *>(src + o);
            rec.w[o] = v2.x; rec.w[o + 1] = v2.y;
        }
        const uint32_t n = (uint32_t)(rec.w[RW - 1] >> 58) + 1u;
        Kmer<W> f = km_zero<W>();
        for (uint32_t s = 0; s < n; s++) {
            if (s) {
#pragma unroll
                for (int o = 0; o < RW - 1; o++) rec.w[o] = (rec.w[o] >> 2) | (rec.w[o + 1] << 62);
                rec.w[RW - 1] >>= 2;
            }
            Kmer<W> rv;
#pragma unroll
            for (int j = 0; j < W; j++) rv.w[j] = ~rec.w[j];
            rv.w[W - 1] &= km_topmask<W>(k);
            if (s) km_push_back<W>(f, (uint32_t)(rec.w[W - 1] >> ((2 * (k - 1)) & 63)) & 3u, k);
            else f = km_revcomp<W>(rv, k);
            const bool use_r = km_less<W>(rv, f);
            Kmer<W> c;
#pragma unroll
            for (int j = 0; j < W; j++) c.w[j] = use_r ? rv.w[j] : f.w[j];
            const uint32_t hk = km_mix32<W>(c);
            if constexpr (BLOOM) {
                const uint32_t h2 = mix32(hk ^ 0xC2B2AE35u);
                const uint32_t word = (uint32_t)(((uint64_t)h2 * BLOOM_WORDS) >> 32);
                const uint32_t b0 = hk & 31u, b1 = (hk >> 5) & 31u;
                const uint32_t bits = (1u << b0) | (1u << (b1 == b0 ? (b0 + 1u) & 31u : b1));
                const uint32_t old = atomicOr(&bloom[word], bits);
                if ((old & bits) != bits) { my_new++; continue; }        // first sighting: only the filter learns of it
            }
            // (any F: multiply-high of a second mix of the key hash, independent of the table slot and residue bits)
            const uint32_t b = (uint32_t)(((uint64_t)mix32(hk ^ 0x85EBCA6Bu) * F) >> 32);
            const uint32_t pos = atomicAdd(&cursor[b], 1u);
            if (pos < it.cap) {
                uint64_t *dst = kmers + (it.base + (unsigned long long)b * it.cap + pos) * W;
#pragma unroll
                for (int j = 0; j < W; j++) dst[j] = c.w[j];
            }
        }
    }
    if constexpr (BLOOM) {
        for (int o = 32; o > 0; o >>= 1) my_new += __shfl_down(my_new, o);
        if (lane == 0 && my_new) atomicAdd(&n_new, my_new);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < OVF_MAX_F; b += COUNT_THREADS) bucket_fill[(uint64_t)blockIdx.x * OVF_MAX_F + b] = cursor[b];
    if constexpr (BLOOM) { if (threadIdx.x == 0) new_count[blockIdx.x] = n_new; }
}

// one non-empty bucket of a scattered partition: nb canonical k-mers from k-mer index `first`
struct BucketRef { unsigned long long first; uint32_t nb, pad; };

// per item: the fullest bucket; the non-empty buckets of the items that fit go on the work list of
// k_count_buckets (an item whose bucket region overflowed is scattered again by the host with the room it needs)
__global__ __launch_bounds__(256) void k_ovf_check(const OvfItem *__restrict__ items, const uint32_t *__restrict__ bucket_fill,
                                                   uint32_t n_items, uint32_t *__restrict__ max_fill,
                                                   BucketRef *__restrict__ list, uint32_t *__restrict__ list_n,
                                                   unsigned long long *__restrict__ sum_fill /* [n_items]: k-mers the item wrote */) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x) {
        uint32_t mx = 0, ne = 0;
        unsigned long long sm = 0;
        const OvfItem it = items[i];
        for (uint32_t b = 0; b < it.F; b++) {
            const uint32_t f = bucket_fill[(uint64_t)i * OVF_MAX_F + b];
            mx = max(mx, f); ne += f != 0; sm += f;
        }
        max_fill[i] = mx; sum_fill[i] = sm;
        if (mx > it.cap || !ne) continue;
        uint32_t at = atomicAdd(list_n, ne);
        for (uint32_t b = 0; b < it.F; b++) {
            const uint32_t f = bucket_fill[(uint64_t)i * OVF_MAX_F + b];
            if (f) { BucketRef r; r.first = it.base + (unsigned long long)b * it.cap; r.nb = f; r.pad = 0; list[at++] = r; }
        }
    }
}

// Counts the buckets of the work list: PERSISTENT workgroups (two per CU, each needs only the k-mer table),
// workgroup g takes entries g, g + G, g + 2G, ...  A bucket is ~3-9 k-mers per thread of work between fixed
// latencies (launch, list entry -> k-mers, barriers, the row reservation's round trip), so the loop is
// software-pipelined: the k-mers of bucket i+1 are loaded into registers before bucket i is counted, the
// list entry of bucket i+2 before that, and the histogram stays in LDS until the workgroup is done.  The
// same table, the same emit and — should a bucket still not fit — the same residue-class splitting (then
// read from memory again) as k_count_partitions.
template <int W>
__global__ __launch_bounds__(COUNT_THREADS, 8) void k_count_buckets(   // 8 waves per SIMD: two workgroups per CU, <= 64 VGPRs
    const BucketRef *__restrict__ list, uint32_t n_list, const uint64_t *__restrict__ kmers,
    uint32_t threshold, unsigned long long *__restrict__ histo, KeyArr<W> out_keys, uint32_t *__restrict__ out_cnt,
    unsigned long long out_cap, unsigned long long *__restrict__ out_cursor,
    unsigned long long *__restrict__ n_inst, uint32_t *__restrict__ flags, uint32_t dbg_arg /* timing experiments (ABLATE builds) */,
    uint32_t bias /* Bloom mode: 1 */, unsigned long long *__restrict__ n_keys /* Bloom mode: distinct k-mers that reached a table */) {
    const uint32_t dbg = SHK_DBG(dbg_arg);
    constexpr uint32_t S = KmerTable<W>::S;
    // k-mers per thread that travel in registers: covers the bucket size the host aims for (<= 1.1 S, binomial
    // spread of a few per cent); the rare longer bucket is read from memory by the residue path below
    constexpr int PF = (int)((S * 114u / 100u + COUNT_THREADS - 1) / COUNT_THREADS);
    constexpr uint32_t PF_MAX = (uint32_t)PF * COUNT_THREADS;
    __shared__ KmerTable<W> tb;
    __shared__ CountCtlCore ctl;
    constexpr uint32_t ELIST = W == 1 ? S : 64;         // emit list: W >= 2 reuses the state words
    __shared__ uint16_t elist[ELIST];
    __shared__ uint32_t whist[500];                     // this workgroup's histogram over all its buckets
    const int lane = threadIdx.x & 63;
    for (uint32_t b = threadIdx.x; b < 500; b += COUNT_THREADS) whist[b] = 0;
    if (threadIdx.x == 0) ctl.n_inst = 0;
    auto load_ref = [&](uint32_t e) -> BucketRef {
        BucketRef r; r.first = 0; r.nb = 0; r.pad = 0;
        if (e < n_list) r = list[e];
        return r;
    };
    auto fetch_all = [&](const BucketRef &r, Kmer<W> (&d)[PF]) {
        const uint64_t *src = kmers + r.first * W;
#pragma unroll
        for (int u = 0; u < PF; u++) {
            // (uniform base per u + one 32-bit lane offset for all loads: no 64-bit address registers per load)
            const uint64_t *src_u = src + (uint64_t)u * COUNT_THREADS * W;
            const uint32_t i = (uint32_t)u * COUNT_THREADS + threadIdx.x;
            if (i < r.nb) {
#pragma unroll
                for (int j = 0; j < W; j++) d[u].w[j] = src_u[threadIdx.x * (uint32_t)W + (uint32_t)j];
            }
        }
    };
    unsigned long long my_keys = 0;                      // (thread 0: table keys over all buckets of this workgroup)
    auto emit = [&](unsigned long long mine) {
        if (bias && threadIdx.x == 0) my_keys += ctl.n_used;
        if constexpr (W == 1) table_emit<W>(tb, ctl, mine, threshold, histo, out_keys, out_cnt, out_cap, out_cursor, elist, dbg, whist, bias);
        else table_emit<W>(tb, ctl, mine, threshold, histo, out_keys, out_cnt, out_cap, out_cursor, tb.state, dbg, whist, bias);   // (the state words are not read after counting)
    };
    uint32_t e = blockIdx.x;
    BucketRef ref = load_ref(e), ref_n = load_ref(e + gridDim.x);
    Kmer<W> kv[PF];                                      // (one buffer: two would cost the second workgroup per CU its registers)
#pragma unroll
    for (int u = 0; u < PF; u++) kv[u] = km_zero<W>();
    if (ref.nb <= PF_MAX) fetch_all(ref, kv);
    for (; e < n_list; e += gridDim.x) {
        const BucketRef cur = ref;
        ref = ref_n; ref_n = load_ref(e + 2u * gridDim.x);
        const uint32_t nb = cur.nb;
        bool done = false;
        if (nb <= PF_MAX) {
            // ---- the common case: the bucket fits the table, its k-mers are in registers
            __syncthreads();                                             // the previous bucket's rows are out
            if (threadIdx.x == 0) {
                ctl.n_used = 0; ctl.overflow = 0; ctl.n_emit = 0; ctl.wave_cursor = 0; ctl.rec_used = 0;
                ctl.prog_num = 0; ctl.prog_den = nb;
            }
            kmer_table_reset<W>(tb, ctl);
            unsigned long long mine = 0;
#pragma unroll
            for (int u = 0; u < PF; u++) {
                if ((uint32_t)u * COUNT_THREADS < nb && !(ctl.overflow || ctl.n_used > (S / 10) * 9)) {   // (whole waves: uniform enough for the ballot)
                    const uint32_t i = (uint32_t)u * COUNT_THREADS + threadIdx.x;
                    bool fresh = false;
                    if (i < nb && dbg != 3) {
                        const uint32_t h = km_mix32<W>(kv[u]);
                        const int r = lds_insert<W, false>(tb, ctl, kv[u], h ^ __builtin_amdgcn_alignbit(h, h, 19), 1u);
                        if (!r) ctl.overflow = 1;
                        fresh = r == 2;
                        mine++;
                    }
                    if (dbg == 3 && i < nb && kv[u].w[0] == 0x123456789ull) ctl.overflow = 1;
                    // most k-mers of an error-rich bucket are new keys: one LDS atomic per wave for the fill level
                    const unsigned long long fm = __ballot(fresh);
                    if (lane == 0 && fm) atomicAdd(&ctl.n_used, (uint32_t)__popcll(fm));
                }
                __builtin_amdgcn_sched_barrier(0);                       // one insert at a time: no hoisting of all PF hashes (registers)
            }
            // the registers are free again: the next bucket's k-mers travel while this one is scanned,
            // its rows reserved and written
            if (ref.nb && ref.nb <= PF_MAX) fetch_all(ref, kv);
            if (dbg == 2 || dbg == 3) continue;                          // timing experiments: no emit
            __syncthreads();
            if (!(ctl.overflow != 0 || ctl.n_used > (S / 10) * 9)) { emit(mine); done = true; }
        } else if (ref.nb && ref.nb <= PF_MAX) fetch_all(ref, kv);
        if (done) continue;
        // ---- a bucket that does not fit (or is longer than the registers hold): residue classes of the key
        // hash, every round reads the bucket from memory again
        const uint64_t *src = kmers + cur.first * W;
        __syncthreads();
        if (threadIdx.x == 0) { ctl.sp = 1; ctl.st_res[0] = 0; ctl.st_step[0] = 1; ctl.st_factor[0] = 1; ctl.st_next[0] = 0; }
        while (true) {
            __syncthreads();
            if (ctl.sp == 0) break;
            const uint32_t top = ctl.sp - 1;
            const uint32_t res = ctl.st_res[top] + ctl.st_next[top] * ctl.st_step[top], mod = ctl.st_step[top] * ctl.st_factor[top];
            __syncthreads();
            if (threadIdx.x == 0) {
                if (++ctl.st_next[top] == ctl.st_factor[top]) ctl.sp--;
                ctl.n_used = 0; ctl.overflow = 0; ctl.n_emit = 0; ctl.wave_cursor = 0; ctl.rec_used = 0;
                ctl.prog_num = 0; ctl.prog_den = nb;
            }
            kmer_table_reset<W>(tb, ctl);
            unsigned long long mine = 0;
            auto fetch = [&](uint32_t i, Kmer<W> &c) {
                if (i < nb) {
#pragma unroll
                    for (int j = 0; j < W; j++) c.w[j] = src[(uint64_t)i * W + j];
                }
            };
            Kmer<W> nxt = km_zero<W>();
            fetch(threadIdx.x, nxt);
            for (uint32_t i0 = 0; i0 < nb; i0 += COUNT_THREADS) {
                if (ctl.overflow || ctl.n_used > (S / 10) * 9) { atomicMax(&ctl.prog_num, i0 + 1u); break; }
                const uint32_t i = i0 + threadIdx.x;
                const Kmer<W> c = nxt;
                fetch(i + COUNT_THREADS, nxt);
                bool fresh = false;
                if (i < nb) {
                    const uint32_t h = km_mix32<W>(c);
                    if (mod == 1 || ((h >> 20) & (mod - 1u)) == res) {
                        const int r = lds_insert<W, false>(tb, ctl, c, h ^ __builtin_amdgcn_alignbit(h, h, 19), 1u);
                        if (!r) ctl.overflow = 1;
                        fresh = r == 2;
                        mine++;
                    }
                }
                const unsigned long long fm = __ballot(fresh);
                if (lane == 0 && fm) atomicAdd(&ctl.n_used, (uint32_t)__popcll(fm));
            }
            __syncthreads();
            const bool over = ctl.overflow != 0 || ctl.n_used > (S / 10) * 9;
            if (over) {
                if (threadIdx.x == 0) atomicAdd(&flags[1], 1u);      // (statistic: bucket rounds split by residue class)
                if (mod >= 4096 || ctl.sp + 1 > 16) { if (threadIdx.x == 0) flags[0] = 1; break; }
                __syncthreads();
                if (threadIdx.x == 0) {
                    const uint32_t num = ctl.prog_num ? ctl.prog_num : ctl.prog_den;
                    const double est = (double)ctl.n_used * (double)ctl.prog_den / (double)num;
                    uint32_t factor = 2;
                    while ((double)factor * (0.6 * S) < est && mod * factor < 4096u) factor <<= 1;
                    ctl.st_res[ctl.sp] = res; ctl.st_step[ctl.sp] = mod; ctl.st_factor[ctl.sp] = factor; ctl.st_next[ctl.sp] = 0;
                    ctl.sp += 1;
                }
                continue;
            }
            emit(mine);
        }
    }
    __syncthreads();
    if (dbg != 1)
    for (uint32_t b = threadIdx.x; b < 500; b += COUNT_THREADS)
        if (whist[b]) atomicAdd(&histo[b], (unsigned long long)whist[b]);
    if (threadIdx.x == 0 && ctl.n_inst && dbg != 8) atomicAdd(n_inst, ctl.n_inst);
    if (threadIdx.x == 0 && bias && my_keys) atomicAdd(n_keys, my_keys);
}

}  // namespace shk
