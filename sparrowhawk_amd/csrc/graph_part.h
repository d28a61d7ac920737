// graph_part.h — a10 assembly:create_graph and a11 assembly:correct_graph (SPEC S8-S9): partitioned membership tables,
// adjacency + unique-neighbour ids, tips, bubbles
// (included by pipeline.hip inside namespace shk, after the device-side views and count_part.h)
#pragma once

// ------------------------------------------------------------------------------------------
// a10: graph table over the solid set
// ------------------------------------------------------------------------------------------
// Graph partition of a k-mer = low bits of the smallest canonical ntHash (32-bit state) over its
// gm-mers; strand-symmetric, so a k-mer and its reverse complement agree.
struct MinScan {
    Nt32State first, last;      // hash state of the first / last gm-mer
    uint32_t h_first, h_last;   // their canonical hashes
    uint32_t min_wo_first;      // min over gm-mers 1 .. w-1
    uint32_t min_wo_last;       // min over gm-mers 0 .. w-2
    __device__ __forceinline__ uint32_t min_all() const { return min(min_wo_first, h_first); }
};

// base j (0 = first) of a k-mer
template <int W> __device__ __forceinline__ uint32_t km_base(const Kmer<W> &x, int k, int j) {
    return km_bits2<W>(x, 2 * (k - 1 - j));
}

template <int W> __device__ __forceinline__ MinScan km_min_scan(const Kmer<W> &x, int k, int gm) {
    MinScan r;
    Nt32State nt{0, 0};
    for (int j = 0; j < gm; j++) nt32_init_step(nt, km_base<W>(x, k, j), (unsigned)j);
    r.first = nt; r.h_first = nt32_canonical(nt);
    r.min_wo_first = 0xFFFFFFFFu; r.min_wo_last = r.h_first;
    const int w = k - gm + 1;
    uint32_t h = r.h_first;
    for (int q = 1; q < w; q++) {
        nt32_roll(nt, km_base<W>(x, k, q - 1), km_base<W>(x, k, q + gm - 1), (unsigned)gm);
        h = nt32_canonical(nt);
        r.min_wo_first = min(r.min_wo_first, h);
        if (q < w - 1) r.min_wo_last = min(r.min_wo_last, h);
    }
    r.last = nt; r.h_last = h;
    return r;
}
// The same scan with the roll terms taken from a 16-entry LDS table ([out<<2|in], as in pass 1): the
// two hot kernels (k_gp_count, k_adjacency) do k + 8 hash steps per node.
__device__ __forceinline__ void nt32_fill_lut(uint2 *lut, unsigned gm) {      // threads 0..15, then __syncthreads()
    if (threadIdx.x < 16) {
        const uint32_t out = threadIdx.x >> 2, in = threadIdx.x & 3u;
        uint2 v;
        v.x = rol32(nt32_seed(out), gm) ^ nt32_seed(in);
        v.y = ror32(nt32_seed(3u - out), 1) ^ rol32(nt32_seed(3u - in), gm - 1);
        lut[threadIdx.x] = v;
    }
}
// hash state of the all-A gm-mer (wave-uniform: computed once per kernel, not per node)
__device__ __forceinline__ Nt32State nt32_all_a(int gm) {
    Nt32State nt; nt.fh = 0; nt.rh = 0;
    for (int j = 0; j < gm; j++) { nt.fh ^= rol32(nt32_seed(0), (unsigned)j); nt.rh ^= rol32(nt32_seed(3), (unsigned)j); }
    return nt;
}
template <int W> __device__ __forceinline__ MinScan km_min_scan_lut(const Kmer<W> &x, int k, int gm, const uint2 *lut,
                                                                    const Nt32State &all_a) {
    MinScan r;
    // The roll terms depend on the bases only, not on the hash: they are fetched eight steps at a time
    // (eight LDS reads in flight) and then folded in — 31 dependent LDS round trips per node otherwise.
    constexpr int CH = 8;
    // the first gm-mer: gm rolls from the all-A window (A leaves, base j enters) — the same LUT path as the scan
    Nt32State nt = all_a;
    for (int j0 = 0; j0 < gm; j0 += CH) {
        uint2 t[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) t[u] = lut[j0 + u < gm ? km_base<W>(x, k, j0 + u) : 0u];    // out = A: index (0 << 2) | in
#pragma unroll
        for (int u = 0; u < CH; u++)
            if (j0 + u < gm) {
                nt.fh = __builtin_amdgcn_alignbit(nt.fh, nt.fh, 31) ^ t[u].x;
                nt.rh = __builtin_amdgcn_alignbit(nt.rh, nt.rh, 1) ^ t[u].y;
            }
    }
    r.first = nt; r.h_first = nt32_canonical(nt);
    r.min_wo_first = 0xFFFFFFFFu; r.min_wo_last = r.h_first;
    const int w = k - gm + 1;
    uint32_t h = r.h_first;
    for (int q0 = 1; q0 < w; q0 += CH) {
        uint2 t[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) {
            const int q = q0 + u;
            t[u] = lut[q < w ? ((km_base<W>(x, k, q - 1) << 2) | km_base<W>(x, k, q + gm - 1)) : 0u];
        }
#pragma unroll
        for (int u = 0; u < CH; u++) {
            const int q = q0 + u;
            if (q < w) {
                nt.fh = __builtin_amdgcn_alignbit(nt.fh, nt.fh, 31) ^ t[u].x;
                nt.rh = __builtin_amdgcn_alignbit(nt.rh, nt.rh, 1) ^ t[u].y;
                h = nt32_canonical(nt);
                r.min_wo_first = min(r.min_wo_first, h);
                if (q < w - 1) r.min_wo_last = min(r.min_wo_last, h);
            }
        }
    }
    r.last = nt; r.h_last = h;
    return r;
}
// the gm-mer that follows the last one when base b is appended / precedes the first when b is prepended
__device__ __forceinline__ uint32_t nt32_next_hash(Nt32State s, uint32_t out, uint32_t in, unsigned gm) {
    nt32_roll(s, out, in, gm);
    return nt32_canonical(s);
}
__device__ __forceinline__ uint32_t nt32_prev_hash(const Nt32State &s, uint32_t new_first, uint32_t old_last, unsigned gm) {
    // inverse of nt32_roll: s is the state of (x0 .. x_{gm-1}); result: state of (b, x0 .. x_{gm-2})
    const uint32_t fh = ror32(s.fh ^ rol32(nt32_seed(new_first), gm) ^ nt32_seed(old_last), 1);
    const uint32_t rh = rol32(s.rh ^ ror32(nt32_seed(3u - new_first), 1) ^ rol32(nt32_seed(3u - old_last), gm - 1), 1);
    return fh < rh ? fh : rh;
}
// the same two with the roll terms from the LDS table: lut[out<<2|in] holds exactly the terms of the
// forward roll (out leaves, in enters) and of its inverse (new_first = out, old_last = in)
__device__ __forceinline__ uint32_t nt32_next_hash_lut(const Nt32State &s, uint32_t out, uint32_t in, const uint2 *lut) {
    const uint2 t = lut[(out << 2) | in];
    const uint32_t fh = __builtin_amdgcn_alignbit(s.fh, s.fh, 31) ^ t.x;
    const uint32_t rh = __builtin_amdgcn_alignbit(s.rh, s.rh, 1) ^ t.y;
    return fh < rh ? fh : rh;
}
__device__ __forceinline__ uint32_t nt32_prev_hash_lut(const Nt32State &s, uint32_t new_first, uint32_t old_last, const uint2 *lut) {
    const uint2 t = lut[(new_first << 2) | old_last];
    const uint32_t a = s.fh ^ t.x, b = s.rh ^ t.y;
    const uint32_t fh = __builtin_amdgcn_alignbit(a, a, 1);            // ror 1
    const uint32_t rh = __builtin_amdgcn_alignbit(b, b, 31);           // rol 1
    return fh < rh ? fh : rh;
}
// Placement inside a mini table: its keys share a minimiser but are otherwise unrelated; an
// add/shift/xor mix of the key words spreads them (integer multiplies are quarter rate on CDNA).
template <int W> __device__ __forceinline__ uint64_t gt_hash(const Kmer<W> &x) {
    uint32_t a = 0x9E3779B9u, b = 0x85EBCA6Bu;
#pragma unroll
    for (int j = 0; j < W; j++) {
        const uint32_t lo = (uint32_t)x.w[j], hi = (uint32_t)(x.w[j] >> 32);
        a = mix32(a ^ lo ^ __builtin_amdgcn_alignbit(hi, hi, 17));
        b = (b ^ hi) + __builtin_amdgcn_alignbit(lo, lo, 11);
        b ^= b >> 15; b += b << 7;
    }
    return ((uint64_t)(b ^ a) << 32) | a;          // high word: fingerprint, low word: slot
}
template <int W> __device__ __forceinline__ uint32_t gt_partition_of(const GraphTable &gt, const Kmer<W> &x, int k) {
    return km_min_scan<W>(x, k, gt.gm).min_all() & gt.gp_mask;
}

// rows per graph partition (rows arrive grouped: one atomic per run of equal partitions in a wave)
template <int W>
__global__ __launch_bounds__(256) void k_gp_count(KeyArr<W> keys, uint32_t n, int k, GraphTable gt,
                                                  uint32_t *__restrict__ gp_of, uint32_t *__restrict__ gp_cnt) {
    __shared__ uint2 lut[16];
    nt32_fill_lut(lut, (unsigned)gt.gm);
    const Nt32State all_a = nt32_all_a(gt.gm);
    __syncthreads();
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_round = (n + stride - 1) / stride * stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        uint32_t p = 0xFFFFFFFFu;
        if (i < n) {
            const MinScan ms = km_min_scan_lut<W>(keys.load(i), k, gt.gm, lut, all_a);
            p = ms.min_all() & gt.gp_mask; gp_of[i] = p;
            if (gt.scan) {                                      // kept for k_graph_local
                gt.scan[i] = make_uint2(ms.first.fh, ms.first.rh);
                gt.scan[(size_t)gt.scan_n + i] = make_uint2(ms.last.fh, ms.last.rh);
                gt.scan[2 * (size_t)gt.scan_n + i] = make_uint2(ms.min_wo_first, ms.min_wo_last);
            }
        }
        unsigned long long todo = __ballot(p != 0xFFFFFFFFu);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lp = (uint32_t)__shfl((int)p, leader);
            const unsigned long long same = __ballot(p == lp) & todo;
            if ((threadIdx.x & 63) == leader) atomicAdd(&gp_cnt[lp], (uint32_t)__popcll(same));
            todo &= ~same;
        }
    }
}

// table sizes (power of two >= 2 x rows, at least 8) and their exclusive prefix sum, plus the
// exclusive prefix sum of the row counts (row list offsets); one workgroup
// slots of a partition's mini table: the power of two >= 2 x rows, at least 8 (no loop: this sits 32 times in every thread of
// the one workgroup of k_gp_scan, where a shift loop cost 20 us)
__device__ __forceinline__ uint32_t gp_table_size(uint32_t rows) {
    const uint32_t want = 2u * rows;
    return want <= 8u ? 8u : 1u << (32 - __clz((int)(want - 1u)));
}
__global__ __launch_bounds__(1024) void k_gp_scan(uint32_t *__restrict__ gp_cnt /* read, then zeroed: k_gp_rows uses it as its cursors */, uint32_t GP,
                                                  unsigned long long *__restrict__ off, uint32_t *__restrict__ msk,
                                                  uint32_t *__restrict__ roff, unsigned long long *__restrict__ total) {
    // One workgroup on the critical path: what counts is the number of DEPENDENT memory round trips.  Chunks of
    // 16384 partitions: 16 independent coalesced loads per thread into LDS (padded: thread t then reads its run of
    // 16 consecutive counts from 16 different banks), thread sums scanned across the block, the run written out.
    // (A loop of dependent loads per thread took 43 us for GP = 16384, the 16-tile version before it 38 us.)
    constexpr uint32_t PER = 16, CH = 1024 * PER;
    __shared__ uint32_t lc[CH + CH / PER];
    __shared__ unsigned long long run_off[1024];
    __shared__ uint32_t run_roff[1024];
    __shared__ unsigned long long wsum[16];
    __shared__ uint32_t rsum[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned long long carry = 0; uint32_t rcarry = 0;
    for (uint32_t c0 = 0; c0 < GP; c0 += CH) {
#pragma unroll
        for (uint32_t j = 0; j < PER; j++) {
            const uint32_t e = j * 1024u + threadIdx.x, p = c0 + e;
            lc[e + (e >> 4)] = p < GP ? gp_cnt[p] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < PER; j++) {
            const uint32_t p = c0 + j * 1024u + threadIdx.x;
            if (p < GP) gp_cnt[p] = 0u;
        }
        __syncthreads();
        unsigned long long my = 0; uint32_t myr = 0;
#pragma unroll
        for (uint32_t j = 0; j < PER; j++) {
            const uint32_t c = lc[threadIdx.x * (PER + 1u) + j];
            uint32_t sz = gp_table_size(c);
            if (c0 + threadIdx.x * PER + j >= GP) sz = 0;  // (past the last partition)
            my += sz; myr += c;
        }
        unsigned long long incl = my; uint32_t rincl = myr;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long u = __shfl_up(incl, o); const uint32_t ru = (uint32_t)__shfl_up((int)rincl, o);
            if (lane >= o) { incl += u; rincl += ru; }
        }
        if (lane == 63) { wsum[wid] = incl; rsum[wid] = rincl; }
        __syncthreads();
        unsigned long long base = carry + incl - my; uint32_t rbase = rcarry + rincl - myr;
        for (int w = 0; w < wid; w++) { base += wsum[w]; rbase += rsum[w]; }
        run_off[threadIdx.x] = base; run_roff[threadIdx.x] = rbase;     // where run t (16 partitions) starts
        __syncthreads();
        // the outputs, coalesced: element e of the chunk belongs to run e / 16; its offset inside the run is a scan over
        // the 16 lanes of its run (four runs per wave)
#pragma unroll 4
        for (uint32_t j = 0; j < PER; j++) {
            const uint32_t e = j * 1024u + threadIdx.x, p = c0 + e;
            const uint32_t c = lc[e + (e >> 4)];
            uint32_t sz = gp_table_size(c);
            if (p >= GP) sz = 0;
            uint32_t si = sz, ci = c;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const uint32_t u = (uint32_t)__shfl_up((int)si, o), v = (uint32_t)__shfl_up((int)ci, o);
                if ((lane & 15) >= o) { si += u; ci += v; }
            }
            if (p < GP) { off[p] = run_off[e >> 4] + (si - sz); msk[p] = sz - 1u; roff[p] = run_roff[e >> 4] + (ci - c); }
        }
        for (int w = 0; w < 16; w++) { carry += wsum[w]; rcarry += rsum[w]; }
        __syncthreads();                                   // (lc, wsum, rsum, run_* are reused by the next chunk)
    }
    if (threadIdx.x == 0) { *total = carry; roff[GP] = rcarry; }
}

// row list per graph partition: rows[roff[p] .. roff[p+1]) (order inside a partition is arbitrary)
__global__ __launch_bounds__(256) void k_gp_rows(const uint32_t *__restrict__ gp_of, uint32_t n,
                                                 const uint32_t *__restrict__ roff, uint32_t *__restrict__ cursor,
                                                 uint32_t *__restrict__ rows) {
    const int lane = threadIdx.x & 63;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_round = (n + stride - 1) / stride * stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const uint32_t p = i < n ? gp_of[i] : 0xFFFFFFFFu;
        const bool valid = p != 0xFFFFFFFFu;
        // the lanes' partitions (a handful of distinct ones: the rows arrive grouped): leader lane, rank and size of
        // every lane's group first — registers only — then ALL leaders reserve with one atomic instruction (one
        // memory round trip per wave instead of one per distinct partition)
        unsigned long long todo = __ballot(valid);
        int my_leader = lane; uint32_t rank = 0, gsize = 0;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lp = (uint32_t)__shfl((int)p, leader);
            const unsigned long long same = __ballot(p == lp) & todo;
            if (valid && p == lp) { my_leader = leader; rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull)); gsize = (uint32_t)__popcll(same); }
            todo &= ~same;
        }
        const uint32_t ro = valid ? roff[p] : 0u;              // (requested together with the reservation, not after it)
        uint32_t base = 0;
        if (valid && lane == my_leader) base = atomicAdd(&cursor[p], gsize);
        base = (uint32_t)__shfl((int)base, my_leader);
        if (valid) rows[ro + base + rank] = i;
    }
}

// rows -> the order of the row lists (rows[r] is the old id of new row r); the lists become the identity
template <int W>
__global__ __launch_bounds__(256) void k_regroup_rows(KeyArr<W> keys, const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ gp_of,
                                                      uint32_t *__restrict__ rows, uint32_t n, KeyArr<W> out_keys,
                                                      uint32_t *__restrict__ out_cnt, uint32_t *__restrict__ out_gp_of) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const uint32_t i = rows[r];
        out_keys.store(r, keys.load(i));
        out_cnt[r] = cnt[i];
        out_gp_of[r] = gp_of[i];
        rows[r] = r;
    }
}

// membership probe in partition p's table
template <int W>
__device__ __forceinline__ uint32_t gt_lookup_in(const GraphTable &gt, const KeyArr<W> &keys, const Kmer<W> &q,
                                                 uint32_t p) {
    const uint64_t h = gt_hash<W>(q);
    const uint32_t fp = (uint32_t)(h >> 32);
    const unsigned long long base = gt.off[p];
    const uint32_t mask = gt.msk[p];
    uint32_t slot = (uint32_t)h & mask;
    for (uint32_t t = 0; t <= mask; t++) {
        const uint64_t e = gt.e[base + slot];
        if (e == EMPTY64) return NIL;
        if ((uint32_t)(e >> 32) == fp) {
            const uint32_t idx = (uint32_t)e;
            if (km_eq<W>(keys.load(idx), q)) return idx;
        }
        slot = (slot + 1) & mask;
    }
    return NIL;
}
// q must be canonical
template <int W>
__device__ __forceinline__ uint32_t gt_lookup(const GraphTable &gt, const KeyArr<W> &keys, const Kmer<W> &q, int k) {
    return gt_lookup_in<W>(gt, keys, q, gt_partition_of<W>(gt, q, k));
}

// adjacency byte (SPEC S8): bit b = successor by appended base b; bit 4+b = predecessor by
// prepended base b, both relative to the canonical orientation.  Also nb[2i+o]: the out-neighbour
// of oriented node (i,o) when it has exactly one (NIL none, NB_MULTI several) — the correction and
// collapse kernels then follow non-branching paths without hashing.
//
// Two kernels.  k_graph_local, one workgroup per graph partition: builds the partition's mini table
// in LDS from its rows (and stores it for everybody else), then resolves every neighbour candidate
// that falls into the SAME partition (~90 %) against LDS; the others are written, densely, to the
// partition's own query region (no global atomics).  k_graph_remote then answers those queries from
// the stored tables with all lanes busy.  A neighbour's partition follows from this node's gm-mer
// hashes and ONE more hash: appending a base drops the first gm-mer and adds one at the end,
// prepending drops the last and adds one in front.
static constexpr uint32_t ADJ_LDS_SLOTS = 2048;        // 16 KB (8 workgroups per CU); larger (skewed) partitions work in global memory
static constexpr uint32_t NB_MULTI = 0xFFFFFFFEu;

// candidate j of node x: j < 4 successor by appended base j, else predecessor by prepended base j-4
template <int W>
__device__ __forceinline__ Kmer<W> adj_candidate(const Kmer<W> &x, const Kmer<W> &rx, int k, uint32_t j, bool &o) {
    Kmer<W> s = x, rr = rx;
    if (j < 4) { km_push_back<W>(s, j, k); km_push_front<W>(rr, 3u - j, k); }
    else { km_push_front<W>(s, j - 4u, k); km_push_back<W>(rr, 3u - (j - 4u), k); }
    o = km_less<W>(rr, s);
    Kmer<W> c;
#pragma unroll
    for (int w = 0; w < W; w++) c.w[w] = o ? rr.w[w] : s.w[w];
    return c;
}

template <int W>
__global__ __launch_bounds__(256) void k_graph_local(KeyArr<W> keys, int k, GraphTable gt,
                                                     const uint32_t *__restrict__ roff, const uint32_t *__restrict__ rows,
                                                     uint8_t *__restrict__ adj, uint32_t *__restrict__ nb,
                                                     unsigned long long *__restrict__ queries, uint32_t *__restrict__ qcnt,
                                                     uint32_t *__restrict__ overflow) {
    const unsigned gm = (unsigned)gt.gm;
    __shared__ uint2 lut[16];
    // LDS copy of the mini table.  One-word keys (k <= 31): the key itself plus the node id (12 B per slot),
    // so a local probe never leaves the CU; wider keys: fingerprint | id, verified against the key array.
    __shared__ uint64_t tab[ADJ_LDS_SLOTS];                // W == 1: keys; else fingerprint << 32 | id
    __shared__ uint32_t tabi[W == 1 ? ADJ_LDS_SLOTS : 1];  // W == 1: node ids
    __shared__ uint32_t q_fill;
    const uint32_t P = blockIdx.x;
    const uint32_t r0 = roff[P], r1 = roff[P + 1];
    const uint32_t pmask = gt.msk[P];
    const bool in_lds = pmask < ADJ_LDS_SLOTS;
    uint64_t *gtab = gt.e + gt.off[P];
    nt32_fill_lut(lut, gm);
    const Nt32State all_a = nt32_all_a(gt.gm);
    if (threadIdx.x == 0) q_fill = 0;
    // ---- build the mini table (keys are distinct: claim the first empty slot).  A canonical k-mer is never
    // all ones (k odd: at most 2k < 64W bits... or its reverse complement, all zeros, would be smaller)
    constexpr bool KEYS_IN_LDS = W == 1;
    for (uint32_t t = threadIdx.x; t <= pmask; t += blockDim.x) {
        if (in_lds) tab[t] = EMPTY64;
        if (!in_lds || KEYS_IN_LDS) gtab[t] = EMPTY64;
    }
    __syncthreads();
    // A thread's rows (every 256th of the partition's list) are used twice — inserted here, expanded below.  Up to four
    // per thread (1024 rows: every partition whose table fits the LDS) stay in registers: their ids and then their keys
    // are requested in two rounds of independent loads, instead of two dependent loads per row in either phase.
    constexpr int KEEP = 4;
    const bool keep = W <= 2 && (r1 - r0) <= (uint32_t)KEEP * 256u;      // (uniform; wider keys: the registers cost more occupancy than the loads)
    // (four named registers each, chosen by compare-and-select: an indexed array would live in scratch memory)
    static_assert(KEEP == 4, "kept rows are four named registers");
    const uint32_t rt = r0 + threadIdx.x;
    const uint32_t i0 = (keep && rt < r1) ? rows[rt] : NIL, i1 = (keep && rt + 256u < r1) ? rows[rt + 256u] : NIL;
    const uint32_t i2 = (keep && rt + 512u < r1) ? rows[rt + 512u] : NIL, i3 = (keep && rt + 768u < r1) ? rows[rt + 768u] : NIL;
    const Kmer<W> k0 = i0 != NIL ? keys.load(i0) : km_zero<W>(), k1 = i1 != NIL ? keys.load(i1) : km_zero<W>();
    const Kmer<W> k2 = i2 != NIL ? keys.load(i2) : km_zero<W>(), k3 = i3 != NIL ? keys.load(i3) : km_zero<W>();
    auto kept_i = [=](uint32_t q) { return q == 0 ? i0 : (q == 1 ? i1 : (q == 2 ? i2 : i3)); };
    auto kept_key = [=](uint32_t q) {
        Kmer<W> r;
#pragma unroll
        for (int w = 0; w < W; w++) r.w[w] = q == 0 ? k0.w[w] : (q == 1 ? k1.w[w] : (q == 2 ? k2.w[w] : k3.w[w]));
        return r;
    };
    for (uint32_t r = r0 + threadIdx.x, q = 0; r < r1; r += blockDim.x, q++) {
        const uint32_t i = keep ? kept_i(q) : rows[r];
        const Kmer<W> key = keep ? kept_key(q) : keys.load(i);
        const uint64_t h = gt_hash<W>(key);
        const uint64_t entry = (h & 0xFFFFFFFF00000000ull) | (uint64_t)i;
        uint32_t slot = (uint32_t)h & pmask;
        bool done = false;
        for (uint32_t t = 0; t <= pmask; t++) {
            if (in_lds && KEYS_IN_LDS) {
                if (atomicCAS((unsigned long long *)&tab[slot], (unsigned long long)EMPTY64, (unsigned long long)key.w[0]) == EMPTY64) {
                    tabi[slot] = i; gtab[slot] = entry; done = true; break;      // same slot in the stored table
                }
            } else {
                unsigned long long *cell = in_lds ? (unsigned long long *)&tab[slot] : (unsigned long long *)&gtab[slot];
                if (atomicCAS(cell, (unsigned long long)EMPTY64, (unsigned long long)entry) == EMPTY64) { done = true; break; }
            }
            slot = (slot + 1) & pmask;
        }
        if (!done) *overflow = 1;
    }
    __syncthreads();
    if (in_lds && !KEYS_IN_LDS) for (uint32_t t = threadIdx.x; t <= pmask; t += blockDim.x) gtab[t] = tab[t];
    // which slots are taken, a bit each (tables are 8 slots at least and start at multiples of 8: whole bytes, no atomics)
    {
        uint8_t *occ = gt.occ + (gt.off[P] >> 3);
        for (uint32_t t8 = threadIdx.x; t8 <= (pmask >> 3); t8 += blockDim.x) {
            uint32_t bits = 0;
#pragma unroll
            for (uint32_t b = 0; b < 8; b++) {
                const uint64_t e = in_lds ? tab[8u * t8 + b] : __hip_atomic_load(&gtab[8u * t8 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bits |= (e != EMPTY64 ? 1u : 0u) << b;
            }
            occ[t8] = (uint8_t)bits;
        }
    }
    if (SHK_DBG(gt.dbg) == 1) { if (threadIdx.x == 0) qcnt[P] = 0; return; }     // timing experiment: table build only
    // ---- neighbours
    unsigned long long *myq = queries + 8ull * r0;
    const int lane = threadIdx.x & 63;
    const uint32_t n_rows = r1 - r0;
    const uint32_t n_round = (n_rows + blockDim.x - 1) / blockDim.x * blockDim.x;
    for (uint32_t rr_ = threadIdx.x, q = 0; rr_ < n_round; rr_ += blockDim.x, q++) {
        const bool act = rr_ < n_rows;
        const uint32_t i = act ? (keep ? kept_i(q) : rows[r0 + rr_]) : 0u;
        Kmer<W> x = km_zero<W>(), rx = km_zero<W>();
        MinScan ms{};
        uint32_t out_b = 0, last_b = 0;
        if (act) {
            x = keep ? kept_key(q) : keys.load(i);
            rx = km_revcomp<W>(x, k);                              // rc(x+b) = (3-b) + rc(x)[..k-1): one revcomp per node
            if (gt.scan) {                                         // the scan k_gp_count made (three 8-byte loads)
                const uint2 a = gt.scan[i], b = gt.scan[(size_t)gt.scan_n + i], c = gt.scan[2 * (size_t)gt.scan_n + i];
                ms.first.fh = a.x; ms.first.rh = a.y; ms.last.fh = b.x; ms.last.rh = b.y; ms.min_wo_first = c.x; ms.min_wo_last = c.y;
                ms.h_first = min(a.x, a.y); ms.h_last = min(b.x, b.y);
            } else
            if (SHK_DBG(gt.dbg) != 2) ms = km_min_scan_lut<W>(x, k, gt.gm, lut, all_a);      // (2: timing experiment without the scan)
            out_b = km_base<W>(x, k, k - (int)gm);                 // first base of the last gm-mer
            last_b = km_base<W>(x, k, (int)gm - 1);                // last base of the first gm-mer
        }
        uint32_t a = 0, n_out = 0, n_in = 0, u_out = NIL, u_in = NIL;
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            bool remote = false; uint32_t p = 0, xowner = 0; bool cross = false;
            if (act) {
                const uint32_t hmin = j < 4 ? min(ms.min_wo_first, nt32_next_hash_lut(ms.last, out_b, j, lut))
                                            : min(ms.min_wo_last, nt32_prev_hash_lut(ms.first, j - 4u, last_b, lut));
                p = hmin & gt.gp_mask;
                if (gt.world > 1) { xowner = gt.owner_of(hmin); cross = xowner != gt.rank; }     // (sharded assembly: the candidate lives on another rank)
                remote = cross || p != P || !in_lds;
                if (SHK_DBG(gt.dbg) == 3) remote = false;                       // timing experiment: scan + candidates only
                if (SHK_DBG(gt.dbg) == 4 && remote) { remote = false; }         // timing experiment: no remote queue
                else if (!remote && SHK_DBG(gt.dbg) != 3) {
                    bool o; const Kmer<W> c = adj_candidate<W>(x, rx, k, j, o);
                    const uint64_t h = gt_hash<W>(c);
                    const uint32_t fp = (uint32_t)(h >> 32);
                    uint32_t slot = (uint32_t)h & pmask, idx = NIL;
                    for (uint32_t t = 0; t <= pmask; t++) {
                        const uint64_t e = tab[slot];
                        if (e == EMPTY64) break;
                        if constexpr (KEYS_IN_LDS) { if (e == c.w[0]) { idx = tabi[slot]; break; } }
                        else if ((uint32_t)(e >> 32) == fp && km_eq<W>(keys.load((uint32_t)e), c)) { idx = (uint32_t)e; break; }
                        slot = (slot + 1) & pmask;
                    }
                    (void)fp;
                    if (idx != NIL) {
                        a |= 1u << j;
                        // a predecessor q -> (x,0) is the edge (x,1) -> rc(q)
                        if (j < 4) { n_out++; u_out = idx * 2u + (o ? 1u : 0u); }
                        else { n_in++; u_in = idx * 2u + (o ? 0u : 1u); }
                    }
                }
            }
            // queue the remote ones: wave-aggregated append to this partition's region
            const unsigned long long m = __ballot(remote);
            if (m) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&q_fill, (uint32_t)__popcll(m));
                base = (uint32_t)__shfl((int)base, 0);
                if (remote) myq[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] =
                    (unsigned long long)i | ((unsigned long long)j << 32) | ((unsigned long long)p << 35) |
                    (cross ? (1ull << 63) | ((unsigned long long)xowner << 52) : 0ull);
            }
        }
        if (act) {
            adj[i] = (uint8_t)a;
            uint2 v;
            v.x = n_out == 0 ? NIL : (n_out == 1 ? u_out : NB_MULTI);
            v.y = n_in == 0 ? NIL : (n_in == 1 ? u_in : NB_MULTI);
            *reinterpret_cast<uint2 *>(nb + 2ull * i) = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) qcnt[P] = q_fill;
}

// answers the cross-partition queries of partition blockIdx.x
// Every query is a chain of four dependent memory round trips (the query, its node's key and the destination table's place,
// the table entry, the key the entry names); a thread takes up to four queries at a time and moves them through the chain
// together, so that a round trip is waited for once per four queries (round 4; one at a time: 177 us for the 4 M queries
// of the bench isolate).
template <int W>
__global__ __launch_bounds__(256) void k_graph_remote(KeyArr<W> keys, int k, GraphTable gt,
                                                      const uint32_t *__restrict__ roff,
                                                      const unsigned long long *__restrict__ queries,
                                                      const uint32_t *__restrict__ qcnt,
                                                      uint8_t *__restrict__ adj, uint32_t *__restrict__ nb) {
    const uint32_t P = blockIdx.x;
    const unsigned long long *myq = queries + 8ull * roff[P];
    const uint32_t nq = qcnt[P];
    constexpr int U = W <= 2 ? 4 : (W <= 4 ? 2 : 1);       // (wide keys: the registers cost more than the round trips)
    for (uint32_t t0 = threadIdx.x; t0 < nq; t0 += blockDim.x * U) {
        unsigned long long q[U]; bool live[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t t = t0 + (uint32_t)u * blockDim.x;
            q[u] = t < nq ? myq[t] : ~0ull;
            live[u] = !(q[u] >> 63);                            // (bit 63: a cross-rank query, answered by its owner — shard_graph.h — or no query)
        }
        Kmer<W> x[U]; unsigned long long tbase[U]; uint32_t tmask[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = (uint32_t)q[u], p = (uint32_t)(q[u] >> 35) & 0x1FFFFu;
            x[u] = live[u] ? keys.load(i) : km_zero<W>();
            tbase[u] = live[u] ? gt.off[p] : 0ull;
            tmask[u] = live[u] ? gt.msk[p] : 0u;
        }
        Kmer<W> c[U]; bool o[U]; uint32_t fp[U], slot[U]; uint64_t e[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t j = (uint32_t)(q[u] >> 32) & 7u;
            const Kmer<W> rx = km_revcomp<W>(x[u], k);
            c[u] = adj_candidate<W>(x[u], rx, k, j, o[u]);
            const uint64_t h = gt_hash<W>(c[u]);
            fp[u] = (uint32_t)(h >> 32); slot[u] = (uint32_t)h & tmask[u];
        }
        bool taken[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned long long s = tbase[u] + slot[u];
            taken[u] = live[u] && ((gt.occ[s >> 3] >> (s & 7u)) & 1u);
        }
#pragma unroll
        for (int u = 0; u < U; u++) e[u] = taken[u] ? gt.e[tbase[u] + slot[u]] : EMPTY64;
        // first candidate entry of every query (the common end: an empty slot, or the one entry with the fingerprint)
        Kmer<W> cand[U]; bool want[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            want[u] = e[u] != EMPTY64 && (uint32_t)(e[u] >> 32) == fp[u];
            cand[u] = want[u] ? keys.load((uint32_t)e[u]) : km_zero<W>();
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!live[u]) continue;
            uint32_t idx = NIL;
            if (want[u] && km_eq<W>(cand[u], c[u])) idx = (uint32_t)e[u];
            else if (e[u] != EMPTY64) {                         // (rare: walk on from the next slot)
                uint32_t s = (slot[u] + 1u) & tmask[u];
                for (uint32_t t = 1; t <= tmask[u]; t++) {
                    const uint64_t ee = gt.e[tbase[u] + s];
                    if (ee == EMPTY64) break;
                    if ((uint32_t)(ee >> 32) == fp[u] && km_eq<W>(keys.load((uint32_t)ee), c[u])) { idx = (uint32_t)ee; break; }
                    s = (s + 1u) & tmask[u];
                }
            }
            if (idx == NIL) continue;
            const uint32_t i = (uint32_t)q[u], j = (uint32_t)(q[u] >> 32) & 7u;
            atomicOr((uint32_t *)adj + (i >> 2), (1u << j) << (8 * (i & 3u)));
            const uint32_t v = j < 4 ? idx * 2u + (o[u] ? 1u : 0u) : idx * 2u + (o[u] ? 0u : 1u);
            uint32_t *dst = nb + 2ull * i + (j < 4 ? 0 : 1);
            if (atomicCAS(dst, NIL, v) != NIL) atomicExch(dst, NB_MULTI);     // second neighbour of this side
        }
    }
}

// ------------------------------------------------------------------------------------------
// oriented-node view of the graph.  v = idx*2 + o.  Adjacency bits are kept alive-aware, so a
// set bit always leads to an alive node and following an edge is one table lookup.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rev4(uint32_t n) {
    return ((n & 1) << 3) | ((n & 2) << 1) | ((n & 4) >> 1) | ((n & 8) >> 3);
}
__device__ __forceinline__ uint32_t outmask_of(uint32_t adjbyte, uint32_t o) {
    return o ? rev4(adjbyte >> 4) : (adjbyte & 15u);
}

template <int W> struct Graph {
    KeyArr<W> keys;
    const uint32_t *cnt;
    uint8_t *adj;
    GraphTable gt;
    const uint32_t *nb;                        // unique out-neighbour at build time (NIL: none or several)
    int k;
    uint32_t n;
    uint32_t xref = 0;                         // sharded assembly: 0x80000000 — an nb[] entry with this bit (and < NB_MULTI) names a node of another rank
    __device__ __forceinline__ uint32_t outmask(uint32_t v) const { return outmask_of(adj[v >> 1], v & 1); }
    __device__ __forceinline__ uint32_t outdeg(uint32_t v) const { return __popc(outmask(v)); }
    __device__ __forceinline__ uint32_t indeg(uint32_t v) const { return __popc(outmask(v ^ 1)); }
    __device__ __forceinline__ Kmer<W> seq(uint32_t v) const {
        Kmer<W> x = keys.load(v >> 1);
        return (v & 1) ? km_revcomp<W>(x, k) : x;
    }
    // follow the out-edge of v labelled by appended base b (bit must be set)
    __device__ __forceinline__ uint32_t follow(uint32_t v, uint32_t b) const {
        Kmer<W> s = seq(v);
        km_push_back<W>(s, b, k);
        int o; Kmer<W> c = km_canonical<W>(s, k, o);
        uint32_t idx = gt_lookup<W>(gt, keys, c, k);
        return idx == NIL ? NIL : idx * 2 + (uint32_t)o;
    }
    __device__ __forceinline__ uint32_t only_out(uint32_t v) const {   // outdeg(v) must be 1
        // edges are only ever removed: a node that had one out-edge when the graph was built and has
        // one now still has that one.  Otherwise (it had several) look the survivor up.
        const uint32_t c = nb[v];
        if (c < NB_MULTI) return c;                    // NIL cannot occur here (outdeg is 1 now, so it was >= 1)
        return follow(v, (uint32_t)__ffs((int)outmask(v)) - 1);
    }
};

// ------------------------------------------------------------------------------------------
// a11: tips (SPEC S9)
// ------------------------------------------------------------------------------------------
struct TipRec { uint32_t start, junction, len, next; unsigned long long sum; };

// candidates: oriented nodes with indeg 0 and outdeg 1
template <int W>
__global__ __launch_bounds__(256) void k_tip_candidates(Graph<W> g, const uint8_t *__restrict__ alive,
                                                        uint32_t *__restrict__ cand,
                                                        unsigned int *__restrict__ n_cand,
                                                        uint2 *__restrict__ init_tip_head /* first round: both entries of every node := NIL */,
                                                        uint8_t *__restrict__ init_mark /* first round: := 0 */) {
    // one thread per node, both orientations: the alive flag and the adjacency byte are requested together (one memory
    // round trip per node)
    const int lane = threadIdx.x & 63;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_round = (g.n + stride - 1) / stride * stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        uint32_t a = 0; bool al = false;
        if (i < g.n) {
            a = g.adj[i]; al = alive[i] != 0;
            if (init_tip_head) { uint2 nil; nil.x = NIL; nil.y = NIL; init_tip_head[i] = nil; init_mark[i] = 0; }   // (instead of two fills)
        }
#pragma unroll
        for (uint32_t o = 0; o < 2; o++) {
            const bool p = al && __popc(outmask_of(a, o ^ 1u)) == 0 && __popc(outmask_of(a, o)) == 1;
            const unsigned long long m = __ballot(p);
            if (!m) continue;
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(n_cand, (unsigned int)__popcll(m));
            base = __shfl(base, 0);
            if (p) cand[base + __popcll(m & ((1ull << lane) - 1ull))] = 2u * i + o;
        }
    }
}

template <int W>
__global__ __launch_bounds__(256) void k_tip_walk(Graph<W> g, const uint32_t *__restrict__ cand,
                                                  const unsigned int *__restrict__ n_cand_p, TipRec *__restrict__ tips,
                                                  unsigned int *__restrict__ n_tips,
                                                  uint32_t *__restrict__ tip_head) {
    const uint32_t T_TIP = 2u * (uint32_t)g.k;
    const uint32_t n_cand = *n_cand_p;                     // counts stay on the device: one host round trip per round
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cand; c += gridDim.x * blockDim.x) {
        const uint32_t v = cand[c];
        uint32_t cur = v, len = 1, J = NIL;
        unsigned long long sum = g.cnt[v >> 1];
        // one memory round trip per step: everything the next iteration needs of node n (adjacency byte, unique
        // out-neighbour, count) is requested as soon as n is known — a dead end of an error-free genome walks the
        // full 2k steps on the critical path (two dependent reads per step: 70 us; one: 40 us)
        uint32_t a_cur = g.adj[cur >> 1], nb_cur = g.nb[cur];
        for (;;) {
            const uint32_t om = outmask_of(a_cur, cur & 1u);
            if (__popc(om) != 1) break;
            const uint32_t n = nb_cur < NB_MULTI ? nb_cur : g.follow(cur, (uint32_t)__ffs((int)om) - 1u);
            if (n == NIL) break;                           // cannot happen with consistent adjacency
            const uint32_t a_n = g.adj[n >> 1], nb_n = g.nb[n], c_n = g.cnt[n >> 1];
            if (__popc(outmask_of(a_n, (n & 1u) ^ 1u)) >= 2) { J = n; break; }
            len++; sum += c_n; cur = n; a_cur = a_n; nb_cur = nb_n;
            if (len > T_TIP) break;
        }
        if (J == NIL || len > T_TIP) continue;
        const uint32_t t = atomicAdd(n_tips, 1u);
        TipRec r; r.start = v; r.junction = J; r.len = len; r.sum = sum;
        r.next = atomicExch(&tip_head[J], t);
        tips[t] = r;
    }
}

// decide on the snapshot: per junction, at most 4 tips hang off tip_head[J]
template <int W>
__global__ __launch_bounds__(256) void k_tip_decide(Graph<W> g, const TipRec *__restrict__ tips,
                                                    const unsigned int *__restrict__ n_tips_p,
                                                    const uint32_t *__restrict__ tip_head,
                                                    uint8_t *__restrict__ kill) {
    const uint32_t n_tips = *n_tips_p;
    for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < n_tips; a += gridDim.x * blockDim.x) {
        const TipRec me = tips[a];
        const uint32_t d = g.indeg(me.junction);
        uint32_t t = 0; bool best = true;
        const Kmer<W> myfirst = g.keys.load(me.start >> 1);
        for (uint32_t b = tip_head[me.junction]; b != NIL; b = tips[b].next) {
            t++;
            if (b == a) continue;
            const TipRec o = tips[b];
            bool better;                                   // is o better than me?
            if (o.len != me.len) better = o.len > me.len;
            else if (o.sum != me.sum) better = o.sum > me.sum;
            else better = km_less<W>(g.keys.load(o.start >> 1), myfirst);
            if (better) best = false;
        }
        kill[a] = (t < d) ? 1 : (best ? 0 : 1);
    }
}

// mark the nodes of killed tips dead and append them to the removed list
template <int W>
__global__ __launch_bounds__(256) void k_tip_remove(Graph<W> g, const TipRec *__restrict__ tips,
                                                    const unsigned int *__restrict__ n_tips_p, const uint8_t *__restrict__ kill,
                                                    uint32_t *__restrict__ tip_head,
                                                    uint8_t *__restrict__ mark) {
    const uint32_t n_tips = *n_tips_p;
    for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < n_tips; a += gridDim.x * blockDim.x) {
        const TipRec me = tips[a];
        if (kill[a]) {
            uint32_t cur = me.start;
            for (uint32_t i = 0; i < me.len; i++) {
                mark[cur >> 1] = 1;
                if (i + 1 < me.len) cur = g.only_out(cur);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_tip_reset_heads(const TipRec *__restrict__ tips, const unsigned int *__restrict__ n_tips_p,
                                                         uint32_t *__restrict__ tip_head) {
    const uint32_t n_tips = *n_tips_p;
    for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < n_tips; a += gridDim.x * blockDim.x)
        tip_head[tips[a].junction] = NIL;
}

// ------------------------------------------------------------------------------------------
// a11: bubbles (SPEC S9)
// ------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void k_fork_candidates(Graph<W> g, const uint8_t *__restrict__ alive,
                                                         uint32_t *__restrict__ cand,
                                                         unsigned int *__restrict__ n_cand) {
    const int lane = threadIdx.x & 63;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_round = (g.n + stride - 1) / stride * stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        uint32_t a = 0; bool al = false;
        if (i < g.n) { a = g.adj[i]; al = alive[i] != 0; }
#pragma unroll
        for (uint32_t o = 0; o < 2; o++) {
            const bool p = al && __popc(outmask_of(a, o)) >= 2;
            const unsigned long long m = __ballot(p);
            if (!m) continue;
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(n_cand, (unsigned int)__popcll(m));
            base = __shfl(base, 0);
            if (p) cand[base + __popcll(m & ((1ull << lane) - 1ull))] = 2u * i + o;
        }
    }
}

template <int W>
__global__ __launch_bounds__(256) void k_bubble(Graph<W> g, const uint32_t *__restrict__ cand,
                                                const unsigned int *__restrict__ n_cand_p, uint8_t *__restrict__ mark) {
    const uint32_t T_BUB = 2u * (uint32_t)g.k;
    const uint32_t n_cand = *n_cand_p;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cand; c += gridDim.x * blockDim.x) {
        const uint32_t S = cand[c];
        const uint32_t om = g.outmask(S);
        uint32_t first[4], end[4], len[4];
        unsigned long long sum[4];
        bool ok[4];
#pragma unroll
        for (uint32_t b = 0; b < 4; b++) {
            ok[b] = false; first[b] = NIL; end[b] = NIL; len[b] = 0; sum[b] = 0;
            if (!((om >> b) & 1u)) continue;
            const uint32_t bn = g.follow(S, b);
            if (bn == NIL || g.indeg(bn) != 1) continue;
            first[b] = bn;
            uint32_t cur = bn, l = 1;
            unsigned long long s = g.cnt[bn >> 1];
            for (;;) {
                if (g.outdeg(cur) != 1) break;
                const uint32_t n = g.only_out(cur);
                if (n == NIL) break;
                if (g.indeg(n) >= 2) { end[b] = n; ok[b] = true; break; }
                if (l + 1 > T_BUB) break;
                l++; s += g.cnt[n >> 1]; cur = n;
            }
            len[b] = l; sum[b] = s;
        }
#pragma unroll
        for (uint32_t a = 0; a < 4; a++) {
            if (!ok[a]) continue;
            const uint32_t E = end[a];
            // evaluate the bubble only from the side with key(S) <= key(rc(E))
            {
                const Kmer<W> ks = g.keys.load(S >> 1), ke = g.keys.load(E >> 1);
                bool le;
                if (km_less<W>(ks, ke)) le = true;
                else if (km_less<W>(ke, ks)) le = false;
                else le = (S & 1u) <= ((E ^ 1u) & 1u);
                if (!le) continue;
            }
            uint32_t grp = 0; bool best = true;
            const Kmer<W> fa = g.keys.load(first[a] >> 1);
#pragma unroll
            for (uint32_t b = 0; b < 4; b++) {
                if (!ok[b] || end[b] != E) continue;
                grp++;
                if (b == a) continue;
                const unsigned long long l = sum[b] * len[a], r = sum[a] * len[b];
                bool better;                               // is branch b better than a?
                if (l != r) better = l > r;
                else if (len[b] != len[a]) better = len[b] < len[a];
                else better = km_less<W>(g.keys.load(first[b] >> 1), fa);
                if (better) best = false;
            }
            if (grp >= 2 && !best) {
                uint32_t cur = first[a];
                for (uint32_t i = 0; i < len[a]; i++) {
                    mark[cur >> 1] = 1;
                    if (i + 1 < len[a]) cur = g.only_out(cur);
                }
            }
        }
    }
}

// gather marked alive nodes into the removed list, clear alive
__global__ __launch_bounds__(256) void k_collect_marked(uint32_t n, uint8_t *__restrict__ mark,
                                                        uint8_t *__restrict__ alive,
                                                        uint32_t *__restrict__ removed,
                                                        unsigned int *__restrict__ n_removed) {
    const int lane = threadIdx.x & 63;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_round = (n + stride - 1) / stride * stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        bool p = false;
        if (i < n && mark[i]) { mark[i] = 0; if (alive[i]) { alive[i] = 0; p = true; } }
        const unsigned long long m = __ballot(p);
        if (!m) continue;
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(n_removed, (unsigned int)__popcll(m));
        base = __shfl(base, 0);
        if (p) removed[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
    }
}

__device__ __forceinline__ void adj_clear_bit(uint8_t *adj, uint32_t idx, uint32_t bit) {
    uint32_t *wptr = (uint32_t *)adj + (idx >> 2);
    atomicAnd(wptr, ~((1u << bit) << (8 * (idx & 3u))));
}

// for every removed node: clear the reciprocal edge bit in each neighbour, then its own byte
template <int W>
__global__ __launch_bounds__(256) void k_apply_removed(Graph<W> g, const uint32_t *__restrict__ removed,
                                                       const unsigned int *__restrict__ n_removed_p) {
    const uint32_t n_removed = *n_removed_p;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_removed; t += gridDim.x * blockDim.x) {
        const uint32_t r = removed[t];
        const uint32_t a = g.adj[r];
        const Kmer<W> x = g.keys.load(r);
        const uint32_t fb = km_first_base<W>(x, g.k), lb = km_last_base<W>(x);
        for (uint32_t b = 0; b < 4; b++) {
            if ((a >> b) & 1u) {                            // edge (r,0) -> u
                Kmer<W> s = x; km_push_back<W>(s, b, g.k);
                int o; Kmer<W> c = km_canonical<W>(s, g.k, o);
                const uint32_t u = gt_lookup<W>(g.gt, g.keys, c, g.k);
                if (u != NIL) adj_clear_bit(g.adj, u, o == 0 ? 4 + fb : 3 - fb);
            }
            if ((a >> (4 + b)) & 1u) {                      // edge p -> (r,0), p spelled b + x[..k-1)
                Kmer<W> s = x; km_push_front<W>(s, b, g.k);
                int o; Kmer<W> c = km_canonical<W>(s, g.k, o);
                const uint32_t u = gt_lookup<W>(g.gt, g.keys, c, g.k);
                if (u != NIL) adj_clear_bit(g.adj, u, o == 0 ? lb : 4 + (3 - lb));
            }
        }
        uint32_t *wptr = (uint32_t *)g.adj + (r >> 2);
        atomicAnd(wptr, ~(0xFFu << (8 * (r & 3u))));
    }
}

