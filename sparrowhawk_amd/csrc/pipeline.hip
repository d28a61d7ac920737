// pipeline.hip — HIP kernels (gfx950, wave64) and the device pipeline of the assembly path.
//
// Reference rows replaced (the Rust lives in rust/sparrowhawk-asm, NOT IN TREE,
// so citations are to the observable phases in www/src/components/pages/AssemblyPage.vue):
//   a4/a5  preprocess:*:loop      -> k_count_segments   (extract + canonicalise + ntHash + insert)
//   a6     histo                  -> k_histogram
//   a8     preprocess:*:filtering -> k_compact          (ballot / prefix-sum stream compaction)
//   a10    assembly:create_graph  -> k_gt_insert, k_adjacency
//   a11    assembly:correct_graph -> k_tip_*, k_bubble_*, k_apply_removed
//   a12    assembly:collapse_graph-> k_succ, k_mark_splitters, k_walk_segments, k_emit
// Integer/hash work only: no MFMA anywhere on this path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "kmer.h"
#include "pipeline.h"

namespace shk {

#define HIPCHK(call)                                                                      \
    do {                                                                                  \
        hipError_t _e = (call);                                                           \
        if (_e != hipSuccess) {                                                           \
            err = std::string(#call) + ": " + hipGetErrorString(_e);                      \
            return -5;                                                                    \
        }                                                                                 \
    } while (0)

static constexpr uint32_t NIL = 0xFFFFFFFFu;
static constexpr uint64_t EMPTY64 = ~0ull;
static constexpr int MAX_PROBE = 4096;
static constexpr int SPLIT_LOG_DEFAULT = 5;    // one sampled splitter every ~32 oriented nodes

// ------------------------------------------------------------------------------------------
// device-side views
// ------------------------------------------------------------------------------------------
template <int W> struct KeyArr {               // SoA key storage
    uint64_t *w[W];
    __device__ __forceinline__ Kmer<W> load(uint64_t i) const {
        Kmer<W> x;
#pragma unroll
        for (int j = 0; j < W; j++) x.w[j] = w[j][i];
        return x;
    }
    __device__ __forceinline__ void store(uint64_t i, const Kmer<W> &x) const {
#pragma unroll
        for (int j = 0; j < W; j++) w[j][i] = x.w[j];
    }
};

template <int W> struct CountTable {           // open addressing, linear probing
    KeyArr<W> keys;
    uint32_t *cnt;
    uint32_t *state;                           // W >= 2 only: 0 empty, 1 being written, 2 ready
    uint64_t mask;
};

// Graph membership table, split into GP mini tables by minimiser partition: a k-mer and its graph
// neighbours share their minimiser nine times out of ten, and the rows arrive grouped by partition
// (pass 2 emits partition by partition), so the 8 probes of a node and of the rows around it go to
// one ~32 KB region that stays in the XCD's L2 — instead of 8 scattered 64-byte fabric requests per
// node into one 134 MB table (the flat version: 3.2 GB fetched for 5 M nodes, profiles/r01_s2_start).
struct GraphTable {                            // entry = fingerprint<<32 | node index
    uint64_t *e;
    const unsigned long long *off;             // [GP] first slot of the partition's table
    const uint32_t *msk;                       // [GP] its size - 1 (power of two)
    uint32_t gp_mask;                          // GP - 1
    int gm;                                    // minimiser length of the graph partition function
};

}  // namespace shk
#include "count_part.h"
namespace shk {

// ------------------------------------------------------------------------------------------
// count table insert (global memory; every concurrent access is an agent-scope atomic)
// ------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ bool ct_insert(const CountTable<W> &t, const Kmer<W> &key, uint64_t h) {
    uint64_t slot = h & t.mask;
    if constexpr (W == 1) {
        for (int p = 0; p < MAX_PROBE; p++) {
            unsigned long long old = atomicCAS((unsigned long long *)&t.keys.w[0][slot],
                                               (unsigned long long)EMPTY64,
                                               (unsigned long long)key.w[0]);
            if (old == EMPTY64 || old == key.w[0]) {
                atomicAdd(&t.cnt[slot], 1u);
                return true;
            }
            slot = (slot + 1) & t.mask;
        }
        return false;
    } else {
        int probes = 0;
        for (;;) {
            uint32_t st = __hip_atomic_load(&t.state[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool won = false;
            if (st == 0) {
                st = atomicCAS(&t.state[slot], 0u, 1u);
                won = (st == 0);
            }
            if (won) {
#pragma unroll
                for (int j = 0; j < W; j++)
                    __hip_atomic_store(&t.keys.w[j][slot], key.w[j], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&t.state[slot], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&t.cnt[slot], 1u);
                return true;
            }
            if (st == 1) continue;             // owner is mid-write: poll the same slot again
            bool eq = true;
#pragma unroll
            for (int j = 0; j < W; j++) {
                uint64_t v = __hip_atomic_load(&t.keys.w[j][slot], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                eq = eq && (v == key.w[j]);
            }
            if (eq) {
                atomicAdd(&t.cnt[slot], 1u);
                return true;
            }
            slot = (slot + 1) & t.mask;
            if (++probes > MAX_PROBE) return false;
        }
    }
}

template <int W> __device__ __forceinline__ bool ct_occupied(const CountTable<W> &t, uint64_t slot) {
    if constexpr (W == 1) return t.keys.w[0][slot] != EMPTY64;
    else return t.state[slot] == 2u;
}

// ------------------------------------------------------------------------------------------
// a4/a5: one lane per segment; both strands and the ntHash pair roll base by base
// ------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void k_count_segments(const uint32_t *__restrict__ bases,
                                                        const uint32_t *__restrict__ seg_off,
                                                        uint32_t n_seg, int k, CountTable<W> tab,
                                                        uint32_t *__restrict__ overflow,
                                                        unsigned long long *__restrict__ n_inst) {
    // pre-rotated ntHash seed tables (wave-uniform)
    const uint64_t so0 = rol64(SHK_NT_A, (unsigned)k), so1 = rol64(SHK_NT_C, (unsigned)k),
                   so2 = rol64(SHK_NT_G, (unsigned)k), so3 = rol64(SHK_NT_T, (unsigned)k);
    const uint64_t ro0 = ror64(SHK_NT_T, 1), ro1 = ror64(SHK_NT_G, 1), ro2 = ror64(SHK_NT_C, 1),
                   ro3 = ror64(SHK_NT_A, 1);
    const uint64_t ri0 = rol64(SHK_NT_T, (unsigned)(k - 1)), ri1 = rol64(SHK_NT_G, (unsigned)(k - 1)),
                   ri2 = rol64(SHK_NT_C, (unsigned)(k - 1)), ri3 = rol64(SHK_NT_A, (unsigned)(k - 1));
    unsigned long long mine = 0;
    for (uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x; seg < n_seg;
         seg += gridDim.x * blockDim.x) {
        const uint32_t start = seg_off[seg], end = seg_off[seg + 1];
        Kmer<W> f = km_zero<W>(), r = km_zero<W>();
        NtState nt{0, 0};
        uint32_t word = bases[start >> 4];
        for (uint32_t pos = start; pos < end; pos++) {
            if ((pos & 15u) == 0) word = bases[pos >> 4];
            const uint32_t b = (word >> (2 * (pos & 15u))) & 3u;
            const uint32_t i = pos - start;
            if (i >= (uint32_t)k) {
                const uint32_t out = km_first_base<W>(f, k);
                nt.fh = rol64(nt.fh, 1) ^ sel4(out, so0, so1, so2, so3) ^ nt_seed(b);
                nt.rh = ror64(nt.rh, 1) ^ sel4(out, ro0, ro1, ro2, ro3) ^ sel4(b, ri0, ri1, ri2, ri3);
            } else {
                nt_init_step(nt, b, i);
            }
            km_push_back<W>(f, b, k);
            km_push_front<W>(r, 3 - b, k);
            if (i + 1 >= (uint32_t)k) {
                const bool use_r = km_less<W>(r, f);
                Kmer<W> c;
#pragma unroll
                for (int j = 0; j < W; j++) c.w[j] = use_r ? r.w[j] : f.w[j];
                if (!ct_insert<W>(tab, c, nt_canonical(nt))) *overflow = 1;
                mine++;
            }
        }
    }
    // one atomic per wave
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_inst, mine);
}

// ------------------------------------------------------------------------------------------
// a6: spectrum histogram (SPEC S5): LDS bins, one global add per bin per block
// ------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void k_histogram(CountTable<W> tab, uint64_t n_slots,
                                                   unsigned long long *__restrict__ histo) {
    __shared__ uint32_t h[500];
    for (int i = threadIdx.x; i < 500; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots;
         s += (uint64_t)gridDim.x * blockDim.x) {
        if (ct_occupied<W>(tab, s)) {
            uint32_t c = tab.cnt[s];
            atomicAdd(&h[c >= 500 ? 499 : c - 1], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 500; i += blockDim.x)
        if (h[i]) atomicAdd(&histo[i], (unsigned long long)h[i]);
}

// ------------------------------------------------------------------------------------------
// a8: filter + compaction: wave ballot, lane prefix by popcount, one cursor add per wave
// ------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void k_compact(CountTable<W> tab, uint64_t n_slots,
                                                 uint32_t threshold, KeyArr<W> out_keys,
                                                 uint32_t *__restrict__ out_cnt,
                                                 unsigned long long *__restrict__ cursor) {
    // one global atomic per block-step (a returning atomic on one address sustains only ~88 / us:
    // one per wave made this kernel 23 ms, profiles/r01_baseline_global_atomics)
    __shared__ uint32_t wave_tot[4];
    __shared__ unsigned long long blk_base;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (n_slots + stride - 1) / stride * stride;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_round; s += stride) {
        bool p = false;
        uint32_t c = 0;
        if (s < n_slots && ct_occupied<W>(tab, s)) {
            c = tab.cnt[s];
            p = c > threshold;
        }
        const unsigned long long m = __ballot(p);
        if (lane == 0) wave_tot[wid] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
            blk_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        if (p) {
            uint64_t o = blk_base + __popcll(m & ((1ull << lane) - 1ull));
            for (int w = 0; w < wid; w++) o += wave_tot[w];
            out_keys.store(o, tab.keys.load(s));
            out_cnt[o] = c;
        }
        __syncthreads();
    }
}

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
template <int W> __device__ __forceinline__ MinScan km_min_scan_lut(const Kmer<W> &x, int k, int gm, const uint2 *lut) {
    MinScan r;
    // the first gm-mer: gm rolls from the all-A window (A leaves, base j enters) — the same LUT path as the scan
    Nt32State nt;
    nt.fh = 0; nt.rh = 0;
    for (int j = 0; j < gm; j++) { nt.fh ^= rol32(nt32_seed(0), (unsigned)j); nt.rh ^= rol32(nt32_seed(3), (unsigned)j); }   // wave-uniform
    for (int j = 0; j < gm; j++) {
        const uint2 t = lut[km_base<W>(x, k, j)];          // out = A: index (0 << 2) | in
        nt.fh = __builtin_amdgcn_alignbit(nt.fh, nt.fh, 31) ^ t.x;
        nt.rh = __builtin_amdgcn_alignbit(nt.rh, nt.rh, 1) ^ t.y;
    }
    r.first = nt; r.h_first = nt32_canonical(nt);
    r.min_wo_first = 0xFFFFFFFFu; r.min_wo_last = r.h_first;
    const int w = k - gm + 1;
    uint32_t h = r.h_first;
    for (int q = 1; q < w; q++) {
        const uint32_t idx = (km_base<W>(x, k, q - 1) << 2) | km_base<W>(x, k, q + gm - 1);
        const uint2 t = lut[idx];
        nt.fh = __builtin_amdgcn_alignbit(nt.fh, nt.fh, 31) ^ t.x;
        nt.rh = __builtin_amdgcn_alignbit(nt.rh, nt.rh, 1) ^ t.y;
        h = nt32_canonical(nt);
        r.min_wo_first = min(r.min_wo_first, h);
        if (q < w - 1) r.min_wo_last = min(r.min_wo_last, h);
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
    __syncthreads();
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_round = (n + stride - 1) / stride * stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        uint32_t p = 0xFFFFFFFFu;
        if (i < n) { p = km_min_scan_lut<W>(keys.load(i), k, gt.gm, lut).min_all() & gt.gp_mask; gp_of[i] = p; }
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
__global__ __launch_bounds__(1024) void k_gp_scan(const uint32_t *__restrict__ gp_cnt, uint32_t GP,
                                                  unsigned long long *__restrict__ off, uint32_t *__restrict__ msk,
                                                  uint32_t *__restrict__ roff, unsigned long long *__restrict__ total) {
    __shared__ unsigned long long wsum[16];
    __shared__ uint32_t rsum[16];
    const uint32_t per = (GP + 1023) / 1024;
    const uint32_t p0 = threadIdx.x * per, p1 = min(GP, p0 + per);
    unsigned long long mine = 0; uint32_t rmine = 0;
    for (uint32_t p = p0; p < p1; p++) {
        const uint32_t c = gp_cnt[p];
        uint32_t sz = 8; const uint32_t want = 2u * c;
        while (sz < want) sz <<= 1;
        msk[p] = sz - 1u; mine += sz; rmine += c;
    }
    unsigned long long incl = mine; uint32_t rincl = rmine;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long u = __shfl_up(incl, o); const uint32_t ru = (uint32_t)__shfl_up((int)rincl, o);
        if (lane >= o) { incl += u; rincl += ru; }
    }
    if (lane == 63) { wsum[wid] = incl; rsum[wid] = rincl; }
    __syncthreads();
    unsigned long long base = 0; uint32_t rbase = 0;
    for (int w = 0; w < wid; w++) { base += wsum[w]; rbase += rsum[w]; }
    unsigned long long run = base + incl - mine; uint32_t rrun = rbase + rincl - rmine;
    for (uint32_t p = p0; p < p1; p++) {
        off[p] = run; run += (unsigned long long)msk[p] + 1ull;
        roff[p] = rrun; rrun += gp_cnt[p];
    }
    if (threadIdx.x == 1023) { *total = base + incl; roff[GP] = rbase + rincl; }
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
        unsigned long long todo = __ballot(p != 0xFFFFFFFFu);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lp = (uint32_t)__shfl((int)p, leader);
            const unsigned long long same = __ballot(p == lp) & todo;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&cursor[lp], (uint32_t)__popcll(same));
            base = (uint32_t)__shfl((int)base, leader);
            if (p == lp) rows[roff[lp] + base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = i;
            todo &= ~same;
        }
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
    __shared__ uint64_t tab[ADJ_LDS_SLOTS];
    __shared__ uint32_t q_fill;
    const uint32_t P = blockIdx.x;
    const uint32_t r0 = roff[P], r1 = roff[P + 1];
    const uint32_t pmask = gt.msk[P];
    const bool in_lds = pmask < ADJ_LDS_SLOTS;
    uint64_t *gtab = gt.e + gt.off[P];
    nt32_fill_lut(lut, gm);
    if (threadIdx.x == 0) q_fill = 0;
    // ---- build the mini table (keys are distinct: claim the first empty slot)
    if (in_lds) { for (uint32_t t = threadIdx.x; t <= pmask; t += blockDim.x) tab[t] = EMPTY64; }
    else { for (uint32_t t = threadIdx.x; t <= pmask; t += blockDim.x) gtab[t] = EMPTY64; }
    __syncthreads();
    for (uint32_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const uint32_t i = rows[r];
        const uint64_t h = gt_hash<W>(keys.load(i));
        const uint64_t entry = (h & 0xFFFFFFFF00000000ull) | (uint64_t)i;
        uint32_t slot = (uint32_t)h & pmask;
        bool done = false;
        for (uint32_t t = 0; t <= pmask; t++) {
            unsigned long long *cell = in_lds ? (unsigned long long *)&tab[slot] : (unsigned long long *)&gtab[slot];
            if (atomicCAS(cell, (unsigned long long)EMPTY64, (unsigned long long)entry) == EMPTY64) { done = true; break; }
            slot = (slot + 1) & pmask;
        }
        if (!done) *overflow = 1;
    }
    __syncthreads();
    if (in_lds) for (uint32_t t = threadIdx.x; t <= pmask; t += blockDim.x) gtab[t] = tab[t];
    // ---- neighbours
    unsigned long long *myq = queries + 8ull * r0;
    const int lane = threadIdx.x & 63;
    const uint32_t n_rows = r1 - r0;
    const uint32_t n_round = (n_rows + blockDim.x - 1) / blockDim.x * blockDim.x;
    for (uint32_t rr_ = threadIdx.x; rr_ < n_round; rr_ += blockDim.x) {
        const bool act = rr_ < n_rows;
        const uint32_t i = act ? rows[r0 + rr_] : 0u;
        Kmer<W> x = km_zero<W>(), rx = km_zero<W>();
        MinScan ms{};
        uint32_t out_b = 0, last_b = 0;
        if (act) {
            x = keys.load(i);
            rx = km_revcomp<W>(x, k);                              // rc(x+b) = (3-b) + rc(x)[..k-1): one revcomp per node
            ms = km_min_scan_lut<W>(x, k, gt.gm, lut);
            out_b = km_base<W>(x, k, k - (int)gm);                 // first base of the last gm-mer
            last_b = km_base<W>(x, k, (int)gm - 1);                // last base of the first gm-mer
        }
        uint32_t a = 0, n_out = 0, n_in = 0, u_out = NIL, u_in = NIL;
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            bool remote = false; uint32_t p = 0;
            if (act) {
                p = (j < 4 ? min(ms.min_wo_first, nt32_next_hash_lut(ms.last, out_b, j, lut))
                           : min(ms.min_wo_last, nt32_prev_hash_lut(ms.first, j - 4u, last_b, lut))) & gt.gp_mask;
                remote = p != P || !in_lds;
                if (!remote) {
                    bool o; const Kmer<W> c = adj_candidate<W>(x, rx, k, j, o);
                    const uint64_t h = gt_hash<W>(c);
                    const uint32_t fp = (uint32_t)(h >> 32);
                    uint32_t slot = (uint32_t)h & pmask, idx = NIL;
                    for (uint32_t t = 0; t <= pmask; t++) {
                        const uint64_t e = tab[slot];
                        if (e == EMPTY64) break;
                        if ((uint32_t)(e >> 32) == fp && km_eq<W>(keys.load((uint32_t)e), c)) { idx = (uint32_t)e; break; }
                        slot = (slot + 1) & pmask;
                    }
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
                    (unsigned long long)i | ((unsigned long long)j << 32) | ((unsigned long long)p << 35);
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
template <int W>
__global__ __launch_bounds__(256) void k_graph_remote(KeyArr<W> keys, int k, GraphTable gt,
                                                      const uint32_t *__restrict__ roff,
                                                      const unsigned long long *__restrict__ queries,
                                                      const uint32_t *__restrict__ qcnt,
                                                      uint8_t *__restrict__ adj, uint32_t *__restrict__ nb) {
    const uint32_t P = blockIdx.x;
    const unsigned long long *myq = queries + 8ull * roff[P];
    const uint32_t nq = qcnt[P];
    for (uint32_t t = threadIdx.x; t < nq; t += blockDim.x) {
        const unsigned long long q = myq[t];
        const uint32_t i = (uint32_t)q, j = (uint32_t)(q >> 32) & 7u, p = (uint32_t)(q >> 35);
        const Kmer<W> x = keys.load(i);
        const Kmer<W> rx = km_revcomp<W>(x, k);
        bool o; const Kmer<W> c = adj_candidate<W>(x, rx, k, j, o);
        const uint32_t idx = gt_lookup_in<W>(gt, keys, c, p);
        if (idx == NIL) continue;
        atomicOr((uint32_t *)adj + (i >> 2), (1u << j) << (8 * (i & 3u)));
        const uint32_t u = j < 4 ? idx * 2u + (o ? 1u : 0u) : idx * 2u + (o ? 0u : 1u);
        uint32_t *slot = nb + 2ull * i + (j < 4 ? 0 : 1);
        if (atomicCAS(slot, NIL, u) != NIL) atomicExch(slot, NB_MULTI);     // second neighbour of this side
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
                                                        unsigned int *__restrict__ n_cand) {
    const int lane = threadIdx.x & 63;
    const uint32_t total = g.n * 2;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_round = (total + stride - 1) / stride * stride;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n_round; v += stride) {
        bool p = false;
        if (v < total && alive[v >> 1]) p = (g.indeg(v) == 0) && (g.outdeg(v) == 1);
        const unsigned long long m = __ballot(p);
        if (!m) continue;
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(n_cand, (unsigned int)__popcll(m));
        base = __shfl(base, 0);
        if (p) cand[base + __popcll(m & ((1ull << lane) - 1ull))] = v;
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
        for (;;) {
            if (g.outdeg(cur) != 1) break;
            const uint32_t n = g.only_out(cur);
            if (n == NIL) break;                           // cannot happen with consistent adjacency
            if (g.indeg(n) >= 2) { J = n; break; }
            len++; sum += g.cnt[n >> 1]; cur = n;
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
    const uint32_t total = g.n * 2;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_round = (total + stride - 1) / stride * stride;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n_round; v += stride) {
        bool p = false;
        if (v < total && alive[v >> 1]) p = g.outdeg(v) >= 2;
        const unsigned long long m = __ballot(p);
        if (!m) continue;
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(n_cand, (unsigned int)__popcll(m));
        base = __shfl(base, 0);
        if (p) cand[base + __popcll(m & ((1ull << lane) - 1ull))] = v;
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

// ------------------------------------------------------------------------------------------
// a12: collapse (SPEC S10).  Simple links over oriented nodes, splitters = all heads plus a 1/32
// sample, one walker per splitter, the splitter list ranked by pointer jumping, then every node
// scatters its base into the contig buffer.
//   winfo[v] = {succ(v) or NIL, count(v>>1)}   one 8-byte read per walker step
//   ol[v]    = {owner splitter, position in its segment}
// A node is sampled by a hash of its ID: a walker decides "is my successor a splitter" from the id it
// just read, without touching the successor (heads are never reached through a simple link: a node
// with a simple predecessor is not a head).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool node_sampled(uint32_t v, uint32_t split_mask) {
    return ((mix32(v ^ 0x5bd1e995u) >> 9) & split_mask) == 0;
}

static constexpr int SS_ITEMS = 16;            // oriented nodes per thread of k_succ_split

template <int W>
__global__ __launch_bounds__(256) void k_succ_split(Graph<W> g, const uint8_t *__restrict__ alive,
                                                    uint2 *__restrict__ winfo, uint32_t *__restrict__ spl,
                                                    uint2 *__restrict__ ol, unsigned int *__restrict__ n_spl,
                                                    uint32_t split_mask) {
    __shared__ uint32_t wtot[SS_ITEMS * 4];
    __shared__ uint32_t woff[SS_ITEMS * 4];
    __shared__ uint32_t blk_base;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t total = g.n * 2;
    const uint32_t base = blockIdx.x * (256u * SS_ITEMS);          // even: v and v^1 sit in adjacent lanes
    uint32_t pbits = 0;
#pragma unroll 1
    for (int it = 0; it < SS_ITEMS; it++) {
        const uint32_t v = base + (uint32_t)it * 256u + threadIdx.x;
        uint32_t s = NIL, c = 0; bool al = false;
        if (v < total) {
            al = alive[v >> 1] != 0;
            if (al) {
                c = g.cnt[v >> 1];
                if (g.outdeg(v) == 1) {
                    const uint32_t u = g.only_out(v);
                    if (u != NIL && g.indeg(u) == 1 && u != v && u != (v ^ 1u)) s = u;
                }
            }
            uint2 w; w.x = s; w.y = c; winfo[v] = w;
        }
        const uint32_t sp = (uint32_t)__shfl_xor((int)s, 1);       // succ of the mirror node
        const bool p = al && (sp == NIL || node_sampled(v, split_mask));   // head or sampled
        const unsigned long long m = __ballot(p);
        if (lane == 0) wtot[it * 4 + wid] = (uint32_t)__popcll(m);
        pbits |= (p ? 1u : 0u) << it;
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                        // exclusive scan of the 64 wave totals
        const uint32_t t = threadIdx.x < SS_ITEMS * 4 ? wtot[threadIdx.x] : 0u;
        uint32_t incl = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += u; }
        if (threadIdx.x < SS_ITEMS * 4) woff[threadIdx.x] = incl - t;
        if (threadIdx.x == 63) blk_base = incl ? atomicAdd(n_spl, incl) : 0u;      // ONE global atomic per block
    }
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < SS_ITEMS; it++) {
        const uint32_t v = base + (uint32_t)it * 256u + threadIdx.x;
        const bool p = (pbits >> it) & 1u;
        const unsigned long long m = __ballot(p);
        uint2 o; o.x = NIL; o.y = 0;
        if (p) {
            const uint32_t i = blk_base + woff[it * 4 + wid] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            spl[i] = v; o.x = i;
        }
        if (v < total) ol[v] = o;
    }
}

struct SegRec { uint32_t node, next_spl, len, last; unsigned long long sum; uint32_t head, pad; };

template <int W>
__global__ __launch_bounds__(256) void k_walk_segments(const uint2 *__restrict__ winfo,
                                                       const uint32_t *__restrict__ spl, uint32_t n_spl,
                                                       uint2 *__restrict__ ol, SegRec *__restrict__ segs,
                                                       uint32_t split_mask) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_spl; i += gridDim.x * blockDim.x) {
        const uint32_t s = spl[i];
        uint32_t cur = s, len = 0, nxt;
        unsigned long long sum = 0;
        for (;;) {
            const uint2 w = winfo[cur];
            if (cur != s) { uint2 o; o.x = i; o.y = len; ol[cur] = o; }
            sum += w.y;
            len++;
            nxt = w.x;
            if (nxt == NIL || node_sampled(nxt, split_mask)) break;
            cur = nxt;
        }
        SegRec r; r.node = s; r.len = len; r.last = cur; r.sum = sum;
        r.next_spl = (nxt == NIL) ? NIL : ol[nxt].x;       // splitters got their owner in k_succ_split
        r.head = winfo[s ^ 1u].x == NIL ? 1u : 0u; r.pad = 0;
        segs[i] = r;
    }
}

// ---- splitter-list ranking on the device (pointer jumping over ~2N/64 elements) ------------------
// P: predecessor pointer converging to the chain's head splitter (heads point to themselves);
// A: nodes before this splitter in its chain; K: counts before it.
__global__ __launch_bounds__(256) void k_rank_init(const SegRec *__restrict__ segs, uint32_t n_spl,
                                                   uint32_t *__restrict__ P, uint32_t *__restrict__ A,
                                                   unsigned long long *__restrict__ K) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_spl; s += gridDim.x * blockDim.x) {
        const SegRec r = segs[s];
        if (r.head) { P[s] = s; A[s] = 0; K[s] = 0; }
        if (r.next_spl != NIL) { P[r.next_spl] = s; A[r.next_spl] = r.len; K[r.next_spl] = r.sum; }
    }
}
__global__ __launch_bounds__(256) void k_rank_jump(uint32_t n_spl, const uint32_t *__restrict__ Pi,
                                                   const uint32_t *__restrict__ Ai,
                                                   const unsigned long long *__restrict__ Ki,
                                                   uint32_t *__restrict__ Po, uint32_t *__restrict__ Ao,
                                                   unsigned long long *__restrict__ Ko) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_spl; s += gridDim.x * blockDim.x) {
        const uint32_t p = Pi[s];
        Po[s] = Pi[p]; Ao[s] = Ai[s] + Ai[p]; Ko[s] = Ki[s] + Ki[p];
    }
}
struct HeadRec { uint32_t spl, head_node, tail_node, emit; unsigned long long len, kc; };
// every chain's tail splitter reports the chain to its head's record slot.  A unitig exists on
// both strands; the strand to emit is the lexicographically smaller spelling (SPEC S10), which
// the first k characters decide: seq(head) against seq(rc(tail)).
template <int W>
__global__ __launch_bounds__(256) void k_rank_tails(Graph<W> g, const SegRec *__restrict__ segs, uint32_t n_spl,
                                                    const uint32_t *__restrict__ P, const uint32_t *__restrict__ A,
                                                    const unsigned long long *__restrict__ K,
                                                    HeadRec *__restrict__ heads, uint32_t *__restrict__ slot_of,
                                                    unsigned int *__restrict__ n_heads) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_spl; s += gridDim.x * blockDim.x) {
        const SegRec r = segs[s];
        if (r.next_spl != NIL) continue;
        const uint32_t root = P[s];
        if (root >= n_spl || !segs[root].head) continue;       // (cannot happen: a chain with a tail has a head)
        const uint32_t slot = atomicAdd(n_heads, 1u);          // one per chain
        HeadRec h; h.spl = root; h.head_node = segs[root].node; h.tail_node = r.last;
        {
            const Kmer<W> a = g.seq(h.head_node), b = g.seq(h.tail_node ^ 1u);
            if (km_less<W>(a, b)) h.emit = 1;
            else if (km_less<W>(b, a)) h.emit = 0;
            else h.emit = h.head_node <= (h.tail_node ^ 1u);       // the chain is its own mirror, or a tie on ids
        }
        h.len = (unsigned long long)A[s] + r.len; h.kc = K[s] + r.sum;
        heads[slot] = h; slot_of[root] = slot;
    }
}

// per node: splitter -> chain head -> output offset (~0 = chain not emitted)
template <int W>
__global__ __launch_bounds__(256) void k_emit(Graph<W> g, const uint8_t *__restrict__ alive,
                                              const uint2 *__restrict__ ol,
                                              const uint32_t *__restrict__ P,
                                              const uint32_t *__restrict__ A,
                                              const uint32_t *__restrict__ slot_of,
                                              const unsigned long long *__restrict__ head_off,
                                              char *__restrict__ out) {
    const uint32_t total = g.n * 2;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < total; v += gridDim.x * blockDim.x) {
        if (!alive[v >> 1]) continue;
        const uint2 own = ol[v];
        const uint32_t s = own.x;
        if (s == NIL) continue;
        const uint32_t slot = slot_of[P[s]];
        if (slot == NIL) continue;
        const unsigned long long off = head_off[slot];
        if (off == ~0ull) continue;
        const uint32_t pos = A[s] + own.y;
        const Kmer<W> x = g.seq(v);
        char *dst = out + off;
        const uint32_t ACGT = 0x54474341u;                 // 'A','C','G','T' little-endian
        dst[g.k - 1 + pos] = (char)((ACGT >> (8 * km_last_base<W>(x))) & 0xFF);
        if (pos == 0) {
            for (int i = 0; i + 1 < g.k; i++) {
                dst[i] = (char)((ACGT >> (8 * km_bits2<W>(x, 2 * (g.k - 1 - i)))) & 0xFF);
            }
        }
    }
}

// compaction of (key, count) rows by count > threshold (used when the fitted threshold is above
// the one the counting pass emitted with)
template <int W>
__global__ __launch_bounds__(256) void k_compact_rows(KeyArr<W> in_keys, const uint32_t *__restrict__ in_cnt,
                                                      uint64_t n, uint32_t threshold, KeyArr<W> out_keys,
                                                      uint32_t *__restrict__ out_cnt,
                                                      unsigned long long *__restrict__ cursor) {
    __shared__ uint32_t wave_tot[4];
    __shared__ unsigned long long blk_base;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (n + stride - 1) / stride * stride;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        uint32_t c = i < n ? in_cnt[i] : 0u;
        const bool p = c > threshold;
        const unsigned long long m = __ballot(p);
        if (lane == 0) wave_tot[wid] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
            blk_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;    // one atomic per block-step
        }
        __syncthreads();
        if (p) {
            uint64_t o = blk_base + __popcll(m & ((1ull << lane) - 1ull));
            for (int w = 0; w < wid; w++) o += wave_tot[w];
            out_keys.store(o, in_keys.load(i));
            out_cnt[o] = c;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_max_u32(const uint32_t *__restrict__ a, uint64_t n, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        m = max(m, a[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// Process-wide cache of device allocations: a handle lives for one preprocess+assemble, and
// hipMalloc/hipFree of its multi-GB buffers cost more than the kernels (measured ~10 ms/step).
// Blocks are returned here on release and reused by the next handle; shk_release_cached_memory()
// gives them back to the driver.
struct DevPool {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> free_blocks;     // (device, bytes) -> block
    bool enabled = true;
    DevPool() { const char *v = getenv("SHK_NO_POOL"); enabled = !(v && *v == '1'); }
    static int cur_dev() { int d = 0; (void)hipGetDevice(&d); return d; }
    void *get(size_t &bytes, hipError_t &e) {
        bytes = (bytes + 4095) & ~(size_t)4095;
        // big scratch buffers vary a little from handle to handle: round them up so the cached block fits again
        if (bytes > ((size_t)256 << 20)) bytes = (bytes + ((size_t)256 << 20) - 1) & ~(((size_t)256 << 20) - 1);
        const int dev = cur_dev();
        if (enabled) {
            std::lock_guard<std::mutex> lk(mu);
            auto it = free_blocks.lower_bound(std::make_pair(dev, bytes));
            if (it != free_blocks.end() && it->first.first == dev && it->first.second <= bytes + bytes / 2 + (1u << 20)) {
                void *p = it->second; bytes = it->first.second; free_blocks.erase(it); e = hipSuccess; return p;
            }
        }
        void *p = nullptr;
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess && enabled) {              // out of memory: drop the cache and retry once
            trim();
            e = hipMalloc(&p, bytes);
        }
        return e == hipSuccess ? p : nullptr;
    }
    void put(void *p, size_t bytes) {                  // called with the owning handle's device current
        if (!p) return;
        if (!enabled) { (void)hipFree(p); return; }
        std::lock_guard<std::mutex> lk(mu);
        free_blocks.emplace(std::make_pair(cur_dev(), bytes), p);
    }
    void trim() {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &kv : free_blocks) (void)hipFree(kv.second);
        free_blocks.clear();
    }
};
static DevPool &dev_pool() { static DevPool *p = new DevPool(); return *p; }   // never destroyed (HIP teardown order)
void device_pool_trim() { dev_pool().trim(); }
void *device_pool_alloc(size_t &bytes) { hipError_t e; return dev_pool().get(bytes, e); }
void device_pool_release(void *p, size_t bytes) { dev_pool().put(p, bytes); }

template <typename T> struct DevBuf {
    T *p = nullptr; size_t n = 0; size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(bytes, o.bytes); }
    void release() { if (p) { dev_pool().put(p, bytes); p = nullptr; n = 0; bytes = 0; } }
    int alloc(size_t count, std::string &err) {
        release();
        if (count == 0) count = 1;
        size_t b = count * sizeof(T);
        hipError_t e;
        p = (T *)dev_pool().get(b, e);
        if (!p) { err = std::string("hipMalloc: ") + hipGetErrorString(e); return -4; }
        n = count; bytes = b; return 0;
    }
};

// pinned host staging buffer (D2H of contigs at full PCIe rate), cached the same way
struct PinnedBuf {
    char *p = nullptr; size_t bytes = 0;
    static std::mutex &mu() { static std::mutex m; return m; }
    static std::multimap<size_t, void *> &cache() { static auto *c = new std::multimap<size_t, void *>(); return *c; }
    ~PinnedBuf() { if (p) { std::lock_guard<std::mutex> lk(mu()); cache().emplace(bytes, p); } }
    int alloc(size_t b, std::string &err) {
        b = (b + 4095) & ~(size_t)4095; if (!b) b = 4096;
        if (p && bytes >= b) return 0;
        if (p) { std::lock_guard<std::mutex> lk(mu()); cache().emplace(bytes, p); p = nullptr; bytes = 0; }
        {
            std::lock_guard<std::mutex> lk(mu());
            auto it = cache().lower_bound(b);
            if (it != cache().end() && it->first <= 2 * b + (1u << 20)) { p = (char *)it->second; bytes = it->first; cache().erase(it); return 0; }
        }
        hipError_t e = hipHostMalloc((void **)&p, b, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; err = std::string("hipHostMalloc: ") + hipGetErrorString(e); return -4; }
        bytes = b; return 0;
    }
};

// streams of freed handles are reused too (create + destroy cost ~0.2 ms per handle); every use of a
// handle's stream ends in a synchronize, so a pooled stream is idle
static std::mutex g_stream_mu;
static std::multimap<int, hipStream_t> &stream_cache() { static auto *c = new std::multimap<int, hipStream_t>(); return *c; }
static hipStream_t stream_pool_get(int dev) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    auto it = stream_cache().find(dev);
    if (it == stream_cache().end()) return nullptr;
    hipStream_t s = it->second; stream_cache().erase(it); return s;
}
static void stream_pool_put(int dev, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    if (stream_cache().size() < 16) { stream_cache().emplace(dev, s); return; }
    (void)hipStreamDestroy(s);
}

static inline int grid_for(uint64_t work, int block = 256, int max_blocks = 256 * 16) {
    uint64_t b = (work + block - 1) / block;
    if (b < 1) b = 1;
    if (b > (uint64_t)max_blocks) b = max_blocks;
    return (int)b;
}

struct EvTimer {
    hipEvent_t a, b; hipStream_t st; bool ok = false;
    explicit EvTimer(hipStream_t s) : st(s) {
        ok = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess;
        if (ok) (void)hipEventRecord(a, st);
    }
    double stop() {
        if (!ok) return 0.0;
        (void)hipEventRecord(b, st); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        return ms;
    }
    ~EvTimer() { if (ok) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } }
};

static inline uint64_t env_u64(const char *name, uint64_t dflt) {
    const char *v = getenv(name);
    return (v && *v) ? strtoull(v, nullptr, 10) : dflt;
}

// minimiser length of the counting partitions and of the graph partitions
static inline int part_m(int k) { return k - (k >= 23 ? 16 : 8) + 1; }

template <int W> class Pipeline : public IPipeline {
public:
    explicit Pipeline(int k) : k_(k) {}
    ~Pipeline() override { if (stream_) stream_pool_put(stream_dev_, stream_); }
    int init(std::string &err) {
        HIPCHK(hipGetDevice(&stream_dev_));
        stream_ = stream_pool_get(stream_dev_);
        if (!stream_) HIPCHK(hipStreamCreate(&stream_));
        HIPCHK(ctl_.alloc(16, err) ? hipErrorOutOfMemory : hipSuccess);
        return 0;
    }
    StageTimes &times() override { return times_; }
    void *stream() override { return (void *)stream_; }
    uint64_t total_instances() const override { return total_instances_; }
    uint64_t n_distinct() const override { return n_distinct_; }
    uint64_t n_solid() const override { return n_solid_; }

    // ---- counting --------------------------------------------------------------------------
    int alloc_table(uint64_t slots, std::string &err) {
        for (int j = 0; j < W; j++) if (int rc = tkeys_[j].alloc(slots, err)) return rc;
        if (int rc = tcnt_.alloc(slots, err)) return rc;
        if (W > 1) if (int rc = tstate_.alloc(slots, err)) return rc;
        tslots_ = slots;
        if (W == 1) HIPCHK(hipMemsetAsync(tkeys_[0].p, 0xFF, slots * 8, stream_));
        else HIPCHK(hipMemsetAsync(tstate_.p, 0, slots * 4, stream_));
        HIPCHK(hipMemsetAsync(tcnt_.p, 0, slots * 4, stream_));
        return 0;
    }
    CountTable<W> table_view() {
        CountTable<W> t;
        for (int j = 0; j < W; j++) t.keys.w[j] = tkeys_[j].p;
        t.cnt = tcnt_.p; t.state = tstate_.p; t.mask = tslots_ - 1;
        return t;
    }

    int count_batch_global(const uint32_t *d_bases, const uint32_t *d_seg_off, uint64_t n_seg,
                           uint64_t n_bases, std::string &err) {
        if (n_seg >= 0xFFFFFFFFull || n_bases >= 0xFFFFFFFFull) { err = "batch too large (>= 2^32 bases)"; return -1; }
        if (n_seg == 0) return 0;
        // single batch per table in this version; size from the instance upper bound
        const uint64_t inst_ub = n_bases - n_seg * (uint64_t)(k_ - 1);
        if (tslots_ == 0) {
            uint64_t want = inst_ub / 4 + 1024;
            uint64_t slots = 1ull << 16;
            while (slots < want) slots <<= 1;
            if (int rc = alloc_table(slots, err)) return rc;
            pending_.clear();
        }
        pending_.push_back({d_bases, d_seg_off, n_seg});
        for (;;) {
            HIPCHK(hipMemsetAsync(ctl_.p, 0, 16 * sizeof(unsigned long long), stream_));
            EvTimer t(stream_);
            hipLaunchKernelGGL(k_count_segments<W>, dim3(grid_for(n_seg)), dim3(256), 0, stream_, d_bases,
                               d_seg_off, (uint32_t)n_seg, k_, table_view(), (uint32_t *)(ctl_.p + 1),
                               ctl_.p + 0);
            HIPCHK(hipGetLastError());
            double ms = t.stop();
            unsigned long long h[2];
            HIPCHK(hipMemcpyAsync(h, ctl_.p, sizeof h, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            if ((uint32_t)h[1] == 0) {
                times_.add("count_kernel", ms);
                total_instances_ += h[0];
                return 0;
            }
            // table too small: grow 4x and recount every batch seen so far
            if (pending_.size() > 1) { err = "count table overflow across batches"; return -6; }
            times_.add("count_retry", ms);
            if (int rc = alloc_table(tslots_ * 4, err)) return rc;
        }
    }

    // ---- partitioned counting (count_part.h) ---------------------------------------------------
    template <int WBLK>
    void launch_partition(const uint32_t *d_bases, const uint32_t *d_seg_off, uint32_t n_seg) {
        hipLaunchKernelGGL((k_partition<W, WBLK>), dim3(pp_.G), dim3(PART_THREADS), 0, stream_, d_bases, d_seg_off,
                           n_seg, pp_, recs_.p, fill_.p, (uint32_t *)(ctl_.p + 8));
    }

    int count_batch(const uint32_t *d_bases, const uint32_t *d_seg_off, uint64_t n_seg, uint64_t n_bases,
                    std::string &err) override {
        if (global_mode_) return count_batch_global(d_bases, d_seg_off, n_seg, n_bases, err);
        if (n_seg >= 0xFFFFFFFFull || n_bases >= 0xFFFFFFFFull) { err = "batch too large (>= 2^32 bases)"; return -1; }
        if (n_seg == 0) return 0;
        // several batches per handle (chunked / streamed / > 2^32-base inputs): the previous batch's records
        // are packed densely into a buffer of their own; pass 2 then reads one run per batch and partition
        if (have_parts_) if (int rc = pack_current_batch(err)) return rc;
        constexpr int RW = 2 * W;
        const int wblk = k_ >= 23 ? 16 : 8;
        const uint64_t inst_ub = n_bases - n_seg * (uint64_t)(k_ - 1);
        int cus = 256;
        { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); }
        const uint64_t n_super = (n_seg + PART_THREADS - 1) / PART_THREADS;
        pp_.k = k_; pp_.m = k_ - wblk + 1; pp_.dbg_nostore = (uint32_t)env_u64("SHK_DEBUG_NOSTORE", 0);
        pp_.max_n = std::min<uint32_t>(32u * RW - 3u - (uint32_t)(k_ - 1), 63u);
        if (uint64_t mn = env_u64("SHK_PART_MAXN", 0)) pp_.max_n = std::min<uint32_t>(pp_.max_n, (uint32_t)mn);
        pp_.G = (uint32_t)std::min<uint64_t>((uint64_t)std::min(cus, 256), n_super);
        uint32_t P = 64;
        // instances per partition: sized so that the distinct k-mers of a 100x isolate load the LDS k-mer table to ~40 %
        const uint64_t per_part = env_u64("SHK_PART_INST", W == 1 ? 100000 : 40000);
        while (P < (uint32_t)PART_MAX_P && (uint64_t)P * per_part < inst_ub) P <<= 1;
        if (uint64_t fp = env_u64("SHK_PART_P", 0)) P = (uint32_t)fp;
        if (forced_P_) P = forced_P_;
        if (!batches_.empty()) P = pp_.P;                 // every batch uses the first batch's partitioning
        pp_.P = P;
        // records per slice: mean run length is ~(WBLK+1)/2 k-mers (shorter if max_n caps it); 2x slack
        const uint64_t per_rec = pp_.max_n >= 16 ? 4 : 2;
        uint64_t cap = inst_ub / (per_rec * P * pp_.G) + 32;
        for (int attempt = 0; attempt < 2; attempt++) {
            pp_.slice_cap = (uint32_t)cap;
            const uint64_t n_slices = (uint64_t)P * pp_.G;
            if (int rc = recs_.alloc(n_slices * cap * RW, err)) return rc;
            if (int rc = fill_.alloc(n_slices, err)) return rc;
            HIPCHK(hipMemsetAsync(fill_.p, 0, n_slices * 4, stream_));
            HIPCHK(hipMemsetAsync(ctl_.p, 0, 16 * sizeof(unsigned long long), stream_));
            EvTimer t(stream_);
            if (wblk == 16) launch_partition<16>(d_bases, d_seg_off, (uint32_t)n_seg);
            else launch_partition<8>(d_bases, d_seg_off, (uint32_t)n_seg);
            HIPCHK(hipGetLastError());
            const double ms = t.stop();
            unsigned long long h[2];
            HIPCHK(hipMemcpyAsync(h, ctl_.p + 8, sizeof h, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            const uint32_t *fl = (const uint32_t *)&h[0];
            if (fl[1]) { err = "a read segment exceeds 32768 bases (split it on the host)"; return -1; }
            const uint32_t max_fill = (uint32_t)h[1];
            if (max_fill <= cap) {
                times_.add("partition_kernel", ms);
                have_parts_ = true;
                // run table of the local layout: one run per (partition, producer workgroup)
                if (int rc = run_off_.alloc(n_slices, err)) return rc;
                if (int rc = run_cnt_.alloc(n_slices, err)) return rc;
                hipLaunchKernelGGL(k_make_runs, dim3(grid_for(n_slices)), dim3(256), 0, stream_, fill_.p, pp_, recs_.p,
                                   (uint32_t)RW, run_off_.p, run_cnt_.p);
                HIPCHK(hipGetLastError());
                run_view_.run_addr16 = run_off_.p; run_view_.run_cnt = run_cnt_.p;
                run_view_.S = pp_.G; run_view_.k = k_; n_count_parts_ = pp_.P; run_view_.dbg = (uint32_t)env_u64("SHK_DEBUG_P2", 0);
                return 0;
            }
            times_.add("partition_retry", ms);
            cap = (uint64_t)max_fill + 8;               // exact from the counting run
        }
        err = "partition slices overflowed twice";
        return -6;
    }

    // ---- batches ---------------------------------------------------------------------------------
    struct BatchRecs { DevBuf<uint64_t> dense; std::vector<unsigned long long> part_off; /* [P+1], records */ };

    // the batch that still sits in its [p][g] slices -> a dense buffer (partition-major), slices released
    int pack_current_batch(std::string &err) {
        constexpr int RW = 2 * W;
        DevBuf<unsigned long long> tot, base;
        if (int rc = tot.alloc(pp_.P, err)) return rc;
        if (int rc = base.alloc(pp_.P, err)) return rc;
        hipLaunchKernelGGL(k_part_totals, dim3(pp_.P), dim3(256), 0, stream_, fill_.p, pp_, tot.p);
        HIPCHK(hipGetLastError());
        std::vector<unsigned long long> h(pp_.P);
        HIPCHK(hipMemcpyAsync(h.data(), tot.p, (size_t)pp_.P * 8, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        std::unique_ptr<BatchRecs> b(new BatchRecs());
        b->part_off.assign(pp_.P + 1, 0);
        for (uint32_t p = 0; p < pp_.P; p++) b->part_off[p + 1] = b->part_off[p] + h[p];
        if (int rc = b->dense.alloc(b->part_off[pp_.P] * RW + 2, err)) return rc;
        HIPCHK(hipMemcpyAsync(base.p, b->part_off.data(), (size_t)pp_.P * 8, hipMemcpyHostToDevice, stream_));
        EvTimer t(stream_);
        hipLaunchKernelGGL((k_pack_partition<2 * W>), dim3(pp_.P), dim3(256), 0, stream_, recs_.p, fill_.p, pp_, base.p, b->dense.p);
        HIPCHK(hipGetLastError());
        times_.add("batch_pack_kernel", t.stop());
        HIPCHK(hipStreamSynchronize(stream_));
        batches_.push_back(std::move(b));
        recs_.release(); fill_.release(); run_off_.release(); run_cnt_.release(); have_parts_ = false;
        return 0;
    }

    // run tables over the packed batches: partition p, run b = batch b's records of p
    int make_batch_run_view(std::string &err) {
        const uint32_t nb = (uint32_t)batches_.size();
        if (nb > 256) { err = "more than 256 batches per handle"; return -1; }
        const uint64_t n_runs = (uint64_t)pp_.P * nb;
        std::vector<unsigned long long> addr16(n_runs); std::vector<uint32_t> cnt(n_runs);
        for (uint32_t p = 0; p < pp_.P; p++)
            for (uint32_t b = 0; b < nb; b++) {
                const BatchRecs &B = *batches_[b];
                const unsigned long long c = B.part_off[p + 1] - B.part_off[p];
                if (c > 0xFFFFFFFFull) { err = "partition too large in one batch"; return -1; }
                addr16[(uint64_t)p * nb + b] = ((unsigned long long)(uintptr_t)B.dense.p >> 4) + B.part_off[p] * (unsigned long long)W;
                cnt[(uint64_t)p * nb + b] = (uint32_t)c;
            }
        if (int rc = run_off_.alloc(n_runs, err)) return rc;
        if (int rc = run_cnt_.alloc(n_runs, err)) return rc;
        HIPCHK(hipMemcpyAsync(run_off_.p, addr16.data(), n_runs * 8, hipMemcpyHostToDevice, stream_));
        HIPCHK(hipMemcpyAsync(run_cnt_.p, cnt.data(), n_runs * 4, hipMemcpyHostToDevice, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        run_view_.run_addr16 = run_off_.p; run_view_.run_cnt = run_cnt_.p;
        run_view_.S = nb; run_view_.k = k_; n_count_parts_ = pp_.P; run_view_.dbg = 0;
        have_parts_ = true;
        return 0;
    }
    void expect_more_batches() override { if (!forced_P_ && batches_.empty() && !have_parts_) forced_P_ = (uint32_t)PART_MAX_P; }

    // pass 2 into (keys, cnt) with the given emit threshold; sizes the output by retrying.
    // Partitions whose distinct k-mers do not fit the LDS table are listed by the first launch and
    // then repartitioned at k-mer level (k_ovf_scatter -> k_count_buckets); should a bucket region
    // overflow (extreme skew), that partition falls back to in-kernel residue-class re-runs.
    int run_count_partitions(const RunView &rv, uint32_t n_parts, uint32_t threshold, DevBuf<uint64_t> (&keys)[W],
                             DevBuf<uint32_t> &cnt, uint64_t &n_rows, uint64_t hist_out[500], uint64_t &inst_out,
                             uint64_t cap_hint, double &ms_out, std::string &err) {
        if (n_parts == 0) { n_rows = 0; inst_out = 0; memset(hist_out, 0, 500 * 8); ms_out = 0; return 0; }
        constexpr uint32_t S = CountShared<W>::S;
        const bool repartition = env_u64("SHK_NO_REPARTITION", 0) == 0;
        DevBuf<unsigned long long> dh;
        DevBuf<OvfRec> d_ovf; DevBuf<OvfItem> d_items; DevBuf<uint32_t> d_fill, d_list, d_maxfill; DevBuf<uint64_t> d_kmers;
        if (int rc = dh.alloc(500, err)) return rc;
        if (repartition) if (int rc = d_ovf.alloc(n_parts, err)) return rc;
        uint64_t cap = cap_hint;
        for (int attempt = 0; attempt < 2; attempt++) {
            for (int j = 0; j < W; j++) if (int rc = keys[j].alloc(cap, err)) return rc;
            if (int rc = cnt.alloc(cap, err)) return rc;
            HIPCHK(hipMemsetAsync(dh.p, 0, 500 * 8, stream_));
            HIPCHK(hipMemsetAsync(ctl_.p, 0, 16 * sizeof(unsigned long long), stream_));
            KeyArr<W> ok; for (int j = 0; j < W; j++) ok.w[j] = keys[j].p;
            EvTimer t(stream_);
            hipLaunchKernelGGL(k_count_partitions<W>, dim3(n_parts), dim3(COUNT_THREADS), 0, stream_, rv, threshold,
                               dh.p, ok, cnt.p, (unsigned long long)cap, ctl_.p + 0, ctl_.p + 1,
                               (uint32_t *)(ctl_.p + 2), (const uint32_t *)nullptr, repartition ? d_ovf.p : (OvfRec *)nullptr,
                               (uint32_t *)(ctl_.p + 3));
            HIPCHK(hipGetLastError());
            ms_out = t.stop();
            unsigned long long h[4];
            HIPCHK(hipMemcpyAsync(h, ctl_.p, sizeof h, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            const uint32_t n_ovf = (uint32_t)h[3];
            if (n_ovf) {
                EvTimer t2(stream_);
                std::vector<OvfRec> ov(n_ovf);
                HIPCHK(hipMemcpy(ov.data(), d_ovf.p, (size_t)n_ovf * sizeof(OvfRec), hipMemcpyDeviceToHost));
                std::vector<OvfItem> items(n_ovf);
                for (uint32_t i = 0; i < n_ovf; i++) {
                    // buckets sized by INSTANCES (1.5 table sizes each): the distinct/instance estimate of the
                    // aborted round is biased high (repeats show up late), and a bucket that turns out to hold
                    // too many distinct k-mers only costs itself a second pass over its own k-mer list
                    uint32_t F = 2;
                    while ((double)F * (1.5 * S) < (double)ov[i].instances && F < OVF_MAX_F) F <<= 1;
                    (void)ov[i].est_distinct;
                    const unsigned long long capb = ov[i].instances * env_u64("SHK_OVF_CAP_PCT", 150) / (100ull * F) + 256;   // 50 % slack
                    items[i].p = ov[i].p; items[i].F = F; items[i].cap = (uint32_t)std::min<unsigned long long>(capb, 0xFFFFFFF0ull);
                    items[i].pad = 0; items[i].base = 0;
                }
                // scatter + count; an item whose bucket region overflows (a Poisson tail of heavy k-mers in one
                // bucket) is scattered again with twice the room, at most three times, then re-run by residue classes
                std::vector<uint32_t> bad;
                size_t n_good_total = 0;
                const int max_passes = (int)env_u64("SHK_OVF_MAX_PASSES", 4);
                for (int pass = 0; pass < max_passes && !items.empty(); pass++) {
                    const uint32_t ni = (uint32_t)items.size();
                    unsigned long long tot = 0;
                    for (auto &it : items) { it.base = tot; tot += (unsigned long long)it.F * it.cap; }
                    if (int rc = d_items.alloc(ni, err)) return rc;
                    if (int rc = d_fill.alloc((size_t)ni * OVF_MAX_F, err)) return rc;
                    if (int rc = d_kmers.alloc(tot * W, err)) return rc;
                    HIPCHK(hipMemcpyAsync(d_items.p, items.data(), (size_t)ni * sizeof(OvfItem), hipMemcpyHostToDevice, stream_));
                    hipLaunchKernelGGL(k_ovf_scatter<W>, dim3(ni), dim3(COUNT_THREADS), 0, stream_, rv, d_items.p, d_kmers.p, d_fill.p);
                    HIPCHK(hipGetLastError());
                    if (int rc = d_maxfill.alloc(ni, err)) return rc;
                    hipLaunchKernelGGL(k_ovf_check, dim3(grid_for(ni)), dim3(256), 0, stream_, d_items.p, d_fill.p, ni, d_maxfill.p);
                    HIPCHK(hipGetLastError());
                    std::vector<uint32_t> mxf(ni);
                    HIPCHK(hipMemcpyAsync(mxf.data(), d_maxfill.p, (size_t)ni * 4, hipMemcpyDeviceToHost, stream_));
                    HIPCHK(hipStreamSynchronize(stream_));
                    std::vector<OvfItem> again;
                    uint32_t n_good = 0, max_f = 2;
                    for (uint32_t i = 0; i < ni; i++) {
                        if (mxf[i] <= items[i].cap) { n_good++; max_f = std::max(max_f, items[i].F); }
                        else if (pass + 1 < max_passes && (unsigned long long)mxf[i] + 256 < 0xFFFFFFF0ull) {
                            OvfItem it = items[i]; it.cap = mxf[i] + 256; again.push_back(it);       // the exact need is known now
                        } else bad.push_back(items[i].p);
                    }
                    if (n_good) {                                    // (overflowed items were switched off on the device)
                        hipLaunchKernelGGL(k_count_buckets<W>, dim3(max_f, ni), dim3(COUNT_THREADS), 0, stream_,
                                           d_items.p, d_kmers.p, d_fill.p, threshold, dh.p, ok, cnt.p, (unsigned long long)cap,
                                           ctl_.p + 0, ctl_.p + 1, (uint32_t *)(ctl_.p + 2));
                        HIPCHK(hipGetLastError());
                        HIPCHK(hipStreamSynchronize(stream_));      // d_items / d_kmers are reused by the next pass
                        n_good_total += n_good;
                    }
                    items.swap(again);
                }
                if (!bad.empty()) {
                    if (int rc = d_list.alloc(bad.size(), err)) return rc;
                    HIPCHK(hipMemcpyAsync(d_list.p, bad.data(), bad.size() * 4, hipMemcpyHostToDevice, stream_));
                    hipLaunchKernelGGL(k_count_partitions<W>, dim3((unsigned)bad.size()), dim3(COUNT_THREADS), 0, stream_, rv, threshold,
                                       dh.p, ok, cnt.p, (unsigned long long)cap, ctl_.p + 0, ctl_.p + 1,
                                       (uint32_t *)(ctl_.p + 2), (const uint32_t *)d_list.p, (OvfRec *)nullptr, (uint32_t *)nullptr);
                    HIPCHK(hipGetLastError());
                }
                ms_out += t2.stop();
                times_.add("count_repartitioned_x1", (double)n_good_total);
                times_.add("count_residue_rerun_x1", (double)bad.size());
                HIPCHK(hipMemcpyAsync(h, ctl_.p, sizeof h, hipMemcpyDeviceToHost, stream_));
                HIPCHK(hipStreamSynchronize(stream_));
            }
            HIPCHK(hipMemcpyAsync(hist_out, dh.p, 500 * 8, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            if ((uint32_t)h[2]) { err = "partition too large for the LDS table even after 4096-way residue splitting"; return -6; }
            n_rows = h[0]; inst_out = h[1];
            if (n_rows <= cap) return 0;
            cap = n_rows;                                 // exact; run again
        }
        err = "row buffer overflowed twice";
        return -6;
    }

    int histogram(uint64_t histo[500], uint32_t emit_threshold, std::string &err) override {
        if (!global_mode_) {
            memset(histo, 0, 500 * 8);
            n_distinct_ = 0; n_emitted_ = 0; emit_threshold_ = emit_threshold;
            if (!batches_.empty()) {
                if (have_parts_ && recs_.p) if (int rc = pack_current_batch(err)) return rc;
                if (int rc = make_batch_run_view(err)) return rc;
            }
            if (have_parts_) {
                const uint64_t inst_ub_rows = total_rows_hint();
                double ms = 0; uint64_t inst = 0;
                if (int rc = run_count_partitions(run_view_, n_count_parts_, emit_threshold, ekeys_, ecnt_, n_emitted_, histo, inst,
                                                  inst_ub_rows, ms, err)) return rc;
                times_.add("count_kernel", ms);
                total_instances_ = inst;
            }
            for (int i = 0; i < 500; i++) { histo_[i] = histo[i]; n_distinct_ += histo[i]; }
            return 0;
        }
        DevBuf<unsigned long long> dh;
        if (int rc = dh.alloc(500, err)) return rc;
        HIPCHK(hipMemsetAsync(dh.p, 0, 500 * 8, stream_));
        if (tslots_) {
            EvTimer t(stream_);
            hipLaunchKernelGGL(k_histogram<W>, dim3(grid_for(tslots_)), dim3(256), 0, stream_, table_view(),
                               tslots_, dh.p);
            HIPCHK(hipGetLastError());
            times_.add("histogram_kernel", t.stop());
        }
        HIPCHK(hipMemcpyAsync(histo, dh.p, 500 * 8, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        n_distinct_ = 0;
        for (int i = 0; i < 500; i++) { histo_[i] = histo[i]; n_distinct_ += histo[i]; }
        return 0;
    }

    uint64_t total_rows_hint() const {
        // rows = distinct k-mers above the emit threshold; unknown before the pass, retried if short
        uint64_t slots = (uint64_t)pp_.P * pp_.G * pp_.slice_cap;         // records >= rows / max_n
        if (!batches_.empty()) { slots = 0; for (auto &b : batches_) slots += b->part_off[pp_.P]; }
        return std::max<uint64_t>(1u << 16, slots / 8);
    }

    int compact_into(uint32_t threshold, uint64_t expect, DevBuf<uint64_t> (&keys)[W], DevBuf<uint32_t> &cnt,
                     std::string &err) {
        for (int j = 0; j < W; j++) if (int rc = keys[j].alloc(expect, err)) return rc;
        if (int rc = cnt.alloc(expect, err)) return rc;
        if (!tslots_ || !expect) return 0;
        HIPCHK(hipMemsetAsync(ctl_.p, 0, 16 * sizeof(unsigned long long), stream_));
        KeyArr<W> ok; for (int j = 0; j < W; j++) ok.w[j] = keys[j].p;
        hipLaunchKernelGGL(k_compact<W>, dim3(grid_for(tslots_)), dim3(256), 0, stream_, table_view(), tslots_,
                           threshold, ok, cnt.p, ctl_.p + 0);
        HIPCHK(hipGetLastError());
        unsigned long long got = 0;
        HIPCHK(hipMemcpyAsync(&got, ctl_.p, 8, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        if (got != expect) { err = "compaction count mismatch"; return -6; }
        return 0;
    }

    int filter(uint32_t threshold, std::string &err) override {
        uint64_t expect = 0;
        if (threshold >= 500) { err = "threshold out of range"; return -1; }
        for (uint32_t c = 1; c <= 500; c++) if (c > threshold) expect += histo_[c - 1];
        if (expect >= 0x7FFFFFFFull) { err = "too many solid k-mers for 32-bit node ids"; return -1; }
        EvTimer t(stream_);
        if (global_mode_) {
            if (int rc = compact_into(threshold, expect, skeys_, scnt_, err)) return rc;
        } else if (threshold == emit_threshold_ || n_emitted_ == 0) {
            if (n_emitted_ != expect && threshold == emit_threshold_) { err = "emitted row count disagrees with the histogram"; return -6; }
            for (int j = 0; j < W; j++) skeys_[j].swap(ekeys_[j]);
            scnt_.swap(ecnt_);
            for (int j = 0; j < W; j++) ekeys_[j].release();
            ecnt_.release();
        } else {
            if (threshold < emit_threshold_) { err = "filter threshold below the emit threshold"; return -6; }
            for (int j = 0; j < W; j++) if (int rc = skeys_[j].alloc(expect, err)) return rc;
            if (int rc = scnt_.alloc(expect, err)) return rc;
            HIPCHK(hipMemsetAsync(ctl_.p, 0, 16 * sizeof(unsigned long long), stream_));
            KeyArr<W> ik, ok;
            for (int j = 0; j < W; j++) { ik.w[j] = ekeys_[j].p; ok.w[j] = skeys_[j].p; }
            hipLaunchKernelGGL(k_compact_rows<W>, dim3(grid_for(n_emitted_)), dim3(256), 0, stream_, ik, ecnt_.p,
                               n_emitted_, threshold, ok, scnt_.p, ctl_.p + 0);
            HIPCHK(hipGetLastError());
            unsigned long long got = 0;
            HIPCHK(hipMemcpyAsync(&got, ctl_.p, 8, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            if (got != expect) { err = "row compaction count mismatch"; return -6; }
            for (int j = 0; j < W; j++) ekeys_[j].release();
            ecnt_.release();
        }
        times_.add("filter_kernel", t.stop());
        n_solid_ = expect;
        graph_ready_ = false;
        return 0;
    }

    int copy_out(DevBuf<uint64_t> (&keys)[W], DevBuf<uint32_t> &cnt, uint64_t n, uint64_t *hk, uint32_t *hc,
                 std::string &err) {
        std::vector<uint64_t> tmp(n ? n : 1);
        for (int j = 0; j < W; j++) {
            if (n) HIPCHK(hipMemcpy(tmp.data(), keys[j].p, n * 8, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < n; i++) hk[i * W + j] = tmp[i];
        }
        if (n) HIPCHK(hipMemcpy(hc, cnt.p, n * 4, hipMemcpyDeviceToHost));
        return 0;
    }

    int get_distinct(uint64_t *keys, uint32_t *counts, uint64_t cap, std::string &err) override {
        if (cap < n_distinct_) { err = "buffer too small"; return -1; }
        DevBuf<uint64_t> dk[W]; DevBuf<uint32_t> dc;
        if (global_mode_) {
            if (int rc = compact_into(0, n_distinct_, dk, dc, err)) return rc;
        } else if (n_distinct_) {
            // stage inspection: run the counting pass again keeping every row
            if (!have_parts_ || (!recs_.p && batches_.empty())) { err = "partition buffers already released"; return -2; }
            uint64_t rows = 0, hist[500], inst = 0; double ms = 0;
            if (int rc = run_count_partitions(run_view_, n_count_parts_, 0, dk, dc, rows, hist, inst, n_distinct_, ms, err)) return rc;
            if (rows != n_distinct_) { err = "distinct row count mismatch"; return -6; }
        }
        return copy_out(dk, dc, n_distinct_, keys, counts, err);
    }
    int get_solid(uint64_t *keys, uint32_t *counts, uint64_t cap, std::string &err) override {
        if (cap < n_solid_) { err = "buffer too small"; return -1; }
        return copy_out(skeys_, scnt_, n_solid_, keys, counts, err);
    }

    // ---- shard layer (one process per GPU; DESIGN.md "Multi-GPU") --------------------------------
    int shard_partition(const uint32_t *d_bases, const uint32_t *d_seg_off, uint64_t n_seg, uint64_t n_bases,
                        uint32_t n_partitions, std::vector<uint64_t> &part_records, std::string &err) override {
        if (global_mode_) { err = "shard layer needs the partitioned counting mode"; return -1; }
        if (n_partitions < 1 || n_partitions > (uint32_t)PART_MAX_P || (n_partitions & (n_partitions - 1))) {
            err = "n_partitions must be a power of two <= 16384"; return -1;
        }
        forced_P_ = n_partitions;
        part_records.assign(n_partitions, 0);
        if (int rc = count_batch(d_bases, d_seg_off, n_seg, n_bases, err)) return rc;
        if (!have_parts_) return 0;                       // no segments on this rank
        DevBuf<unsigned long long> tot;
        if (int rc = tot.alloc(pp_.P, err)) return rc;
        hipLaunchKernelGGL(k_part_totals, dim3(pp_.P), dim3(256), 0, stream_, fill_.p, pp_, tot.p);
        HIPCHK(hipGetLastError());
        std::vector<unsigned long long> h(pp_.P);
        HIPCHK(hipMemcpyAsync(h.data(), tot.p, (size_t)pp_.P * 8, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        for (uint32_t p = 0; p < pp_.P; p++) part_records[p] = h[p];
        return 0;
    }
    uint32_t rec_words() const override { return 2 * W; }

    int shard_pack(void *d_send, const uint64_t *base_records, uint32_t n_partitions, std::string &err) override {
        if (!have_parts_) return 0;
        if (n_partitions != pp_.P) { err = "partition count mismatch"; return -1; }
        DevBuf<unsigned long long> base;
        if (int rc = base.alloc(pp_.P, err)) return rc;
        HIPCHK(hipMemcpyAsync(base.p, base_records, (size_t)pp_.P * 8, hipMemcpyHostToDevice, stream_));
        EvTimer t(stream_);
        hipLaunchKernelGGL((k_pack_partition<2 * W>), dim3(pp_.P), dim3(256), 0, stream_, recs_.p, fill_.p, pp_, base.p,
                           (uint64_t *)d_send);
        HIPCHK(hipGetLastError());
        times_.add("shard_pack_kernel", t.stop());
        HIPCHK(hipStreamSynchronize(stream_));
        // the local slices are no longer needed once packed
        recs_.release(); fill_.release(); run_off_.release(); run_cnt_.release(); have_parts_ = false;
        return 0;
    }

    // d_recv: records received from all sources; run tables [n_owned][n_sources] on the host
    int shard_count(const void *d_recv, const uint64_t *run_off, const uint32_t *run_cnt, uint32_t n_owned,
                    uint32_t n_sources, uint32_t emit_threshold, uint64_t histo[500], std::string &err) override {
        if (n_sources < 1 || n_sources > 256) { err = "1..256 sources"; return -1; }
        const uint64_t n_runs = (uint64_t)n_owned * n_sources;
        if (int rc = run_off_.alloc(n_runs, err)) return rc;
        if (int rc = run_cnt_.alloc(n_runs, err)) return rc;
        std::vector<unsigned long long> addr16(n_runs);
        for (uint64_t i = 0; i < n_runs; i++) addr16[i] = ((unsigned long long)(uintptr_t)d_recv >> 4) + run_off[i] * (unsigned long long)W;   // RW/2 = W
        if (n_runs) {
            HIPCHK(hipMemcpyAsync(run_off_.p, addr16.data(), n_runs * 8, hipMemcpyHostToDevice, stream_));
            HIPCHK(hipMemcpyAsync(run_cnt_.p, run_cnt, n_runs * 4, hipMemcpyHostToDevice, stream_));
        }
        shard_recv_ = d_recv;
        run_view_.run_addr16 = run_off_.p; run_view_.run_cnt = run_cnt_.p;
        run_view_.S = n_sources; run_view_.k = k_; n_count_parts_ = n_owned; run_view_.dbg = 0;
        uint64_t total_recs = 0;
        for (uint64_t i = 0; i < n_runs; i++) total_recs += run_cnt[i];
        memset(histo, 0, 500 * 8);
        n_emitted_ = 0; emit_threshold_ = emit_threshold; n_distinct_ = 0;
        double ms = 0; uint64_t inst = 0;
        have_parts_ = n_runs != 0;
        if (int rc = run_count_partitions(run_view_, n_owned, emit_threshold, ekeys_, ecnt_, n_emitted_, histo, inst,
                                          std::max<uint64_t>(1u << 16, total_recs), ms, err)) return rc;
        times_.add("count_kernel", ms);
        total_instances_ = inst;
        for (int i = 0; i < 500; i++) { histo_[i] = histo[i]; n_distinct_ += histo[i]; }
        return 0;
    }

    // local rows with count > threshold; device pointers stay owned by the pipeline
    int shard_rows(uint32_t threshold, const void **keys_soa, const void **cnt, uint64_t *n, std::string &err) override {
        if (int rc = filter(threshold, err)) return rc;
        for (int j = 0; j < W; j++) keys_soa[j] = skeys_[j].p;
        *cnt = scnt_.p; *n = n_solid_;
        return 0;
    }

    // install the gathered solid set (every rank holds all of it) and the global statistics
    int shard_set_solid(const void *const *keys_soa, const void *cnt, uint64_t n, const uint64_t histo[500],
                        uint64_t total_instances, std::string &err) override {
        if (n >= 0x7FFFFFFFull) { err = "too many solid k-mers for 32-bit node ids"; return -1; }
        DevBuf<uint64_t> nk[W]; DevBuf<uint32_t> nc;
        for (int j = 0; j < W; j++) {
            if (int rc = nk[j].alloc(n, err)) return rc;
            if (n) HIPCHK(hipMemcpyAsync(nk[j].p, keys_soa[j], n * 8, hipMemcpyDeviceToDevice, stream_));
        }
        if (int rc = nc.alloc(n, err)) return rc;
        if (n) HIPCHK(hipMemcpyAsync(nc.p, cnt, n * 4, hipMemcpyDeviceToDevice, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        for (int j = 0; j < W; j++) skeys_[j].swap(nk[j]);
        scnt_.swap(nc);
        n_solid_ = n; total_instances_ = total_instances; n_distinct_ = 0;
        for (int i = 0; i < 500; i++) { histo_[i] = histo[i]; n_distinct_ += histo[i]; }
        graph_ready_ = false;
        return 0;
    }

    // ---- graph -----------------------------------------------------------------------------
    Graph<W> graph_view() {
        Graph<W> g;
        for (int j = 0; j < W; j++) g.keys.w[j] = skeys_[j].p;
        g.cnt = scnt_.p; g.adj = adj_.p; g.nb = nb_.p; g.k = k_;
        g.gt.e = gt_.p; g.gt.off = gt_off_.p; g.gt.msk = gt_msk_.p; g.gt.gp_mask = gp_ - 1u; g.gt.gm = part_m(k_);
        g.n = (uint32_t)n_solid_;
        return g;
    }

    int build_graph(std::string &err) override {
        const uint64_t n = n_solid_;
        // count table is no longer needed once the solid set exists
        for (int j = 0; j < W; j++) tkeys_[j].release();
        tcnt_.release(); tstate_.release(); tslots_ = 0;
        recs_.release(); fill_.release(); run_off_.release(); run_cnt_.release(); shard_recv_ = nullptr; batches_.clear();
        // graph partitions: 256-512 rows each (mini tables of <= 2048 slots fit 16 KB of LDS); the minimiser length is the counting pass's, so rows that
        // arrive grouped by counting partition are grouped by graph partition too
        gp_ = 64;
        while (gp_ < 131072u && (uint64_t)gp_ * 512u < n) gp_ <<= 1;
        gt_slots_ = 4 * n + 8ull * gp_;               // >= sum of max(8, pow2 >= 2 x rows)
        if (int rc = gt_.alloc(gt_slots_, err)) return rc;
        if (int rc = gt_off_.alloc(gp_, err)) return rc;
        if (int rc = gt_msk_.alloc(gp_, err)) return rc;
        DevBuf<uint32_t> gp_of, gp_cnt, gp_roff, gp_rows;
        if (int rc = gp_of.alloc(n, err)) return rc;
        if (int rc = gp_cnt.alloc(gp_, err)) return rc;
        if (int rc = gp_roff.alloc(gp_ + 1, err)) return rc;
        if (int rc = gp_rows.alloc(n, err)) return rc;
        DevBuf<unsigned long long> queries;              // 8 slots per row, grouped by partition; only a prefix is touched
        if (int rc = queries.alloc(8 * n + 8, err)) return rc;
        if (int rc = adj_.alloc((n + 8) & ~3ull, err)) return rc;
        if (int rc = adj0_.alloc(n, err)) return rc;
        if (int rc = nb_.alloc(2 * n + 2, err)) return rc;
        if (int rc = alive_.alloc(n, err)) return rc;
        HIPCHK(hipMemsetAsync(gp_cnt.p, 0, (size_t)gp_ * 4, stream_));   // (the mini tables are initialised by their builders)
        HIPCHK(hipMemsetAsync(adj_.p, 0, adj_.n, stream_));
        HIPCHK(hipMemsetAsync(alive_.p, 1, n ? n : 1, stream_));
        HIPCHK(hipMemsetAsync(ctl_.p, 0, 16 * sizeof(unsigned long long), stream_));
        if (n) {
            Graph<W> g = graph_view();
            EvTimer t(stream_);
            hipLaunchKernelGGL(k_gp_count<W>, dim3(grid_for(n)), dim3(256), 0, stream_, g.keys, (uint32_t)n, k_, g.gt,
                               gp_of.p, gp_cnt.p);
            hipLaunchKernelGGL(k_gp_scan, dim3(1), dim3(1024), 0, stream_, gp_cnt.p, gp_, gt_off_.p, gt_msk_.p, gp_roff.p,
                               ctl_.p + 2);
            HIPCHK(hipMemsetAsync(gp_cnt.p, 0, (size_t)gp_ * 4, stream_));          // reused as the row-list cursors
            hipLaunchKernelGGL(k_gp_rows, dim3(grid_for(n)), dim3(256), 0, stream_, gp_of.p, (uint32_t)n, gp_roff.p, gp_cnt.p,
                               gp_rows.p);
            HIPCHK(hipGetLastError());
            times_.add("graph_table_kernel", t.stop());
            EvTimer t2(stream_);
            hipLaunchKernelGGL(k_graph_local<W>, dim3(gp_), dim3(256), 0, stream_, g.keys, k_, g.gt, gp_roff.p, gp_rows.p,
                               adj_.p, nb_.p, queries.p, gp_cnt.p, (uint32_t *)(ctl_.p + 1));
            hipLaunchKernelGGL(k_graph_remote<W>, dim3(gp_), dim3(256), 0, stream_, g.keys, k_, g.gt, gp_roff.p, queries.p,
                               gp_cnt.p, adj_.p, nb_.p);
            HIPCHK(hipGetLastError());
            times_.add("adjacency_kernel", t2.stop());
            HIPCHK(hipMemcpyAsync(adj0_.p, adj_.p, n, hipMemcpyDeviceToDevice, stream_));
            unsigned long long h[3];
            HIPCHK(hipMemcpyAsync(h, ctl_.p, sizeof h, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            if ((uint32_t)h[1] || h[2] > gt_slots_) { err = "graph table overflow"; return -6; }
        }
        graph_ready_ = true;
        return 0;
    }

    int read_ctl(unsigned int &v, int slot, std::string &err) {
        unsigned long long h = 0;
        HIPCHK(hipMemcpyAsync(&h, ctl_.p + slot, 8, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        v = (unsigned int)h;
        return 0;
    }

    // marked alive nodes -> removed list (count at ctl_[slot]) -> their edges cleared in the neighbours
    int apply_marks(Graph<W> &g, DevBuf<uint8_t> &mark, DevBuf<uint32_t> &removed, int slot, std::string &err) {
        const uint32_t n = (uint32_t)n_solid_;
        hipLaunchKernelGGL(k_collect_marked, dim3(grid_for(n)), dim3(256), 0, stream_, n, mark.p, alive_.p,
                           removed.p, (unsigned int *)(ctl_.p + slot));
        hipLaunchKernelGGL(k_apply_removed<W>, dim3(1024), dim3(256), 0, stream_, g, removed.p,
                           (const unsigned int *)(ctl_.p + slot));
        HIPCHK(hipGetLastError());
        return 0;
    }

    int correct(bool tips, bool bubbles, std::string &err) override {
        if (!graph_ready_) { err = "graph not built"; return -2; }
        const uint32_t n = (uint32_t)n_solid_;
        tips_removed_ = bubbles_removed_ = 0; rounds_ = 0;
        if (n == 0 || (!tips && !bubbles)) return 0;
        Graph<W> g = graph_view();
        DevBuf<uint32_t> cand, tip_head, removed;
        DevBuf<uint8_t> mark, kill;
        DevBuf<TipRec> tiprec;
        if (int rc = cand.alloc(2ull * n, err)) return rc;
        if (int rc = removed.alloc(n, err)) return rc;
        if (int rc = mark.alloc(n, err)) return rc;
        HIPCHK(hipMemsetAsync(mark.p, 0, n, stream_));
        if (tips) {
            // every candidate may turn out to be a tip: sized for all oriented nodes, so that no count has to
            // come back to the host inside a round (the counters live in ctl_: 3 candidates, 4 tips, 5/6 removed)
            if (int rc = tip_head.alloc(2ull * n, err)) return rc;
            if (int rc = tiprec.alloc(2ull * n, err)) return rc;
            if (int rc = kill.alloc(2ull * n, err)) return rc;
            HIPCHK(hipMemsetAsync(tip_head.p, 0xFF, 2ull * n * 4, stream_));
        }
        EvTimer t(stream_);
        const dim3 G(1024), B(256);
        for (int round = 0; round < 32; round++) {
            HIPCHK(hipMemsetAsync(ctl_.p + 3, 0, 4 * 8, stream_));
            if (tips) {
                hipLaunchKernelGGL(k_tip_candidates<W>, dim3(grid_for(2ull * n)), B, 0, stream_, g, alive_.p,
                                   cand.p, (unsigned int *)(ctl_.p + 3));
                hipLaunchKernelGGL(k_tip_walk<W>, G, B, 0, stream_, g, cand.p, (const unsigned int *)(ctl_.p + 3),
                                   tiprec.p, (unsigned int *)(ctl_.p + 4), tip_head.p);
                hipLaunchKernelGGL(k_tip_decide<W>, G, B, 0, stream_, g, tiprec.p, (const unsigned int *)(ctl_.p + 4),
                                   tip_head.p, kill.p);
                hipLaunchKernelGGL(k_tip_remove<W>, G, B, 0, stream_, g, tiprec.p, (const unsigned int *)(ctl_.p + 4),
                                   kill.p, tip_head.p, mark.p);
                hipLaunchKernelGGL(k_tip_reset_heads, G, B, 0, stream_, tiprec.p, (const unsigned int *)(ctl_.p + 4),
                                   tip_head.p);
                HIPCHK(hipGetLastError());
                if (int rc = apply_marks(g, mark, removed, 5, err)) return rc;
            }
            if (bubbles) {
                HIPCHK(hipMemsetAsync(ctl_.p + 3, 0, 8, stream_));
                hipLaunchKernelGGL(k_fork_candidates<W>, dim3(grid_for(2ull * n)), B, 0, stream_, g, alive_.p,
                                   cand.p, (unsigned int *)(ctl_.p + 3));
                hipLaunchKernelGGL(k_bubble<W>, G, B, 0, stream_, g, cand.p, (const unsigned int *)(ctl_.p + 3), mark.p);
                HIPCHK(hipGetLastError());
                if (int rc = apply_marks(g, mark, removed, 6, err)) return rc;
            }
            unsigned long long h[2];
            HIPCHK(hipMemcpyAsync(h, ctl_.p + 5, sizeof h, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            const unsigned int n1 = (unsigned int)h[0], n2 = (unsigned int)h[1];
            tips_removed_ += n1; bubbles_removed_ += n2; rounds_++;
            if (n1 + n2 == 0) break;
        }
        times_.add("correct_total", t.stop());
        return 0;
    }

    int get_adjacency(uint8_t *adj_initial, uint8_t *adj_final, uint8_t *alive, uint64_t cap,
                      std::string &err) override {
        if (!graph_ready_) { err = "graph not built"; return -2; }
        if (cap < n_solid_) { err = "buffer too small"; return -1; }
        if (!n_solid_) return 0;
        if (adj_initial) HIPCHK(hipMemcpy(adj_initial, adj0_.p, n_solid_, hipMemcpyDeviceToHost));
        if (adj_final) HIPCHK(hipMemcpy(adj_final, adj_.p, n_solid_, hipMemcpyDeviceToHost));
        if (alive) HIPCHK(hipMemcpy(alive, alive_.p, n_solid_, hipMemcpyDeviceToHost));
        return 0;
    }

    // ---- collapse ----------------------------------------------------------------------------
    int collapse(std::vector<RawContig> &out, std::string &err) override {
        out.clear();
        if (!graph_ready_) { err = "graph not built"; return -2; }
        const uint32_t n = (uint32_t)n_solid_;
        if (n == 0) return 0;
        const uint32_t total = 2 * n;
        Graph<W> g = graph_view();
        DevBuf<uint32_t> spl;
        DevBuf<uint2> winfo, ol;
        DevBuf<SegRec> segs;
        if (int rc = winfo.alloc(total, err)) return rc;
        if (int rc = spl.alloc(total, err)) return rc;
        if (int rc = ol.alloc(total, err)) return rc;
        HIPCHK(hipMemsetAsync(ctl_.p + 5, 0, 8, stream_));
        const uint32_t split_mask = (1u << (uint32_t)env_u64("SHK_SPLIT_LOG", SPLIT_LOG_DEFAULT)) - 1u;
        EvTimer t1(stream_);
        hipLaunchKernelGGL(k_succ_split<W>, dim3((total + 256 * SS_ITEMS - 1) / (256 * SS_ITEMS)), dim3(256), 0, stream_, g,
                           alive_.p, winfo.p, spl.p, ol.p, (unsigned int *)(ctl_.p + 5), split_mask);
        HIPCHK(hipGetLastError());
        unsigned int n_spl = 0;
        if (int rc = read_ctl(n_spl, 5, err)) return rc;
        times_.add("collapse_succ_split", t1.stop());
        times_.add("collapse_n_splitters_x1e-3", n_spl * 1e-3);
        if (n_spl) {
            if (int rc = segs.alloc(n_spl, err)) return rc;
            EvTimer t2(stream_);
            hipLaunchKernelGGL(k_walk_segments<W>, dim3(grid_for(n_spl, 256, 1 << 20)), dim3(256), 0, stream_, winfo.p,
                               spl.p, n_spl, ol.p, segs.p, split_mask);
            HIPCHK(hipGetLastError());
            times_.add("collapse_walk", t2.stop());
        }
        // ---- rank the splitter list on the device: prefix of segment lengths by pointer jumping
        DevBuf<uint32_t> Pa, Pb, Aa, Ab, slot_of; DevBuf<unsigned long long> Ka, Kb, d_off; DevBuf<HeadRec> d_heads;
        DevBuf<char> d_out;
        uint32_t *Pf = nullptr, *Af = nullptr;
        std::vector<HeadRec> heads;
        uint64_t covered = 0;
        if (n_spl) {
            if (int rc = Pa.alloc(n_spl, err)) return rc;
            if (int rc = Pb.alloc(n_spl, err)) return rc;
            if (int rc = Aa.alloc(n_spl, err)) return rc;
            if (int rc = Ab.alloc(n_spl, err)) return rc;
            if (int rc = Ka.alloc(n_spl, err)) return rc;
            if (int rc = Kb.alloc(n_spl, err)) return rc;
            if (int rc = slot_of.alloc(n_spl, err)) return rc;
            if (int rc = d_heads.alloc(n_spl, err)) return rc;
            EvTimer tr(stream_);
            // cycle members have no head: give every element a defined pointer first
            HIPCHK(hipMemsetAsync(Pa.p, 0, (size_t)n_spl * 4, stream_));
            HIPCHK(hipMemsetAsync(Aa.p, 0, (size_t)n_spl * 4, stream_));
            HIPCHK(hipMemsetAsync(Ka.p, 0, (size_t)n_spl * 8, stream_));
            HIPCHK(hipMemsetAsync(slot_of.p, 0xFF, (size_t)n_spl * 4, stream_));
            HIPCHK(hipMemsetAsync(ctl_.p + 6, 0, 8, stream_));
            const int gr = grid_for(n_spl);
            hipLaunchKernelGGL(k_rank_init, dim3(gr), dim3(256), 0, stream_, segs.p, n_spl, Pa.p, Aa.p, Ka.p);
            uint32_t *Pi = Pa.p, *Po = Pb.p, *Ai = Aa.p, *Ao = Ab.p; unsigned long long *Ki = Ka.p, *Ko = Kb.p;
            int rounds = 1; while ((1ull << rounds) < (uint64_t)n_spl) rounds++;
            for (int r = 0; r < rounds; r++) {
                hipLaunchKernelGGL(k_rank_jump, dim3(gr), dim3(256), 0, stream_, n_spl, Pi, Ai, Ki, Po, Ao, Ko);
                std::swap(Pi, Po); std::swap(Ai, Ao); std::swap(Ki, Ko);
            }
            Pf = Pi; Af = Ai;
            hipLaunchKernelGGL(k_rank_tails<W>, dim3(gr), dim3(256), 0, stream_, g, segs.p, n_spl, Pi, Ai, Ki, d_heads.p,
                               slot_of.p, (unsigned int *)(ctl_.p + 6));
            HIPCHK(hipGetLastError());
            unsigned int n_heads = 0;
            if (int rc = read_ctl(n_heads, 6, err)) return rc;
            times_.add("collapse_rank_device", tr.stop());
            heads.resize(n_heads);
            if (n_heads) HIPCHK(hipMemcpy(heads.data(), d_heads.p, (size_t)n_heads * sizeof(HeadRec), hipMemcpyDeviceToHost));
        }
        // each unitig exists on both strands: keep the canonical one (decided on the device)
        std::vector<unsigned long long> head_off(heads.size(), ~0ull);
        std::vector<uint32_t> emitted;
        uint64_t out_bytes = 0;
        for (size_t i = 0; i < heads.size(); i++) {
            covered += heads[i].len;
            if (heads[i].emit) {
                head_off[i] = out_bytes; out_bytes += heads[i].len + (uint64_t)(k_ - 1); emitted.push_back((uint32_t)i);
            }
        }
        if (!emitted.empty()) {
            PinnedBuf &hout = hout_;
            if (int rc = hout.alloc(out_bytes, err)) return rc;
            if (int rc = d_off.alloc(head_off.size(), err)) return rc;
            if (int rc = d_out.alloc(out_bytes, err)) return rc;
            HIPCHK(hipMemcpyAsync(d_off.p, head_off.data(), head_off.size() * 8, hipMemcpyHostToDevice, stream_));
            EvTimer t3(stream_);
            hipLaunchKernelGGL(k_emit<W>, dim3(grid_for(total)), dim3(256), 0, stream_, g, alive_.p, ol.p,
                               Pf, Af, slot_of.p, d_off.p, d_out.p);
            HIPCHK(hipGetLastError());
            times_.add("collapse_emit", t3.stop());
            auto tcp = std::chrono::steady_clock::now();
            HIPCHK(hipMemcpyAsync(hout.p, d_out.p, out_bytes, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            out.reserve(emitted.size());
            for (uint32_t i : emitted) {
                RawContig rc; rc.kc = heads[i].kc;
                rc.ext = hout.p + head_off[i]; rc.ext_n = heads[i].len + (uint64_t)(k_ - 1);
                out.push_back(std::move(rc));
            }
            times_.add("collapse_d2h_contigs_host_clock", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tcp).count());
        }
        // ---- circular unitigs (no head): rare; resolved on the host from succ[] (SPEC S10)
        const uint64_t n_alive = (uint64_t)n - tips_removed_ - bubbles_removed_;
        if (covered != 2 * n_alive) {
            std::vector<uint8_t> halive(n);
            HIPCHK(hipMemcpy(halive.data(), alive_.p, n, hipMemcpyDeviceToHost));
            auto tc0 = std::chrono::steady_clock::now();
            std::vector<uint32_t> hsucc(total), hcnt(n);
            std::vector<uint64_t> hkeys((size_t)n * W);
            {
                std::vector<uint2> hw(total);
                HIPCHK(hipMemcpy(hw.data(), winfo.p, (size_t)total * sizeof(uint2), hipMemcpyDeviceToHost));
                for (uint32_t v = 0; v < total; v++) hsucc[v] = hw[v].x;
            }
            HIPCHK(hipMemcpy(hcnt.data(), scnt_.p, (size_t)n * 4, hipMemcpyDeviceToHost));
            {
                std::vector<uint64_t> tmp(n);
                for (int j = 0; j < W; j++) {
                    HIPCHK(hipMemcpy(tmp.data(), skeys_[j].p, (size_t)n * 8, hipMemcpyDeviceToHost));
                    for (uint32_t i = 0; i < n; i++) hkeys[(size_t)i * W + j] = tmp[i];
                }
            }
            // nodes on headed chains: walk them again on the host to mark coverage
            std::vector<SegRec> hseg(n_spl);
            std::vector<uint32_t> hP(n_spl);
            if (n_spl) {
                HIPCHK(hipMemcpy(hseg.data(), segs.p, (size_t)n_spl * sizeof(SegRec), hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(hP.data(), Pf, (size_t)n_spl * 4, hipMemcpyDeviceToHost));
            }
            std::vector<uint8_t> on_chain(n, 0);
            for (uint32_t i = 0; i < n_spl; i++) {
                const uint32_t root = hP[i];
                if (!(root < n_spl && hP[root] == root && hseg[root].head)) continue;   // splitter on a cycle
                uint32_t v = hseg[i].node;
                for (uint32_t j = 0; j < hseg[i].len; j++) { on_chain[v >> 1] = 1; v = hsucc[v]; }
            }
            auto key_of = [&](uint32_t idx) { Kmer<W> x; for (int j = 0; j < W; j++) x.w[j] = hkeys[(size_t)idx * W + j]; return x; };
            std::vector<uint8_t> done(n, 0);
            for (uint32_t i = 0; i < n; i++) {
                if (!halive[i] || on_chain[i] || done[i]) continue;
                // find the smallest canonical k-mer on this cycle (either strand holds the same nodes)
                uint32_t v = i * 2, best = i; uint64_t len = 0;
                do {
                    if (km_less<W>(key_of(v >> 1), key_of(best))) best = v >> 1;
                    v = hsucc[v]; len++;
                    if (v == NIL || len > (uint64_t)total) { err = "collapse: broken cycle"; return -6; }
                } while (v != i * 2);
                RawContig rc; rc.kc = 0;
                const char B[4] = {'A', 'C', 'G', 'T'};
                v = best * 2;
                Kmer<W> x = key_of(best);
                for (int q = 0; q < k_; q++) rc.own.push_back(B[km_bits2<W>(x, 2 * (k_ - 1 - q))]);
                for (uint64_t q = 0; q < len; q++) {
                    done[v >> 1] = 1; rc.kc += hcnt[v >> 1];
                    if (q > 0) {
                        Kmer<W> y = key_of(v >> 1);
                        if (v & 1) y = km_revcomp<W>(y, k_);
                        rc.own.push_back(B[km_last_base<W>(y)]);
                    }
                    v = hsucc[v];
                }
                out.push_back(std::move(rc));
            }
            times_.add("collapse_host_cycles", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count());
        }
        return 0;
    }

private:
    int k_;
    hipStream_t stream_ = nullptr; int stream_dev_ = 0;
    StageTimes times_;
    DevBuf<unsigned long long> ctl_;
    // count table
    DevBuf<uint64_t> tkeys_[W]; DevBuf<uint32_t> tcnt_, tstate_; uint64_t tslots_ = 0;
    struct Batch { const uint32_t *bases, *seg_off; uint64_t n_seg; };
    std::vector<Batch> pending_;
    uint64_t total_instances_ = 0, n_distinct_ = 0, n_solid_ = 0;
    uint64_t histo_[500] = {0};
    // partitioned counting
    bool global_mode_ = env_u64("SHK_COUNT_MODE_GLOBAL", 0) != 0;
    bool have_parts_ = false;
    PartParams pp_{};
    DevBuf<uint64_t> recs_; DevBuf<uint32_t> fill_;
    DevBuf<unsigned long long> run_off_; DevBuf<uint32_t> run_cnt_;
    RunView run_view_{}; uint32_t n_count_parts_ = 0; uint32_t forced_P_ = 0;
    std::vector<std::unique_ptr<BatchRecs>> batches_;
    const void *shard_recv_ = nullptr;
    DevBuf<uint64_t> ekeys_[W]; DevBuf<uint32_t> ecnt_;
    uint64_t n_emitted_ = 0; uint32_t emit_threshold_ = 0;
    // solid set / graph
    DevBuf<uint64_t> skeys_[W]; DevBuf<uint32_t> scnt_;
    DevBuf<uint64_t> gt_; uint64_t gt_slots_ = 0; uint32_t gp_ = 64;
    DevBuf<unsigned long long> gt_off_; DevBuf<uint32_t> gt_msk_;
    DevBuf<uint8_t> adj_, adj0_, alive_;
    PinnedBuf hout_;                  // contigs as downloaded; RawContig::ext points into it
    DevBuf<uint32_t> nb_;
    bool graph_ready_ = false;
    uint64_t tips_removed_ = 0, bubbles_removed_ = 0; int rounds_ = 0;
};

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

IPipeline *make_pipeline(int k, std::string &err) {
    if (device_count() <= 0) { err = "no HIP device available (libshk_hip has no CPU fallback)"; return nullptr; }
    const int W = (2 * k + 63) / 64;
    int rc = -1;
    IPipeline *p = nullptr;
    if (W == 1) { auto *q = new Pipeline<1>(k); rc = q->init(err); p = q; }
    else if (W == 2) { auto *q = new Pipeline<2>(k); rc = q->init(err); p = q; }
    else if (W == 3) { auto *q = new Pipeline<3>(k); rc = q->init(err); p = q; }
    else if (W == 4) { auto *q = new Pipeline<4>(k); rc = q->init(err); p = q; }
    else { err = "k too large for the compiled key widths"; return nullptr; }
    if (rc != 0) { delete p; return nullptr; }
    return p;
}

int device_upload(const void *host, size_t bytes, void **dptr, std::string &err) {
    *dptr = nullptr;
    HIPCHK(hipMalloc(dptr, bytes ? bytes : 4));
    if (bytes) {
        hipError_t e = hipMemcpy(*dptr, host, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(*dptr); *dptr = nullptr; err = hipGetErrorString(e); return -5; }
    }
    return 0;
}
void device_free(void *dptr) { if (dptr) (void)hipFree(dptr); }

// ---- host-side self-test helpers (same arithmetic as the kernels) ---------------------------
template <int W> static int host_canon_t(const char *seq, uint32_t k, uint64_t *out, int *orient) {
    Kmer<W> f = km_zero<W>();
    for (uint32_t i = 0; i < k; i++) {
        uint32_t b;
        switch (seq[i]) { case 'A': case 'a': b = 0; break; case 'C': case 'c': b = 1; break;
                          case 'G': case 'g': b = 2; break; case 'T': case 't': b = 3; break; default: return -1; }
        km_push_back<W>(f, b, (int)k);
    }
    int o; Kmer<W> c = km_canonical<W>(f, (int)k, o);
    for (int j = 0; j < W; j++) out[j] = c.w[j];
    if (orient) *orient = o;
    return 0;
}
int host_canonical(const char *seq, uint32_t k, uint64_t *out, int *orient) {
    const int W = (2 * k + 63) / 64;
    if (W == 1) return host_canon_t<1>(seq, k, out, orient);
    if (W == 2) return host_canon_t<2>(seq, k, out, orient);
    if (W == 3) return host_canon_t<3>(seq, k, out, orient);
    if (W == 4) return host_canon_t<4>(seq, k, out, orient);
    return -1;
}
uint64_t host_nthash(const char *seq, uint32_t k) {
    // roll across the first k bases exactly as the kernel does
    NtState nt{0, 0};
    for (uint32_t i = 0; i < k; i++) {
        uint32_t b = seq[i] == 'A' ? 0 : seq[i] == 'C' ? 1 : seq[i] == 'G' ? 2 : 3;
        nt_init_step(nt, b, i);
    }
    return nt_canonical(nt);
}

}  // namespace shk
