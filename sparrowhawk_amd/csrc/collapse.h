// collapse.h — a12 assembly:collapse_graph (SPEC S10): simple links, splitters, walkers, list ranking, emission
// (included by pipeline.hip inside namespace shk, after the device-side views and count_part.h)
#pragma once

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
// one launch follows HOPS pointers (the reach grows HOPS-fold per launch instead of doubling: a launch
// over ~300 k splitters is all latency, so ceil(log4 n) launches of four dependent reads beat
// ceil(log2 n) launches of two); heads point to themselves with A = K = 0, so overshooting adds nothing
static constexpr int RANK_HOPS = 4;
__global__ __launch_bounds__(256) void k_rank_jump(uint32_t n_spl, const uint32_t *__restrict__ Pi,
                                                   const uint32_t *__restrict__ Ai,
                                                   const unsigned long long *__restrict__ Ki,
                                                   uint32_t *__restrict__ Po, uint32_t *__restrict__ Ao,
                                                   unsigned long long *__restrict__ Ko) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_spl; s += gridDim.x * blockDim.x) {
        uint32_t p = s, a = 0;
        unsigned long long kc = 0;
#pragma unroll
        for (int h = 0; h < RANK_HOPS; h++) { a += Ai[p]; kc += Ki[p]; p = Pi[p]; }
        Po[s] = p; Ao[s] = a; Ko[s] = kc;
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

