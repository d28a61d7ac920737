// collapse.h — a12 assembly:collapse_graph (SPEC S10): simple links, splitters, walkers, list ranking, emission
// (included by pipeline.hip inside namespace shk, after the device-side views and count_part.h)
#pragma once

// ------------------------------------------------------------------------------------------
// a12: collapse (SPEC S10).  Simple links over oriented nodes, splitters = all heads plus a 1/64
// sample; the nodes between two splitters are first contracted into fragments inside LDS tiles (k_local_frag, below),
// one walker per splitter hops over the fragments, the splitter list is ranked by pointer jumping, then every node
// scatters its base into the contig buffer.
//   winfo[v] = {succ(v) or NIL, count(v>>1)}
//   ol[v]    = {head of v's fragment, position in it}; after k_tile_final {chain record, position in the chain}
// A node is sampled by a hash of its ID: a walker decides "is my successor a splitter" from the id it
// just read, without touching the successor (heads are never reached through a simple link: a node
// with a simple predecessor is not a head).
//
// Circular unitigs (a chain of simple links that closes on itself: bacterial chromosomes, plasmids) have no head.
// SPEC S10 cuts them before their smallest k-mer.  The ranking handles them in the same pass as the linear chains:
// next to the usual prefix sums every splitter carries the smallest splitter index in the window of predecessors
// it has seen, with the distance from that splitter to itself.  Once the windows span the ring this is a rank
// relative to the ring's smallest splitter: the ring is "opened" there without a second ranking.  The SPEC's cut
// point is then found by two streaming passes over the nodes (k_ring_min1 / k_ring_min2: coalesced key reads,
// one atomic per wave) that return at once when the graph holds no ring, and the emission rotates the spelling
// by the difference.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool node_sampled(uint32_t v, uint32_t split_mask) {
    return ((mix32(v ^ 0x5bd1e995u) >> 9) & split_mask) == 0;
}

// the most significant 64 bits of the 2k-bit canonical k-mer of node idx (order-preserving prefix; the whole key for k <= 32)
template <int W> __device__ __forceinline__ unsigned long long km_prefix64(const KeyArr<W> &keys, uint32_t idx, int k) {
    const int used = 2 * k - 64 * (W - 1);                 // bits in the top word: 2..64
    const uint64_t hi = keys.w[W - 1][idx];
    if constexpr (W == 1) return hi;
    else return used >= 64 ? hi : ((hi << (64 - used)) | (keys.w[W - 2][idx] >> used));
}
// is the k-mer of oriented node a smaller than that of b?  (equal k-mers: the smaller oriented id)
template <int W> __device__ __forceinline__ bool node_key_less(const Graph<W> &g, uint32_t a, uint32_t b) {
    if ((a >> 1) == (b >> 1)) return a < b;
    const Kmer<W> ka = g.keys.load(a >> 1), kb = g.keys.load(b >> 1);
    if (km_less<W>(ka, kb)) return true;
    if (km_less<W>(kb, ka)) return false;
    return a < b;
}

// ---- fragments: the part of the walk that stays inside a tile of node ids is done in LDS ---------------------------
// The solid rows arrive grouped by minimiser partition, and ~90 % of the simple links join two k-mers of one
// partition (neighbouring k-mers share their minimiser), so most successors of a node sit within a few thousand
// ids of it.  k_local_frag works on tiles of up to LF_TILE oriented nodes: winfo of the tile is read once (coalesced),
// the successors go to LDS, and every FRAGMENT HEAD — a splitter, or a node whose predecessor lies outside the
// tile — walks its fragment there (until the chain ends, leaves the tile or reaches a splitter).  To HBM go
//   ol[v]   = {head of v's fragment, position in the fragment}       (coalesced)
//   frag[h] = {next node after the fragment, nodes, last node, sum of counts, owner splitter, nodes before it in the
//              owner's segment}                                       (one 32-byte record per fragment head)
// and the walkers of k_walk_frags hop from fragment to fragment (one random 32-byte read and one 8-byte write per
// fragment instead of one of each per node).  After the ranking k_tile_final turns ol[v] into {chain record,
// position in the chain}: the per-fragment lookups (owner -> chain, offsets) are done once per fragment head and
// handed to the fragment's nodes through LDS, and the kernels behind it (rings, emission) read ol alone.
// Correct for any row order — a tile that holds no neighbours just makes one-node fragments — fast for the order
// the counting pass produces.
struct FragRec { uint32_t next, len, last, chain_head /* 1: the fragment starts a linear chain */; unsigned long long sum; uint32_t owner, base; };
static_assert(sizeof(FragRec) == 32, "FragRec is one 32-byte record");
static constexpr int LF_THREADS = 1024, LF_ITEMS = 8;
static constexpr uint32_t LF_TILE = LF_THREADS * LF_ITEMS;        // 8192 oriented nodes: 64 KB of LDS, two workgroups per CU
static constexpr uint16_t LF_STOP = 0xFFFFu, LF_DONE = 0xFFFEu;

// v -> (owner splitter, position in its segment); false: no walker owns v (dead, or on a circular unitig without a splitter)
__device__ __forceinline__ bool owner_of(const uint2 *__restrict__ ol, const FragRec *__restrict__ frag, uint32_t v,
                                         uint32_t &owner, uint32_t &pos) {
    const uint2 t = ol[v];
    if (t.x == NIL) return false;
    const uint2 ob = *reinterpret_cast<const uint2 *>(&frag[t.x].owner);
    if (ob.x == NIL) return false;
    owner = ob.x; pos = ob.y + t.y;
    return true;
}

static constexpr int SS_ITEMS = 16;            // oriented nodes per thread of k_succ_split

template <int W>
__global__ __launch_bounds__(256) void k_succ_split(Graph<W> g, const uint8_t *__restrict__ alive,
                                                    uint2 *__restrict__ winfo, uint32_t *__restrict__ spl,
                                                    uint2 *__restrict__ ol, unsigned int *__restrict__ n_spl,
                                                    uint32_t split_mask, unsigned long long *__restrict__ n_alive /* += alive oriented nodes */,
                                                    const unsigned long long *__restrict__ skip = nullptr /* two counters: return at once unless both are 0 */) {
    // (launched behind a correction round whose outcome the host does not know yet: if that round removed nodes the graph
    // is not final and this launch is repeated later — pipeline.hip: rank_chains)
    if (skip && (skip[0] | skip[1])) return;
    __shared__ uint32_t wtot[SS_ITEMS * 4];
    __shared__ uint32_t woff[SS_ITEMS * 4];
    __shared__ uint32_t blk_base, blk_alive;
    if (threadIdx.x == 0) blk_alive = 0;
    __syncthreads();
    uint32_t my_alive = 0;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t total = g.n * 2;
    const uint32_t base = blockIdx.x * (256u * SS_ITEMS);          // even: v and v^1 sit in adjacent lanes
    uint32_t pbits = 0;
    // Everything that depends on v alone is requested at once (alive, count, adjacency byte, unique out-neighbour), then
    // the one dependent read (the neighbour's adjacency byte): two memory round trips per item instead of four nested
    // ones, and two items in flight per thread.
#pragma unroll 2
    for (int it = 0; it < SS_ITEMS; it++) {
        const uint32_t v = base + (uint32_t)it * 256u + threadIdx.x;
        uint32_t s = NIL, c = 0; bool al = false;
        if (v < total) {
            const uint32_t av = g.adj[v >> 1], cv = g.cnt[v >> 1], nbv = g.nb[v];
            al = alive[v >> 1] != 0;
            const uint32_t om = outmask_of(av, v & 1u);
            if (al && __popc(om) == 1) {
                c = cv;
                const uint32_t u = nbv < NB_MULTI ? nbv : g.follow(v, (uint32_t)__ffs((int)om) - 1u);   // (several at build time: look the survivor up)
                // (sharded assembly: a neighbour on another rank ends the LOCAL chain here; the link is stitched across ranks later)
                const bool xr = g.xref && (u & g.xref) && u < NB_MULTI;
                if (!xr && u != NIL && g.indeg(u) == 1 && u != v && u != (v ^ 1u)) s = u;
            } else if (al) c = cv;
            uint2 w; w.x = s; w.y = c; winfo[v] = w;
        }
        my_alive += al ? 1u : 0u;
        const uint32_t sp = (uint32_t)__shfl_xor((int)s, 1);       // succ of the mirror node
        const bool p = al && (sp == NIL || node_sampled(v, split_mask));   // head or sampled
        const unsigned long long m = __ballot(p);
        if (lane == 0) wtot[it * 4 + wid] = (uint32_t)__popcll(m);
        pbits |= (p ? 1u : 0u) << it;
    }
    for (int o = 32; o > 0; o >>= 1) my_alive += __shfl_down(my_alive, o);
    if (lane == 0 && my_alive) atomicAdd(&blk_alive, my_alive);
    __syncthreads();
    if (threadIdx.x == 0 && blk_alive) atomicAdd(n_alive, (unsigned long long)blk_alive);
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
        if (p) {                                                   // (the other entries of ol are written by k_local_frag)
            const uint32_t i = blk_base + woff[it * 4 + wid] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            spl[i] = v;
            uint2 o; o.x = i; o.y = 0; ol[v] = o;
        }
    }
}

// Tiles: nominally LF_ROWS rows, each edge moved forward to the next row that starts a group of rows (row_starts:
// one bit per row, set where the low bits of the minimiser hash change — k_row_starts), so that a tile holds whole
// minimiser partitions when the rows arrive grouped.  Only a hint: any cut is correct.  A tile that ends up
// larger than the LDS arrays is worked off in chunks.
static constexpr uint32_t LF_ROWS = 2560;                 // nominal rows per tile (SHK_TILE_ROWS overrides it: the tests cut small graphs into many tiles)
__device__ __forceinline__ uint32_t next_row_start(const uint32_t *__restrict__ bits, uint32_t n_rows, uint32_t r0, int lane) {
    // wave-wide: the first row >= r0 whose start bit is set, looking 64 words (2048 rows) ahead; r0 if there is none
    if (r0 == 0u || r0 >= n_rows) return r0 < n_rows ? r0 : n_rows;
    const uint32_t nw = (n_rows + 31u) >> 5;
    const uint32_t w = (r0 >> 5) + (uint32_t)lane;
    uint32_t x = w < nw ? bits[w] : 0u;
    if (lane == 0) x &= 0xFFFFFFFFu << (r0 & 31u);
    const unsigned long long m = __ballot(x != 0u);
    if (!m) return r0;
    const int L = __ffsll((long long)m) - 1;
    const uint32_t xl = (uint32_t)__shfl((int)x, L);
    const uint32_t r = (((r0 >> 5) + (uint32_t)L) << 5) + (uint32_t)(__ffs((int)xl) - 1);
    return r < n_rows ? r : n_rows;
}
// (first wave of the workgroup; the caller synchronises)
__device__ __forceinline__ void tile_bounds(const uint32_t *__restrict__ bits, uint32_t n_rows, uint32_t tile_rows, uint32_t *lo, uint32_t *hi) {
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        const uint32_t a = next_row_start(bits, n_rows, blockIdx.x * tile_rows, lane);
        const uint32_t b = blockIdx.x + 1u == gridDim.x ? n_rows : next_row_start(bits, n_rows, (blockIdx.x + 1u) * tile_rows, lane);
        if (lane == 0) { *lo = a; *hi = b; }
    }
}
__global__ __launch_bounds__(256) void k_row_starts(const uint32_t *__restrict__ gp_of, uint32_t n, uint32_t lowmask,
                                                    uint32_t *__restrict__ bits, uint8_t *__restrict__ alive /* := 1 (instead of a fill) */) {
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_round = (n + 63u) & ~63u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool st = i < n && (i == 0u || ((gp_of[i] ^ gp_of[i - 1u]) & lowmask) != 0u);
        if (i < n) alive[i] = 1;
        const unsigned long long m = __ballot(st);
        if ((threadIdx.x & 63) == 0) { bits[i >> 5] = (uint32_t)m; bits[(i >> 5) + 1u] = (uint32_t)(m >> 32); }
    }
}

template <int W>
__global__ __launch_bounds__(LF_THREADS) void k_local_frag(uint32_t n_rows, const uint32_t *__restrict__ row_starts, uint32_t tile_rows,
                                                           const uint8_t *__restrict__ alive,
                                                           const uint2 *__restrict__ winfo, uint2 *__restrict__ ol,
                                                           FragRec *__restrict__ frag, uint32_t split_mask,
                                                           const unsigned long long *__restrict__ skip = nullptr /* as k_succ_split */,
                                                           unsigned int *__restrict__ n_spl_p = nullptr, uint32_t seg_cap = 0, uint32_t *__restrict__ flags = nullptr) {
    if (skip && (skip[0] | skip[1])) return;
    // (k_succ_split is complete: its splitter count is final.  More splitters than the ranking has room for — the host sized
    // it from a bound, without reading the count: the ranking is called off (flag 4, the count beside it) and repeated)
    if (n_spl_p && blockIdx.x == 0 && threadIdx.x == 0 && *n_spl_p > seg_cap) { flags[1] = *n_spl_p; flags[0] = 4; *n_spl_p = 0; }
    __shared__ uint16_t l_succ[LF_TILE];       // local index of the successor; LF_STOP: the fragment ends here; LF_DONE: a walker has passed
    __shared__ uint32_t l_cnt[LF_TILE];        // count; once passed: (local head << 13) | position in the fragment
    __shared__ uint16_t l_heads[LF_TILE];
    __shared__ uint32_t n_heads, t_lo, t_hi;
    tile_bounds(row_starts, n_rows, tile_rows, &t_lo, &t_hi);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t hi = t_hi;
    for (uint32_t c = t_lo; c < hi; c += LF_TILE / 2u) {
        const uint32_t base = 2u * c;                                  // even: v and v^1 sit in adjacent lanes
        const uint32_t size = 2u * (min(c + LF_TILE / 2u, hi) - c);   // oriented nodes of this chunk (<= LF_TILE)
        if (threadIdx.x == 0) n_heads = 0;
        __syncthreads();
#pragma unroll 2
        for (int it = 0; it < LF_ITEMS; it++) {
            const uint32_t j = (uint32_t)it * LF_THREADS + threadIdx.x;
            if ((uint32_t)it * LF_THREADS >= size) break;              // (uniform)
            const uint32_t v = base + j;
            uint2 w; w.x = NIL; w.y = 0; bool al = false;
            if (j < size) { w = winfo[v]; al = alive[v >> 1] != 0; }
            const uint32_t s = w.x;
            const uint32_t sp = (uint32_t)__shfl_xor((int)s, 1);       // succ of the mirror node: its mirror is v's predecessor
            const bool p = al && (sp == NIL || node_sampled(v, split_mask));         // a splitter (head or sampled)
            const bool head = al && (p || ((sp ^ 1u) - base) >= size);               // ... or entered from outside
            if (j < size) {
                l_succ[j] = (s != NIL && (s - base) < size && !node_sampled(s, split_mask)) ? (uint16_t)(s - base) : LF_STOP;
                l_cnt[j] = w.y;
            }
            const unsigned long long m = __ballot(head);
            uint32_t off = 0;
            if (lane == 0 && m) off = atomicAdd(&n_heads, (uint32_t)__popcll(m));
            off = (uint32_t)__shfl((int)off, 0);
            if (head) l_heads[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)j;
        }
        __syncthreads();
        const uint32_t nh = n_heads;
        for (uint32_t h = threadIdx.x; h < nh; h += LF_THREADS) {      // every fragment head walks its fragment in LDS
            const uint32_t j = l_heads[h];
            uint32_t cur = j, pos = 0;
            unsigned long long sum = 0;
            for (;;) {
                const uint32_t nx = l_succ[cur];
                sum += l_cnt[cur];
                l_cnt[cur] = (j << 13) | pos;
                l_succ[cur] = LF_DONE;
                pos++;
                if (nx >= LF_TILE || pos >= LF_TILE) break;            // LF_STOP (or, never: a fragment longer than the tile)
                cur = nx;
            }
            const uint32_t v = base + j;
            FragRec f; f.next = winfo[base + cur].x; f.len = pos; f.last = base + cur; f.sum = sum; f.base = 0;
            const bool chain_head = winfo[v ^ 1u].x == NIL;            // no simple predecessor: the first node of a linear chain
            f.chain_head = chain_head ? 1u : 0u;
            f.owner = (chain_head || node_sampled(v, split_mask)) ? ol[v].x : NIL;   // a splitter's index (k_succ_split)
            frag[v] = f;
        }
        __syncthreads();
#pragma unroll 2
        for (int it = 0; it < LF_ITEMS; it++) {
            const uint32_t j = (uint32_t)it * LF_THREADS + threadIdx.x;
            if (j >= size) break;
            uint2 o; o.x = NIL; o.y = 0;
            if (l_succ[j] == LF_DONE) { const uint32_t e = l_cnt[j]; o.x = base + (e >> 13); o.y = e & (LF_TILE - 1u); }
            ol[base + j] = o;
        }
        __syncthreads();                                               // (the arrays are reused by the next chunk)
    }
}

// head: 0 = has a splitter before it; HEAD_LINEAR = first splitter of a linear chain; HEAD_ORPHAN = a circular
// unitig without any sampled node, spelled from its smallest k-mer (k_orphan_cycles).
struct SegRec { uint32_t node, next_spl, len, last; unsigned long long sum; uint32_t head, pad; };
static constexpr uint32_t HEAD_LINEAR = 1, HEAD_ORPHAN = 2;

// one walker per splitter: from fragment to fragment until the next splitter (a fragment that starts at a splitter
// got its owner in k_succ_split; the others get theirs here)
template <int W>
__global__ __launch_bounds__(256) void k_walk_frags(const uint32_t *__restrict__ spl, const unsigned int *__restrict__ n_spl_p,
                                                    FragRec *__restrict__ frag, SegRec *__restrict__ segs,
                                                    uint32_t split_mask, unsigned long long *__restrict__ n_covered /* += nodes walked */,
                                                    uint32_t total /* oriented nodes */, uint32_t *__restrict__ flags) {
    unsigned long long my_cov = 0;
    const uint32_t n_spl = *n_spl_p;                           // (the count stays on the device; 0 when the ranking was called off)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_spl; i += gridDim.x * blockDim.x) {
        const uint32_t s = spl[i];
        uint32_t cur = s, len = 0, nxt, last, is_head = 0;
        unsigned long long sum = 0;
        for (uint32_t hops = 0;; hops++) {
            const uint4 a = *reinterpret_cast<const uint4 *>(&frag[cur]);            // next, len, last, chain_head
            if (cur == s) is_head = a.w;
            const unsigned long long fs = frag[cur].sum;
            if (cur != s) { uint2 ob; ob.x = i; ob.y = len; *reinterpret_cast<uint2 *>(&frag[cur].owner) = ob; }
            sum += fs; len += a.y; last = a.z; nxt = a.x;
            // (an inconsistent graph — the same k-mer on two rows — can lead a walker onto a record nobody wrote: an error, never a
            // wild access here or in the kernels that read this segment)
            if (last >= total || (nxt != NIL && nxt >= total) || hops >= total) { flags[0] = 1; last = s; nxt = NIL; break; }
            if (nxt == NIL || node_sampled(nxt, split_mask)) break;
            cur = nxt;
        }
        SegRec r; r.node = s; r.len = len; r.last = last; r.sum = sum; r.pad = 0;
        r.next_spl = (nxt == NIL) ? NIL : frag[nxt].owner;       // a splitter's own fragment got its owner in k_succ_split
        r.head = is_head ? HEAD_LINEAR : 0u;
        segs[i] = r;
        my_cov += len;
    }
    for (int o = 32; o > 0; o >>= 1) my_cov += __shfl_down(my_cov, o);
    if ((threadIdx.x & 63) == 0 && my_cov) atomicAdd(n_covered, my_cov);
}

// A circular unitig that holds no sampled node (short ones: the chance is (63/64)^n) is owned by no walker: its
// nodes are the alive nodes without an owner.  Every such node walks its cycle; the one that IS the cycle's smallest
// k-mer acts: in orientation 0 it becomes a head splitter whose one segment is the whole cycle spelled from itself
// (exactly SPEC S10's cut); in orientation 1 it is the mirror strand and nothing is emitted for it.
static constexpr uint32_t ORPHAN_MAX = 1u << 16;          // (no sampled node among n: (63/64)^n; 65536 never happens)
template <int W>
__global__ __launch_bounds__(256) void k_orphan_cycles(Graph<W> g, const uint8_t *__restrict__ alive,
                                                       const uint2 *__restrict__ winfo, uint2 *__restrict__ ol,
                                                       FragRec *__restrict__ frag,
                                                       uint32_t *__restrict__ spl, SegRec *__restrict__ segs,
                                                       unsigned int *__restrict__ n_spl,
                                                       uint32_t seg_cap, uint32_t *__restrict__ flags,
                                                       const unsigned long long *__restrict__ n_alive,
                                                       const unsigned long long *__restrict__ n_covered) {
    if (*n_alive == *n_covered || flags[0] == 4u) return;  // every alive node has an owner: no such ring (the usual case); or the ranking was called off
    const uint32_t total = g.n * 2;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < total; v += gridDim.x * blockDim.x) {
        if (!alive[v >> 1]) continue;
        { uint32_t o_, p_; if (owner_of(ol, frag, v, o_, p_)) continue; }
        const uint2 w0 = winfo[v];
        uint32_t best = v, n = 1, cur = w0.x, last = v;
        unsigned long long sum = w0.y;
        bool ok = true;
        while (cur != v) {
            if (cur == NIL || n >= ORPHAN_MAX) { ok = false; break; }
            const uint2 w = winfo[cur];
            if (node_key_less<W>(g, cur, best)) best = cur;
            sum += w.y; n++; last = cur; cur = w.x;
        }
        if (!ok) { flags[0] = 1; continue; }
        if (best != v || (v & 1u)) continue;                // not the cut point / the mirror strand
        const uint32_t idx = atomicAdd(n_spl, 1u);
        if (idx >= seg_cap) { flags[0] = 2; continue; }
        spl[idx] = v;
        cur = v;
        for (uint32_t j = 0; j < n; j++) {                     // every node a fragment of its own, owned by the new splitter
            uint2 o; o.x = cur; o.y = 0; ol[cur] = o;
            uint2 ob; ob.x = idx; ob.y = j; *reinterpret_cast<uint2 *>(&frag[cur].owner) = ob;
            cur = winfo[cur].x;
        }
        SegRec r; r.node = v; r.next_spl = NIL; r.len = n; r.last = last; r.sum = sum; r.head = HEAD_ORPHAN; r.pad = 0;
        segs[idx] = r;
    }
}

// ---- splitter-list ranking on the device (pointer jumping over ~2N/32 elements) ------------------
// P: predecessor pointer converging to the chain's head splitter (heads point to themselves); A / K: nodes / counts
// between the start of splitter P and the start of this one.  For rings (see the file header): m = the smallest
// splitter index among this splitter and the predecessors its window covers, d / dK = nodes / counts from the start
// of m to the start of this one (the NEAREST occurrence of m going backwards, so d stays below the ring's length
// however often the windows wrap).
struct RankRec { uint32_t P, A, m, d; unsigned long long K, dK; };
// (the splitter count lives on the device: k_orphan_cycles may have appended to the list after the host read it)
__global__ __launch_bounds__(256) void k_rank_init(const SegRec *__restrict__ segs, const unsigned int *__restrict__ n_spl_p,
                                                   RankRec *__restrict__ R) {
    const uint32_t n_spl = *n_spl_p;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_spl; s += gridDim.x * blockDim.x) {
        const SegRec r = segs[s];
        R[s].m = s; R[s].d = 0; R[s].dK = 0;
        if (r.head) { R[s].P = s; R[s].A = 0; R[s].K = 0; }
        if (r.next_spl != NIL && r.next_spl < n_spl) { R[r.next_spl].P = s; R[r.next_spl].A = r.len; R[r.next_spl].K = r.sum; }
    }
}
// one launch follows HOPS pointers (the reach grows HOPS-fold per launch instead of doubling: a launch
// over ~300 k splitters is all latency, so ceil(log4 n) launches of four dependent reads beat
// ceil(log2 n) launches of two); heads point to themselves with A = K = 0, so overshooting adds nothing
static constexpr int RANK_HOPS = 8;                   // hops per launch (round 4: 4 -> 8, ten launches of 9 us became seven of 10 for the bench isolate)
__global__ __launch_bounds__(256) void k_rank_jump(const unsigned int *__restrict__ n_spl_p,
                                                   const RankRec *__restrict__ Ri, RankRec *__restrict__ Ro, uint32_t round = 0) {
    const uint32_t n_spl = *n_spl_p;
    // (the host launches ceil(log4(room)) rounds without knowing the count; once 4^round covers the list every pointer has
    // reached its head and every window spans its ring: the round only carries the records over to the other buffer)
    {
        unsigned long long reach = 1; for (uint32_t r = 0; r < round; r++) reach *= RANK_HOPS;
        if (reach >= (unsigned long long)n_spl + 1ull) {
            for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_spl; s += gridDim.x * blockDim.x) Ro[s] = Ri[s];
            return;
        }
    }
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_spl; s += gridDim.x * blockDim.x) {
        uint32_t p = s, a = 0, bm = 0xFFFFFFFFu, bd = 0;
        unsigned long long kc = 0, bdk = 0;
#pragma unroll
        for (int h = 0; h < RANK_HOPS; h++) {
            const RankRec r = Ri[p];
            // (a, kc): nodes / counts from the start of p to the start of s; p runs from near to far, so the
            // strict '<' keeps the nearest occurrence of the smallest index
            if (r.m < bm) { bm = r.m; bd = r.d + a; bdk = r.dK + kc; }
            a += r.A; kc += r.K;
            p = min(r.P, n_spl - 1u);
        }
        RankRec o; o.P = p; o.A = a; o.K = kc; o.m = bm; o.d = bd; o.dK = bdk;
        Ro[s] = o;
    }
}
// rot: a circular unitig is spelled from its smallest k-mer (SPEC S10): the node at position p of the ring as ranked
// (from its smallest splitter) is written at (p - rot) mod len
struct HeadRec { uint32_t spl, head_node, tail_node, emit, rot, circ; unsigned long long len, kc; };
struct EmitRec { unsigned long long off; uint32_t rot, len; };      // per chain: output offset (~0: not emitted), rotation
struct FinRec { uint32_t slot, base; };                             // per splitter: its chain's record, nodes before it
struct RingMin { unsigned long long prefix; uint32_t vmin, is_ring; };   // per chain record: the ring's smallest k-mer (prefix, node)
// Every chain's last splitter reports the chain: for a linear chain the tail (no next splitter), for a ring the
// splitter in front of the ring's smallest one.  A unitig exists on both strands; of a linear one the strand to
// emit is the lexicographically smaller spelling (SPEC S10), which the first k characters decide: seq(head)
// against seq(rc(tail)); of a circular one it is the strand that holds the smallest k-mer in orientation 0
// (k_ring_rot, once that k-mer is known).
template <int W>
__global__ __launch_bounds__(256) void k_rank_tails(Graph<W> g, const SegRec *__restrict__ segs, const unsigned int *__restrict__ n_spl_p,
                                                    const RankRec *__restrict__ R,
                                                    HeadRec *__restrict__ heads, uint32_t *__restrict__ slot_of,
                                                    RingMin *__restrict__ ringmin,
                                                    unsigned int *__restrict__ n_heads, unsigned int *__restrict__ n_cyc) {
    __shared__ uint32_t blk_cyc;
    if (threadIdx.x == 0) blk_cyc = 0;
    __syncthreads();
    const uint32_t n_spl = *n_spl_p;
    uint32_t my_cyc = 0;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_spl; s += gridDim.x * blockDim.x) {
        const SegRec r = segs[s];
        const RankRec me = R[s];
        const uint32_t hd = segs[me.P].head;
        if (hd != 0u) {                                            // a chain with a head: linear, or an orphan cycle
            if (r.next_spl != NIL) continue;
            const uint32_t root = me.P;
            const uint32_t slot = atomicAdd(n_heads, 1u);          // one per chain
            HeadRec h; h.spl = root; h.head_node = segs[root].node; h.tail_node = r.last; h.rot = 0;
            h.len = (unsigned long long)me.A + r.len; h.kc = me.K + r.sum;
            if (hd == HEAD_ORPHAN) { h.circ = 1; h.emit = 1; }
            else {
                h.circ = 0;
                const Kmer<W> a = g.seq(h.head_node), b = g.seq(h.tail_node ^ 1u);
                if (km_less<W>(a, b)) h.emit = 1;
                else if (km_less<W>(b, a)) h.emit = 0;
                else h.emit = h.head_node <= (h.tail_node ^ 1u);   // the chain is its own mirror, or a tie on ids
            }
            heads[slot] = h; slot_of[root] = slot;
            RingMin rm; rm.prefix = ~0ull; rm.vmin = NIL; rm.is_ring = 0; ringmin[slot] = rm;
        } else {                                                   // on a ring, ranked from its smallest splitter me.m
            my_cyc++;
            if (r.next_spl != me.m) continue;
            const uint32_t root = me.m;
            const uint32_t slot = atomicAdd(n_heads, 1u);
            HeadRec h; h.spl = root; h.head_node = segs[root].node; h.tail_node = r.last; h.circ = 1;
            h.len = (unsigned long long)me.d + r.len; h.kc = me.dK + r.sum;
            h.emit = 0; h.rot = 0;                                 // (k_ring_rot)
            heads[slot] = h; slot_of[root] = slot;
            RingMin rm; rm.prefix = ~0ull; rm.vmin = NIL; rm.is_ring = 1; ringmin[slot] = rm;
        }
    }
    if (my_cyc) atomicAdd(&blk_cyc, my_cyc);
    __syncthreads();
    if (threadIdx.x == 0 && blk_cyc) atomicAdd(n_cyc, blk_cyc);
}
// per splitter: which chain record, and how many nodes of the chain come before it
__global__ __launch_bounds__(256) void k_rank_fin(const SegRec *__restrict__ segs, const unsigned int *__restrict__ n_spl_p,
                                                  const RankRec *__restrict__ R, const uint32_t *__restrict__ slot_of,
                                                  FinRec *__restrict__ fin) {
    const uint32_t n_spl = *n_spl_p;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_spl; s += gridDim.x * blockDim.x) {
        const RankRec me = R[s];
        const bool ring = segs[me.P].head == 0u;
        FinRec f; f.slot = slot_of[ring ? me.m : me.P]; f.base = ring ? me.d : me.A;
        fin[s] = f;
    }
}

// ol[v]: {fragment head, position in the fragment} -> {chain record (NIL: none), position in the chain}.  Same tiles and
// chunks as k_local_frag (a fragment never leaves its chunk); nodes that k_orphan_cycles re-homed are fragments of their own.
__global__ __launch_bounds__(LF_THREADS) void k_tile_final(uint32_t n_rows, const uint32_t *__restrict__ row_starts, uint32_t tile_rows,
                                                           uint2 *__restrict__ ol, const FragRec *__restrict__ frag,
                                                           const FinRec *__restrict__ fin,
                                                           const unsigned long long *__restrict__ skip = nullptr /* as k_succ_split */,
                                                           const uint32_t *__restrict__ flags = nullptr /* [0] == 4: the ranking was called off */) {
    if (skip && (skip[0] | skip[1])) return;
    if (flags && flags[0] == 4u) return;
    __shared__ uint2 l_chain[LF_TILE];         // per local fragment head: {chain record, nodes of the chain before the fragment}
    __shared__ uint32_t t_lo, t_hi;
    tile_bounds(row_starts, n_rows, tile_rows, &t_lo, &t_hi);
    __syncthreads();
    const uint32_t hi = t_hi;
    for (uint32_t c = t_lo; c < hi; c += LF_TILE / 2u) {
        const uint32_t base = 2u * c;
        const uint32_t size = 2u * (min(c + LF_TILE / 2u, hi) - c);
        uint2 t[LF_ITEMS];
#pragma unroll
        for (int it = 0; it < LF_ITEMS; it++) {
            const uint32_t j = (uint32_t)it * LF_THREADS + threadIdx.x;
            const uint32_t v = base + j;
            t[it].x = NIL; t[it].y = 0;
            if (j < size) {
                t[it] = ol[v];
                if (t[it].x == v) {                                    // a fragment head
                    uint2 e; e.x = NIL; e.y = 0;
                    const uint2 ob = *reinterpret_cast<const uint2 *>(&frag[v].owner);
                    if (ob.x != NIL) { const FinRec f = fin[ob.x]; e.x = f.slot; e.y = f.base + ob.y; }
                    l_chain[j] = e;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < LF_ITEMS; it++) {
            const uint32_t j = (uint32_t)it * LF_THREADS + threadIdx.x;
            if (j < size) {
                uint2 o; o.x = NIL; o.y = 0;
                if (t[it].x != NIL) {
                    const uint2 e = l_chain[t[it].x - base];
                    if (e.x != NIL) { o.x = e.x; o.y = e.y + t[it].y; }
                }
                ol[base + j] = o;
            }
        }
        __syncthreads();
    }
}

// ---- the smallest k-mer of every ring (where SPEC S10 cuts it): two streaming passes over the oriented nodes.
// Both return at once when the graph holds no ring (*n_cyc == 0).  Pass 1: the smallest 64-bit key prefix per ring
// (the lanes of a wave nearly always sit on the same ring: one shuffle reduction and one atomic per wave and ring).
// Pass 2: the nodes whose prefix equals it — one, unless k > 32 and two k-mers share 64 leading bits — settle the
// exact minimum among themselves by compare-and-swap.
template <int W>
__global__ __launch_bounds__(256) void k_ring_min1(Graph<W> g, const uint8_t *__restrict__ alive, const uint2 *__restrict__ ol,
                                                   RingMin *__restrict__ ringmin,
                                                   const unsigned int *__restrict__ n_cyc) {
    if (*n_cyc == 0) return;
    // block-level table ring -> smallest prefix seen by this block: the global atomics are one per block and ring
    constexpr uint32_t TS = 64;
    __shared__ uint32_t t_slot[TS];
    __shared__ unsigned long long t_min[TS];
    if (threadIdx.x < TS) { t_slot[threadIdx.x] = NIL; t_min[threadIdx.x] = ~0ull; }
    __syncthreads();
    auto put = [&](uint32_t sl, unsigned long long m) {
        uint32_t i = mix32(sl) & (TS - 1u);
        for (uint32_t t = 0; t < TS; t++) {
            const uint32_t old = atomicCAS(&t_slot[i], NIL, sl);
            if (old == NIL || old == sl) { atomicMin(&t_min[i], m); return; }
            i = (i + 1u) & (TS - 1u);
        }
        atomicMin(&ringmin[sl].prefix, m);                       // table full: more than 64 rings in one block's nodes
    };
    const int lane = threadIdx.x & 63;
    const uint32_t total = g.n * 2;
    const uint32_t stride = gridDim.x * blockDim.x;
    // this thread's rings and their minima so far: two entries, because a ring and its mirror strand interleave in
    // memory (v and v ^ 1 are neighbours); a third ring evicts an entry into the block's table
    uint32_t cs0 = NIL, cs1 = NIL; unsigned long long cm0 = ~0ull, cm1 = ~0ull;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < total; v += stride) {
        if (!alive[v >> 1]) continue;
        const uint32_t sl = ol[v].x;
        if (sl == NIL || !ringmin[sl].is_ring) continue;
        const unsigned long long pf = km_prefix64<W>(g.keys, v >> 1, g.k);
        if (sl == cs0) { if (pf < cm0) cm0 = pf; }
        else if (sl == cs1) { if (pf < cm1) cm1 = pf; }
        else if (cs0 == NIL) { cs0 = sl; cm0 = pf; }
        else {
            if (cs1 != NIL) put(cs1, cm1);
            cs1 = sl; cm1 = pf;
        }
    }
    // wave-level: one table update per distinct ring among the lanes, for either entry
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const uint32_t cs = e ? cs1 : cs0;
        const unsigned long long cm = e ? cm1 : cm0;
        unsigned long long todo = __ballot(cs != NIL);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t ls = (uint32_t)__shfl((int)cs, leader);
            const bool mine = cs == ls;
            unsigned long long x = mine ? cm : ~0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long y = __shfl_xor(x, o); x = y < x ? y : x; }
            if (lane == leader) put(ls, x);
            todo &= ~__ballot(mine);
        }
    }
    __syncthreads();
    if (threadIdx.x < TS && t_slot[threadIdx.x] != NIL) atomicMin(&ringmin[t_slot[threadIdx.x]].prefix, t_min[threadIdx.x]);
}
template <int W>
__global__ __launch_bounds__(256) void k_ring_min2(Graph<W> g, const uint8_t *__restrict__ alive, const uint2 *__restrict__ ol,
                                                   RingMin *__restrict__ ringmin,
                                                   const unsigned int *__restrict__ n_cyc) {
    if (*n_cyc == 0) return;
    const uint32_t total = g.n * 2;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < total; v += gridDim.x * blockDim.x) {
        if (!alive[v >> 1]) continue;
        const uint32_t sl = ol[v].x;
        if (sl == NIL) continue;
        const RingMin rm = ringmin[sl];
        if (!rm.is_ring || km_prefix64<W>(g.keys, v >> 1, g.k) != rm.prefix) continue;
        uint32_t cur = atomicCAS(&ringmin[sl].vmin, NIL, v);
        while (cur != NIL && cur != v && node_key_less<W>(g, v, cur)) {
            const uint32_t prev = atomicCAS(&ringmin[sl].vmin, cur, v);
            if (prev == cur) break;
            cur = prev;
        }
    }
}
// Per ring: SPEC S10 cuts it before its smallest k-mer x, i.e. it is spelled from the oriented node (x, 0) along that
// node's strand, and the unitig is then emitted as min(spelling, revcomp(spelling)).  The reverse complement of that
// spelling is the MIRROR ring spelled from the successor of (x, 1); which of the two is smaller is decided by their
// first k-mers (two different oriented nodes spell two different k-mers), so the smaller one is emitted directly and
// the host never has to reverse-complement a chromosome.  The ring whose smallest k-mer sits in orientation 0 decides
// for both strands; its mirror's record is only written by it.
template <int W>
__global__ __launch_bounds__(256) void k_ring_rot(Graph<W> g, HeadRec *__restrict__ heads, const unsigned int *__restrict__ n_heads_p,
                                                  const RingMin *__restrict__ ringmin, const uint2 *__restrict__ winfo,
                                                  const uint2 *__restrict__ ol,
                                                  const unsigned int *__restrict__ n_cyc, uint32_t *__restrict__ flags) {
    if (*n_cyc == 0) return;
    const uint32_t n_heads = *n_heads_p;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_heads; i += gridDim.x * blockDim.x) {
        const RingMin rm = ringmin[i];
        if (!rm.is_ring) continue;
        if (rm.vmin == NIL) { flags[0] = 3; continue; }             // (cannot happen: every ring has nodes)
        if (rm.vmin & 1u) continue;                                 // the mirror strand: decided by its partner
        const uint2 o = ol[rm.vmin];                                // {this ring's record, position of its smallest k-mer}
        const uint32_t w = winfo[rm.vmin ^ 1u].x;                   // first node of the reverse-complement spelling
        bool mirror = false;
        uint2 ow; ow.x = NIL; ow.y = 0;
        if (w != NIL) {
            ow = ol[w];
            mirror = ow.x != NIL && ow.x != i && km_less<W>(g.seq(w), g.seq(rm.vmin));
        }
        if (mirror) {
            heads[ow.x].emit = 1; heads[ow.x].rot = ow.y;
            heads[i].emit = 0;
        } else {
            heads[i].emit = 1; heads[i].rot = o.y;
        }
    }
}

// per node: chain record -> output offset (~0 = chain not emitted).  One thread takes both orientations of a node:
// one key load and one 16-byte read of ol for the two; the base an oriented node appends is the last base of its
// k-mer (orientation 1: the complement of the first), so a reverse complement is only built for the first node of a
// chain (which spells its whole k-mer).
// The emission plan of an assembly with few chains, on the device (a fragmented one is planned by writer_gpu.h, any other by
// the host): which chain records are emitted and where their text starts — so that k_emit and the download of the text can be
// launched BEHIND the ranking, before the host has seen a single chain record.  plan[0] = 1: done, plan[1] = bytes of text,
// plan[2] = chains emitted; plan[0] = 2: not planned (more chains than max_heads, more text than out_cap, or none ranked:
// the host does it).  One workgroup.  Offsets in chain-record order, as the host's loop assigns them.
__global__ __launch_bounds__(1024) void k_plan_emit(const HeadRec *__restrict__ heads, const unsigned int *__restrict__ n_heads_p, uint32_t k,
                                                   uint32_t max_heads, unsigned long long out_cap, const uint32_t *__restrict__ flags,
                                                   EmitRec *__restrict__ off, unsigned long long *__restrict__ plan) {
    __shared__ unsigned long long wsum[16];
    __shared__ uint32_t wcnt[16];
    const uint32_t nh = *n_heads_p;
    if (nh == 0 || nh > max_heads || flags[0] != 0u) { if (threadIdx.x == 0) plan[0] = 2ull; return; }
    const uint32_t t = threadIdx.x, lane = t & 63u, wid = t >> 6;
    // (max_heads <= 1024: one chain record per thread)
    unsigned long long sz = 0; uint32_t em = 0; HeadRec h{};
    if (t < nh) { h = heads[t]; if (h.emit) { sz = h.len + (unsigned long long)(k - 1u); em = 1u; } }
    unsigned long long incl = sz; uint32_t cincl = em;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long u = __shfl_up(incl, o); const uint32_t cu = (uint32_t)__shfl_up((int)cincl, o);
        if ((int)lane >= o) { incl += u; cincl += cu; }
    }
    if (lane == 63u) { wsum[wid] = incl; wcnt[wid] = cincl; }
    __syncthreads();
    unsigned long long base = 0, total = 0; uint32_t ctotal = 0;
    for (uint32_t w = 0; w < 16u; w++) { if (w < wid) base += wsum[w]; total += wsum[w]; ctotal += wcnt[w]; }
    if (total > out_cap) { if (t == 0) plan[0] = 2ull; return; }
    if (t < nh) { EmitRec e; e.off = em ? base + incl - sz : ~0ull; e.rot = h.rot; e.len = (uint32_t)h.len; off[t] = e; }
    if (t == 0) { plan[1] = total; plan[2] = ctotal; __threadfence(); plan[0] = 1ull; }
}

template <int W>
__global__ __launch_bounds__(256) void k_emit(Graph<W> g, const uint8_t *__restrict__ alive,
                                              const uint2 *__restrict__ ol,
                                              const EmitRec *__restrict__ head_off,
                                              char *__restrict__ out,
                                              const unsigned long long *__restrict__ plan = nullptr /* k_plan_emit's verdict: [0] == 1 or nothing is written */) {
    if (plan && plan[0] != 1ull) return;
    const uint32_t ACGT = 0x54474341u;                     // 'A','C','G','T' little-endian
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += gridDim.x * blockDim.x) {
        if (!alive[i]) continue;
        const uint4 o2 = *reinterpret_cast<const uint4 *>(&ol[2u * i]);      // {chain record, position in the chain} x 2 (k_tile_final)
        if (o2.x == NIL && o2.z == NIL) continue;
        const Kmer<W> x = g.keys.load(i);
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const uint32_t slot = o ? o2.z : o2.x;
            if (slot == NIL) continue;
            const EmitRec er = head_off[slot];
            if (er.off == ~0ull) continue;
            uint32_t pos = o ? o2.w : o2.y;
            if (er.rot) pos = pos >= er.rot ? pos - er.rot : pos + er.len - er.rot;    // a circular unitig starts at its smallest k-mer
            char *dst = out + er.off;
            const uint32_t b = o ? 3u - km_bits2<W>(x, 2 * (g.k - 1)) : km_last_base<W>(x);
            dst[g.k - 1 + pos] = (char)((ACGT >> (8 * b)) & 0xFF);
            if (pos == 0) {
                const Kmer<W> y = o ? km_revcomp<W>(x, g.k) : x;
                for (int j = 0; j + 1 < g.k; j++) dst[j] = (char)((ACGT >> (8 * km_bits2<W>(y, 2 * (g.k - 1 - j)))) & 0xFF);
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

