// fastq_gpu.hip — FASTQ text -> quality-masked, segmented, 2-bit packed reads, on the device.
//
// the reference ingests FASTQ inside preprocess() (web_sys::File -> gz sniff
// -> seq_io reader: /root/reference/AGENTS.md:180-183, sibling rust/orphos-bridge/src/fastx_wasm.rs:53-70;
// caller www/src/workers/Assembler.ts:100).  The host packer (fastq.cpp) does this at ~0.3 Gbases/s
// on one core, 300x slower than the kernels behind it.  Here the text is uploaded as it is and
// parsed by byte-streaming kernels (all HBM-bound):
//   k_nl_count / k_nl_fill   newline positions (line index)
//   k_read_wave<0>           one wave per record: framing checks (SPEC S1), valid-base runs (SPEC S2) -> counts
//   k_read_wave<1>           the same walk again: segment table + the 2-bit stream (gather + encode per output word)
// Only REGULAR input is handled (fixed 4-line framing, no blank lines between records); anything
// else — and every malformed record — is left to the host parser, which owns the error messages.
// The result is the same packed layout bit for bit (tests/test_gpu_parity.py compares both).
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdint.h>
#include <mutex>
#include <string>
#include <vector>

#include "fastq_gpu.h"
#include "pipeline.h"

namespace shk {

#define FQCHK(call)                                                                       \
    do {                                                                                  \
        hipError_t _e = (call);                                                           \
        if (_e != hipSuccess) {                                                           \
            err = std::string(#call) + ": " + hipGetErrorString(_e);                      \
            return -5;                                                                    \
        }                                                                                 \
    } while (0)

static constexpr uint32_t NL_CHUNK = 16384;      // text bytes per workgroup of the newline kernels

// ---- exclusive scan of uint64 (three phases, 4096 items per workgroup) --------------------------
static constexpr int SCAN_T = 1024, SCAN_I = 4;
__global__ __launch_bounds__(SCAN_T) void k_scan_local(unsigned long long *__restrict__ a, uint64_t n,
                                                       unsigned long long *__restrict__ block_sum) {
    __shared__ unsigned long long wsum[SCAN_T / 64];
    const uint64_t base = (uint64_t)blockIdx.x * (SCAN_T * SCAN_I) + (uint64_t)threadIdx.x * SCAN_I;
    unsigned long long v[SCAN_I], mine = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; i++) { v[i] = base + i < n ? a[base + i] : 0ull; mine += v[i]; }
    unsigned long long incl = mine;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    unsigned long long off = 0;
    for (int w = 0; w < wid; w++) off += wsum[w];
    unsigned long long run = off + incl - mine;
#pragma unroll
    for (int i = 0; i < SCAN_I; i++) { if (base + i < n) a[base + i] = run; run += v[i]; }
    if (threadIdx.x == SCAN_T - 1) block_sum[blockIdx.x] = off + incl;
}
__global__ __launch_bounds__(SCAN_T) void k_scan_add(unsigned long long *__restrict__ a, uint64_t n,
                                                     const unsigned long long *__restrict__ block_off) {
    const uint64_t base = (uint64_t)blockIdx.x * (SCAN_T * SCAN_I) + (uint64_t)threadIdx.x * SCAN_I;
    const unsigned long long o = block_off[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_I; i++) if (base + i < n) a[base + i] += o;
}

static inline unsigned grid1(uint64_t work, unsigned block = 256, unsigned cap = 65536) {
    uint64_t b = (work + block - 1) / block; if (b < 1) b = 1; if (b > cap) b = cap; return (unsigned)b;
}

struct Scratch {                                  // blocks of the process-wide device pool, returned on scope exit
    std::vector<std::pair<void *, size_t>> ptrs;
    hipStream_t st = nullptr; bool bound = false;     // the stream the blocks are used on
    void bind(hipStream_t s) { st = s; bound = true; }
    // the pool has no stream-ordering bookkeeping: a block may be handed to another stream's user (the uploader
    // thread) at once, so kernels still in flight on an early error return are waited for before the release
    ~Scratch() {
        if (bound && !ptrs.empty()) (void)hipStreamSynchronize(st);
        for (auto &p : ptrs) if (p.first) device_pool_release(p.first, p.second);
    }
    template <typename T> T *get(size_t count, std::string &err) {
        size_t bytes = (count ? count : 1) * sizeof(T);
        void *p = device_pool_alloc(bytes);
        if (!p) { err = "device allocation failed (FASTQ parser)"; return nullptr; }
        ptrs.emplace_back(p, bytes);
        return (T *)p;
    }
    size_t keep(void *p) { for (auto &q : ptrs) if (q.first == p) { q.first = nullptr; return q.second; } return 0; }
};

// in-place exclusive scan of a[0..n); *total (device) receives the sum
static int exclusive_scan(unsigned long long *a, uint64_t n, unsigned long long *d_total, hipStream_t st, Scratch &sc,
                          std::string &err) {
    const uint64_t per = SCAN_T * SCAN_I;
    const uint64_t nb = (n + per - 1) / per;
    if (n == 0) { FQCHK(hipMemsetAsync(d_total, 0, 8, st)); return 0; }
    unsigned long long *bs = sc.get<unsigned long long>(nb, err);
    if (!bs) return -4;
    hipLaunchKernelGGL(k_scan_local, dim3((unsigned)nb), dim3(SCAN_T), 0, st, a, n, bs);
    if (nb == 1) { FQCHK(hipMemcpyAsync(d_total, bs, 8, hipMemcpyDeviceToDevice, st)); return 0; }
    if (int rc = exclusive_scan(bs, nb, d_total, st, sc, err)) return rc;
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(SCAN_T), 0, st, a, n, bs);
    FQCHK(hipGetLastError());
    return 0;
}

// 0x80 in every byte of w that equals '\n' (exact: no carry between bytes)
__device__ __forceinline__ uint32_t nl_mask(uint32_t w) {
    const uint32_t x = w ^ 0x0A0A0A0Au;
    const uint32_t t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
    return ~t & 0x80808080u;
}

// ---- newline index ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nl_count(const uint8_t *__restrict__ text, uint64_t n,
                                                  unsigned long long *__restrict__ cnt) {
    const uint64_t b0 = (uint64_t)blockIdx.x * NL_CHUNK;
    uint32_t c = 0;
    // 16 bytes per lane per step
    for (uint32_t i = threadIdx.x * 16u; i < NL_CHUNK; i += 256u * 16u) {
        const uint64_t p = b0 + i;
        if (p + 16 <= n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(text + p);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            c += __popc(nl_mask(w[0])) + __popc(nl_mask(w[1])) + __popc(nl_mask(w[2])) + __popc(nl_mask(w[3]));
        } else {
            for (uint64_t q = p; q < n && q < p + 16; q++) c += text[q] == '\n';
        }
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    __shared__ uint32_t ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// line_end[j] = position of the j-th '\n' (in text order).  One workgroup per 16 KB chunk, 16 bytes per
// lane and step: newline masks by word arithmetic, an exclusive scan of the per-lane counts keeps the order.
__global__ __launch_bounds__(256) void k_nl_fill(const uint8_t *__restrict__ text, uint64_t n,
                                                 const unsigned long long *__restrict__ chunk_off,
                                                 unsigned long long *__restrict__ line_end) {
    __shared__ uint32_t wtot[4];
    const uint64_t b0 = (uint64_t)blockIdx.x * NL_CHUNK;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned long long base = chunk_off[blockIdx.x];
    for (uint32_t sub = 0; sub < NL_CHUNK; sub += 256u * 16u) {
        const uint64_t p = b0 + sub + (uint64_t)threadIdx.x * 16u;
        uint32_t m[4] = {0, 0, 0, 0};
        if (p + 16 <= n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(text + p);
            m[0] = nl_mask(v.x); m[1] = nl_mask(v.y); m[2] = nl_mask(v.z); m[3] = nl_mask(v.w);
        } else if (p < n) {
            for (uint32_t q = 0; q < 16 && p + q < n; q++) if (text[p + q] == '\n') m[q >> 2] |= 0x80u << (8 * (q & 3));
        }
        const uint32_t c = __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
        uint32_t incl = c;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += u; }
        if (lane == 63) wtot[wid] = incl;
        __syncthreads();
        uint32_t off = 0, tot = 0;
        for (int w = 0; w < 4; w++) { if (w < wid) off += wtot[w]; tot += wtot[w]; }
        unsigned long long o = base + off + incl - c;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t mm = m[j];
            while (mm) {
                const uint32_t bit = (uint32_t)__ffs((int)mm) - 1u;
                line_end[o++] = p + 4u * j + (bit >> 3);
                mm &= mm - 1u;
            }
        }
        base += tot;
        __syncthreads();
    }
}

// ---- per-record scan -------------------------------------------------------------------------------
struct FqParams {
    uint64_t n;            // text bytes (both parts)
    uint64_t part2_off;    // where the second file starts in the text (== n when there is one file)
    uint64_t part1_len;    // bytes of the first file that were kept
    uint64_t part1_reads;  // records of the first file
    uint64_t n_lines;      // '\n' count (+1 if the text does not end with one)
    uint64_t n_nl;         // '\n' count
    uint32_t k, min_qual;
};

__device__ __forceinline__ uint32_t base_code(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 0; case 'C': case 'c': return 1;
        case 'G': case 'g': return 2; case 'T': case 't': return 3;
        default: return 4;
    }
}
// [start, end) of line j with a trailing '\r' removed
__device__ __forceinline__ void line_span(const uint8_t *text, const unsigned long long *line_end, const FqParams &fp,
                                          uint64_t j, uint64_t &s, uint64_t &e) {
    s = j == 0 ? 0 : line_end[j - 1] + 1;
    e = j < fp.n_nl ? line_end[j] : fp.n;
    if (e > s && text[e - 1] == '\r') e--;
}

// 2-bit code of A/C/G/T (either case) and the validity test of SPEC S2 without a table
__device__ __forceinline__ uint32_t acgt_code(uint32_t c) { return ((c >> 1) ^ (c >> 2)) & 3u; }
__device__ __forceinline__ bool acgt_valid(uint32_t c) {
    const uint32_t u = (c | 0x20u) - 'a';                  // a=0 c=2 g=6 t=19
    return u < 26u && ((0x80045u >> u) & 1u);
}

// One WAVE per record (lane = byte of the 64-byte piece being looked at: coalesced line loads, the
// valid-base mask of a piece is one ballot, runs are found with bit scans on it).
// mode 0: framing checks + counts (segments, kept bases) per record
// mode 1: segment table + the packed stream itself: lane w gathers the 16 bases of output word w of the
//         run; words shared with a neighbouring run are merged with atomicOr (the stream starts zeroed)
template <int MODE>
__global__ __launch_bounds__(256) void k_read_wave(const uint8_t *__restrict__ text,
                                                   const unsigned long long *__restrict__ line_end, FqParams fp,
                                                   uint64_t n_reads, unsigned long long *__restrict__ seg_cnt,
                                                   unsigned long long *__restrict__ base_cnt,
                                                   unsigned long long *__restrict__ stats /* [0]=bad flag [6..70)=input bases, partial sums */,
                                                   uint32_t *__restrict__ seg_off, uint32_t *__restrict__ out,
                                                   uint2 *__restrict__ edge /* mode 1: per segment {head, tail} partial words */) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long my_in = 0;
    for (uint64_t r = wave0; r < n_reads; r += n_waves) {
        // line ends 4r-1 .. 4r+3 -> lanes 0..4
        unsigned long long le = 0;
        if (lane < 5) {
            const uint64_t j = 4 * r + (uint64_t)lane;           // line index + 1
            le = j == 0 ? ~0ull : (j - 1 < fp.n_nl ? line_end[j - 1] : fp.n);
        }
        uint64_t ls[4], e_[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            ls[i] = __shfl(le, i) + 1ull;                        // (~0 + 1 = 0 for the first line)
            e_[i] = __shfl(le, i + 1);
        }
        // Everything that depends only on the line ends is requested in ONE round of loads: the last byte of
        // every line (a trailing '\r'), the first bytes of the header and '+' lines, the counts of pass 0
        // (mode 1) and — for records of up to 64*NP bases, i.e. every short read — all sequence and quality
        // bytes (a '\r' that gets cut off afterwards is no valid base anyway).  The record then costs two
        // dependent memory round trips instead of four.
        constexpr int NP = 4;
        const uint64_t s1 = ls[1], s3 = ls[3];
        const uint64_t Lraw = e_[1] - s1;
        const bool small = Lraw <= 64u * NP;
        uint32_t last_b = 0, head_b = 0;
        if (lane < 4) { const uint64_t e = e_[lane], s0 = ls[lane]; if (e > s0) last_b = text[e - 1]; }
        if (lane == 4 && e_[0] > ls[0]) head_b = text[ls[0]];
        if (lane == 5 && e_[2] > ls[2]) head_b = text[ls[2]];
        uint64_t so = 0, bo = 0;
        if (MODE == 1) { so = seg_cnt[r]; bo = base_cnt[r]; }
        uint32_t sb[NP], qb[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const uint64_t i = 64u * p + (uint64_t)lane;
            sb[p] = 'N'; qb[p] = 0;
            if (small && i < Lraw) { sb[p] = text[s1 + i]; if (s3 + i < fp.n) qb[p] = text[s3 + i]; }
        }
        // trailing '\r' of each line
        const uint32_t cr = (lane < 4 && last_b == '\r') ? 1u : 0u;
#pragma unroll
        for (int i = 0; i < 4; i++) e_[i] -= (uint64_t)__shfl((int)cr, i);
        const uint64_t L = e_[1] - s1;
        if (MODE == 0) {
            const bool ok = e_[0] > ls[0] && __shfl((int)head_b, 4) == '@' && e_[2] > ls[2] && __shfl((int)head_b, 5) == '+' &&
                            L == e_[3] - s3;
            if (!ok) { if (lane == 0) { stats[0] = 1; seg_cnt[r] = 0; base_cnt[r] = 0; } continue; }
            if (lane == 0) my_in += L;
        }
        uint64_t run = 0, nseg = 0, nb = 0;                      // wave-uniform
        // mode 1 packs the 2-bit stream of a short record from the codes it already holds (segmented OR over
        // the lanes of one output word) instead of gathering 16 bytes per word from memory again
        unsigned long long m[NP];
        uint32_t code[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const uint64_t i = 64u * p + (uint64_t)lane;
            const bool valid = small && i < L && acgt_valid(sb[p]) && (int)qb[p] - 33 >= (int)fp.min_qual;
            m[p] = __ballot(valid);
            code[p] = acgt_code(sb[p]);
        }
        auto close_run = [&](uint64_t end_pos /* text position one past the run */) {
            if (MODE == 0 && run > 16384 && lane == 0) stats[0] = 1;       // very long runs are split by the host parser
            if (run >= fp.k) {
                if (MODE == 1) {
                    const uint64_t src = end_pos - run, b0 = bo + nb;
                    if (lane == 0) seg_off[so + nseg] = (uint32_t)b0;
                    if (small) {
                        const uint32_t ra = (uint32_t)(src - s1), rb = (uint32_t)(end_pos - s1);   // the run inside the record
                        // a word is written by the lane that holds its last base of the run; the part of a word that a
                        // piece boundary cuts off travels to the next piece (carry), so only the first and the last
                        // word of a run — shared with the neighbouring runs — need an atomic
                        uint32_t carry_wid = 0xFFFFFFFFu, carry_v = 0;                        // (wave-uniform)
                        // The first and the last word of a run may be shared with the neighbouring runs (other waves).
                        // They are not written here: their bits go to edge[segment] and k_merge_edges puts the shared
                        // words together afterwards — atomicOr on them cost 2 ms per GB of text.
                        const uint32_t hw = (uint32_t)(b0 >> 4), tw = (uint32_t)((b0 + run - 1) >> 4);
                        const bool head_part = (b0 & 15u) != 0u || (hw == tw && ((b0 + run) & 15u) != 0u);
                        const bool tail_part = ((b0 + run) & 15u) != 0u && !(hw == tw && head_part);
                        if (lane == 0) {
                            if (!head_part) edge[so + nseg].x = 0u;
                            if (!tail_part) edge[so + nseg].y = 0u;
                        }
#pragma unroll
                        for (int p = 0; p < NP; p++) {
                            if (ra < 64u * (p + 1) && rb > 64u * p) {                        // (wave-uniform)
                                const uint32_t i = 64u * p + (uint32_t)lane;
                                const bool act = i >= ra && i < rb;
                                const uint64_t q = b0 + (uint64_t)(i - ra);                     // stream position of this lane's base
                                const uint32_t wid = act ? (uint32_t)(q >> 4) : 0xFFFFFFFEu;
                                uint32_t v = act ? code[p] << (2u * ((uint32_t)q & 15u)) : 0u;
                                if (lane == 0 && wid == carry_wid) v |= carry_v;
#pragma unroll
                                for (int d = 1; d < 16; d <<= 1) {                              // inclusive OR over the (<= 16, adjacent) lanes of a word
                                    const uint32_t ov = (uint32_t)__shfl_up((int)v, d), ow = (uint32_t)__shfl_up((int)wid, d);
                                    if (lane >= d && ow == wid) v |= ov;
                                }
                                const bool last = act && ((((uint32_t)q & 15u) == 15u) || i + 1u == rb);   // the word (inside this run) ends here
                                if (last) {
                                    const uint64_t w0 = (uint64_t)wid << 4;
                                    const bool whole = w0 >= b0 && w0 + 16 <= b0 + run;          // all 16 bases belong to this run
                                    if (whole) out[wid] = v;
                                    else if (wid == hw && head_part) edge[so + nseg].x = v;
                                    else edge[so + nseg].y = v;
                                }
                                const bool c63 = act && !last;                                  // meaningful in lane 63 only
                                carry_wid = __builtin_amdgcn_readlane((int)(c63 ? wid : 0xFFFFFFFFu), 63);
                                carry_v = __builtin_amdgcn_readlane((int)v, 63);
                            }
                        }
                    } else {
                        if (lane == 0) edge[so + nseg] = make_uint2(0u, 0u);             // (long records keep the atomics below)
                        // output words [b0/16, (b0+run-1)/16]
                        const uint64_t w_first = b0 >> 4, w_last = (b0 + run - 1) >> 4;
                        for (uint64_t w = w_first + (uint64_t)lane; w <= w_last; w += 64) {
                            const uint64_t lo = w << 4;                               // first stream base of the word
                            const uint64_t from = lo < b0 ? b0 : lo, to = (lo + 16 < b0 + run) ? lo + 16 : b0 + run;
                            uint32_t word = 0;
                            for (uint64_t b = from; b < to; b++) word |= acgt_code(text[src + (b - b0)]) << (2 * (uint32_t)(b - lo));
                            if (from == lo && to == lo + 16) out[w] = word; else atomicOr(&out[w], word);
                        }
                    }
                }
                nseg++; nb += run;
            }
            run = 0;
        };
        auto scan_piece = [&](unsigned long long mm, uint64_t p0) {     // runs inside one 64-base piece (uniform bit scans)
            const uint32_t len = (uint32_t)((L - p0) < 64 ? (L - p0) : 64);
            uint32_t pos = 0;
            while (pos < len) {
                const unsigned long long rest = mm >> pos;
                if (rest & 1ull) {                               // a stretch of valid bases
                    uint32_t ones = (~rest) ? (uint32_t)__ffsll((long long)~rest) - 1u : 64u;
                    if (ones > len - pos) ones = len - pos;
                    run += ones; pos += ones;
                    if (pos < len) close_run(s1 + p0 + pos);     // it ended inside this piece
                } else {
                    close_run(s1 + p0 + pos);                    // (a run that ended exactly at the piece boundary)
                    uint32_t zeros = rest ? (uint32_t)__ffsll((long long)rest) - 1u : 64u;
                    if (zeros > len - pos) zeros = len - pos;
                    pos += zeros;
                }
            }
        };
        if (small) {
#pragma unroll
            for (int p = 0; p < NP; p++) if (64u * p < L) scan_piece(m[p], 64u * p);
        } else {
            for (uint64_t p0 = 0; p0 < L; p0 += 64) {
                const uint64_t i = p0 + (uint64_t)lane;
                bool valid = false;
                if (i < L) valid = acgt_valid(text[s1 + i]) && (int)text[s3 + i] - 33 >= (int)fp.min_qual;
                scan_piece(__ballot(valid), p0);
            }
        }
        close_run(s1 + L);
        if (MODE == 0 && lane == 0) { seg_cnt[r] = nseg; base_cnt[r] = nb; }
    }
    // (one counter would serialise 262 144 same-address atomics: 3.2 ms, the whole cost of this pass; 64 counters)
    if (MODE == 0 && lane == 0 && my_in) atomicAdd(&stats[6 + (blockIdx.x & 63u)], my_in);
}

__global__ void k_set_u32(uint32_t *p, uint32_t v) { *p = v; }

// The words of the packed stream that two or more segments share: the segment that holds the word's first
// base puts them together from its own tail bits and the head bits of the segments that start inside the word
// (a segment of >= 15 bases: at most three segments meet in a word).  seg_off[n_seg] = n_bases is set.
__global__ __launch_bounds__(256) void k_merge_edges(const uint32_t *__restrict__ seg_off, const uint2 *__restrict__ edge,
                                                     uint64_t n_seg, uint32_t *__restrict__ out) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_seg; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = seg_off[j], b = seg_off[j + 1];
        if ((b & 15u) == 0u) continue;                             // the last word is whole
        const uint32_t w = (b - 1u) >> 4;
        const uint2 ej = edge[j];
        uint32_t v;
        if (a > (w << 4)) {
            // the segment lies inside one word: it owns it only if that word starts the stream's shared stretch,
            // i.e. if no earlier segment reaches into it — impossible unless a is the word's first base
            continue;
        }
        v = (a >> 4) == w ? ej.x : ej.y;                           // (a single-word segment keeps all its bits in .x)
        for (uint64_t m = j + 1; m < n_seg && (seg_off[m] >> 4) == w; m++) v |= edge[m].x;
        out[w] |= v;                                               // (words of long records were started with atomics)
    }
}

// byte offset (inside its own file) after record `every*(j+1)` for the progress strings; bit 63 = second file
__global__ __launch_bounds__(256) void k_progress_marks(const unsigned long long *__restrict__ line_end, FqParams fp,
                                                        uint64_t every, uint64_t n_marks, unsigned long long *__restrict__ marks,
                                                        uint64_t first_mark, uint64_t read_base) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_marks; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = every * (first_mark + j + 1) - 1 - read_base;   // last record of the batch (read_base records came in earlier pieces)
        const uint64_t line = 4 * r + 3;
        uint64_t p = line < fp.n_nl ? line_end[line] + 1 : fp.n;
        if (r < fp.part1_reads) marks[j] = p < fp.part1_len ? p : fp.part1_len;
        else { p -= fp.part2_off; const uint64_t l2 = fp.n - fp.part2_off; marks[j] = (p < l2 ? p : l2) | (1ull << 63); }
    }
}

// number of '\n' in text[0..n)
__global__ __launch_bounds__(256) void k_nl_total(const uint8_t *__restrict__ text, uint64_t n, unsigned long long *__restrict__ out) {
    unsigned long long c = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) c += text[p] == '\n';
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}


// bytes of a FASTQ text without its trailing blank lines (the host parser skips them)
static size_t trimmed_len(const uint8_t *t, size_t n) {
    size_t e = n;
    for (;;) {
        if (e >= 2 && t[e - 1] == '\n' && t[e - 2] == '\n') { e -= 1; continue; }
        if (e >= 3 && t[e - 1] == '\n' && t[e - 2] == '\r' && t[e - 3] == '\n') { e -= 2; continue; }
        break;
    }
    if (e == 1 && t[0] == '\n') e = 0;
    if (e == 2 && t[0] == '\r' && t[1] == '\n') e = 0;
    return e;
}

int gpu_pack_fastq(const uint8_t *t1, size_t n1, const uint8_t *t2, size_t n2, uint32_t k, uint32_t min_qual,
                   uint64_t every, void *stream_v, GpuPacked &out, std::string &err, uint64_t read_base,
                   const GpuText *uploaded) {
    hipStream_t st = (hipStream_t)stream_v;
    out = GpuPacked();
    Scratch sc; sc.bind(st);
    if (uploaded && t2) { err = "an uploaded text cannot be combined with a second file"; return -1; }
    // ---- framing that can be decided on the host: trailing blank lines are ignored, the last line of a
    // file may lack its newline (one is supplied between the files)
    const size_t e1 = uploaded ? uploaded->e : trimmed_len(t1, n1), e2 = t2 ? trimmed_len(t2, n2) : 0;
    const bool unterm1 = uploaded ? uploaded->unterminated : (e1 && t1[e1 - 1] != '\n'), unterm2 = e2 && t2[e2 - 1] != '\n';
    const size_t off2 = e1 + ((unterm1 && e2) ? 1 : 0);
    const size_t e = off2 + e2;
    if (e == 0) {                                        // no records at all
        uint32_t *so = sc.get<uint32_t>(2, err), *bs = sc.get<uint32_t>(2, err);
        if (!so || !bs) return -4;
        FQCHK(hipMemsetAsync(so, 0, 8, st)); FQCHK(hipMemsetAsync(bs, 0, 8, st));
        FQCHK(hipStreamSynchronize(st));
        out.bases_bytes = sc.keep(bs); out.seg_off_bytes = sc.keep(so);
        out.d_bases = bs; out.d_seg_off = so; return 0;
    }
    const bool unterminated = e2 ? unterm2 : unterm1;
    uint8_t *text = uploaded ? uploaded->d : sc.get<uint8_t>(e + 32, err);
    if (!text) return -4;
    hipEvent_t ev0, ev1, ev2;
    FQCHK(hipEventCreate(&ev0)); FQCHK(hipEventCreate(&ev1)); FQCHK(hipEventCreate(&ev2));
    struct EvGuard { hipEvent_t a, b, c; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(c); } } evg{ev0, ev1, ev2};
    FQCHK(hipEventRecord(ev0, st));
    if (!uploaded) {
        if (e1) FQCHK(hipMemcpyAsync(text, t1, e1, hipMemcpyHostToDevice, st));
        if (off2 > e1) FQCHK(hipMemsetAsync(text + e1, '\n', 1, st));
        if (e2) FQCHK(hipMemcpyAsync(text + off2, t2, e2, hipMemcpyHostToDevice, st));
        FQCHK(hipMemsetAsync(text + e, 0, 32, st));
    }
    FQCHK(hipEventRecord(ev1, st));

    const uint64_t n_chunks = (e + NL_CHUNK - 1) / NL_CHUNK;
    unsigned long long *chunk_cnt = sc.get<unsigned long long>(n_chunks, err);
    unsigned long long *d_tot = sc.get<unsigned long long>(8 + 64, err);     // [8..72): partial sums of the input bases
    if (!chunk_cnt || !d_tot) return -4;
    FQCHK(hipMemsetAsync(d_tot, 0, (8 + 64) * 8, st));
    hipLaunchKernelGGL(k_nl_count, dim3((unsigned)n_chunks), dim3(256), 0, st, text, (uint64_t)e, chunk_cnt);
    if (int rc = exclusive_scan(chunk_cnt, n_chunks, d_tot, st, sc, err)) return rc;
    unsigned long long n_nl = 0;
    FQCHK(hipMemcpyAsync(&n_nl, d_tot, 8, hipMemcpyDeviceToHost, st));
    FQCHK(hipStreamSynchronize(st));
    FqParams fp; fp.n = e; fp.n_nl = n_nl; fp.n_lines = n_nl + (unterminated ? 1 : 0); fp.k = k; fp.min_qual = min_qual;
    fp.part2_off = off2; fp.part1_len = e1; fp.part1_reads = 0;
    if (fp.n_lines % 4 != 0) return 1;                    // irregular framing: the host parser decides
    const uint64_t n_reads = fp.n_lines / 4;
    if (e2) {                                             // the first file must hold whole records too
        hipLaunchKernelGGL(k_nl_total, dim3(grid1(off2, 256, 4096)), dim3(256), 0, st, text, (uint64_t)off2, d_tot + 6);
        unsigned long long l1 = 0;
        FQCHK(hipMemcpyAsync(&l1, d_tot + 6, 8, hipMemcpyDeviceToHost, st));
        FQCHK(hipStreamSynchronize(st));
        if (l1 % 4 != 0) return 1;
        fp.part1_reads = l1 / 4;
    } else fp.part1_reads = n_reads;
    unsigned long long *line_end = sc.get<unsigned long long>(n_nl + 1, err);
    if (!line_end) return -4;
    hipLaunchKernelGGL(k_nl_fill, dim3((unsigned)n_chunks), dim3(256), 0, st, text, (uint64_t)e, chunk_cnt, line_end);

    unsigned long long *seg_cnt = sc.get<unsigned long long>(n_reads, err);
    unsigned long long *base_cnt = sc.get<unsigned long long>(n_reads, err);
    if (!seg_cnt || !base_cnt) return -4;
    hipLaunchKernelGGL(k_read_wave<0>, dim3(grid1(n_reads * 64)), dim3(256), 0, st, text, line_end, fp, n_reads, seg_cnt, base_cnt,
                       d_tot + 2, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint2 *)nullptr);
    if (int rc = exclusive_scan(seg_cnt, n_reads, d_tot + 4, st, sc, err)) return rc;
    if (int rc = exclusive_scan(base_cnt, n_reads, d_tot + 5, st, sc, err)) return rc;
    unsigned long long h[8 + 64];
    FQCHK(hipMemcpyAsync(h, d_tot, sizeof h, hipMemcpyDeviceToHost, st));
    FQCHK(hipStreamSynchronize(st));
    h[3] = 0;
    for (int i = 0; i < 64; i++) h[3] += h[8 + i];
    if (h[2]) return 1;                                   // a malformed record: the host parser reports it
    const uint64_t n_seg = h[4], n_bases = h[5];
    if (n_bases >= 0xFFFFFFF0ull) { err = "input exceeds 2^32 bases per batch"; return -1; }
    const uint64_t n_words = (n_bases >> 4) + 2;          // partial word + one spare word (as PackedReads::finish)
    uint32_t *seg_off = sc.get<uint32_t>(n_seg + 1, err);
    uint32_t *bases = sc.get<uint32_t>(n_words, err);
    uint2 *edge = sc.get<uint2>(n_seg + 1, err);
    if (!seg_off || !bases || !edge) return -4;
    FQCHK(hipMemsetAsync(bases, 0, n_words * 4, st));
    if (n_seg)
        hipLaunchKernelGGL(k_read_wave<1>, dim3(grid1(n_reads * 64)), dim3(256), 0, st, text, line_end, fp, n_reads, seg_cnt, base_cnt,
                           d_tot + 2, seg_off, bases, edge);
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, seg_off + n_seg, (uint32_t)n_bases);
    if (n_seg) hipLaunchKernelGGL(k_merge_edges, dim3(grid1(n_seg)), dim3(256), 0, st, seg_off, edge, n_seg, bases);
    FQCHK(hipGetLastError());
    // progress marks
    if (every) {
        const uint64_t first_mark = read_base / every;
        const uint64_t n_marks = (read_base + n_reads) / every - first_mark;
        out.first_mark = first_mark;
        if (n_marks) {
            unsigned long long *marks = sc.get<unsigned long long>(n_marks, err);
            if (!marks) return -4;
            hipLaunchKernelGGL(k_progress_marks, dim3(grid1(n_marks)), dim3(256), 0, st, line_end, fp, every, n_marks, marks, first_mark, read_base);
            out.progress_bytes.resize(n_marks);
            FQCHK(hipMemcpyAsync(out.progress_bytes.data(), marks, n_marks * 8, hipMemcpyDeviceToHost, st));
        }
    }
    FQCHK(hipEventRecord(ev2, st));
    FQCHK(hipStreamSynchronize(st));
    { float a = 0, b = 0; (void)hipEventElapsedTime(&a, ev0, ev1); (void)hipEventElapsedTime(&b, ev1, ev2); out.h2d_ms = uploaded ? uploaded->h2d_ms : a; out.kernels_ms = b; }
    out.bases_bytes = sc.keep(bases); out.seg_off_bytes = sc.keep(seg_off);
    out.d_bases = bases; out.d_seg_off = seg_off;
    out.n_seg = n_seg; out.n_bases = n_bases; out.n_reads = n_reads; out.n_input_bases = h[3];
    return 0;
}

int gpu_upload_text(const uint8_t *t, size_t n, int device, GpuText &out, std::string &err) {
    out = GpuText();
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); err = "hipSetDevice failed"; return -5; }
    out.e = trimmed_len(t, n);
    out.unterminated = out.e && t[out.e - 1] != '\n';
    size_t bytes = out.e + 32;
    out.d = (uint8_t *)device_pool_alloc(bytes);
    if (!out.d) { err = "out of device memory for the FASTQ text"; return -4; }
    out.pool_bytes = bytes;
    // (creating and destroying a stream costs ~0.4 ms each: one upload stream per device is kept for the process;
    // uploads are issued by one thread at a time per handle, and two handles sharing the stream merely queue up)
    static std::mutex mu;
    static hipStream_t up_streams[64] = {};
    hipStream_t s = nullptr;
    hipError_t e = hipSuccess;
    {
        std::lock_guard<std::mutex> lk(mu);
        const int di = device >= 0 && device < 64 ? device : 0;
        if (!up_streams[di]) e = hipStreamCreateWithFlags(&up_streams[di], hipStreamNonBlocking);
        s = up_streams[di];
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (e == hipSuccess && out.e) e = hipMemcpyAsync(out.d, t, out.e, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(out.d + out.e, 0, 32, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    out.h2d_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (e != hipSuccess) { err = std::string("upload of the FASTQ text: ") + hipGetErrorString(e); gpu_text_free(out); return -5; }
    return 0;
}
void gpu_text_free(GpuText &t) {
    if (t.d) device_pool_release(t.d, t.pool_bytes);
    t = GpuText();
}

void gpu_packed_free(GpuPacked &p) {
    if (p.d_bases) device_pool_release(p.d_bases, p.bases_bytes);
    if (p.d_seg_off) device_pool_release(p.d_seg_off, p.seg_off_bytes);
    p.d_bases = nullptr; p.d_seg_off = nullptr;
}

}  // namespace shk
