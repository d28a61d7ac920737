// fastq_gpu.hip — FASTQ text -> quality-masked, segmented, 2-bit packed reads, on the device.
//
// the reference ingests FASTQ inside preprocess() (web_sys::File -> gz sniff
// -> seq_io reader: /root/reference/AGENTS.md:180-183, sibling rust/orphos-bridge/src/fastx_wasm.rs:53-70;
// caller www/src/workers/Assembler.ts:100).  The host packer (fastq.cpp) does this at ~0.3 Gbases/s
// on one core, 300x slower than the kernels behind it.  Here the text is uploaded as it is and
// parsed by byte-streaming kernels (all HBM-bound):
//   k_nl_count / k_nl_fill   newline positions (line index)
//   k_read_scan              per record: framing checks (SPEC S1), valid-base runs (SPEC S2) -> counts
//   k_read_emit              segment table: stream offset + text position of every kept run
//   k_pack_words             one lane per 16-base output word: gather + encode
// Only REGULAR input is handled (fixed 4-line framing, no blank lines between records); anything
// else — and every malformed record — is left to the host parser, which owns the error messages.
// The result is the same packed layout bit for bit (tests/test_gpu_parity.py compares both).
#include <hip/hip_runtime.h>
#include <stdint.h>
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
    ~Scratch() { for (auto &p : ptrs) if (p.first) device_pool_release(p.first, p.second); }
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
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t x = w[j] ^ 0x0A0A0A0Au;                    // zero byte where '\n'
                c += ((x & 0xFFu) == 0) + ((x & 0xFF00u) == 0) + ((x & 0xFF0000u) == 0) + ((x & 0xFF000000u) == 0);
            }
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

// line_end[j] = position of the j-th '\n' (in text order).  One workgroup per chunk; inside a chunk the
// order is kept by scanning 64-byte pieces wave by wave.
__global__ __launch_bounds__(256) void k_nl_fill(const uint8_t *__restrict__ text, uint64_t n,
                                                 const unsigned long long *__restrict__ chunk_off,
                                                 unsigned long long *__restrict__ line_end) {
    __shared__ uint32_t piece_cnt[NL_CHUNK / 64];          // newlines per 64-byte piece
    __shared__ uint32_t piece_off[NL_CHUNK / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * NL_CHUNK;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // pass 1: one wave per 64-byte piece (lane = byte)
    for (uint32_t pc = wid; pc < NL_CHUNK / 64; pc += 4) {
        const uint64_t p = b0 + (uint64_t)pc * 64 + lane;
        const bool nl = p < n && text[p] == '\n';
        const unsigned long long m = __ballot(nl);
        if (lane == 0) piece_cnt[pc] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    // exclusive scan of the 256 piece counts (one per thread)
    {
        const uint32_t v = piece_cnt[threadIdx.x];
        uint32_t incl = v;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += u; }
        __shared__ uint32_t wtot[4];
        if (lane == 63) wtot[wid] = incl;
        __syncthreads();
        uint32_t off = 0;
        for (int w = 0; w < wid; w++) off += wtot[w];
        piece_off[threadIdx.x] = off + incl - v;
    }
    __syncthreads();
    const unsigned long long base = chunk_off[blockIdx.x];
    for (uint32_t pc = wid; pc < NL_CHUNK / 64; pc += 4) {
        const uint64_t p = b0 + (uint64_t)pc * 64 + lane;
        const bool nl = p < n && text[p] == '\n';
        const unsigned long long m = __ballot(nl);
        if (nl) line_end[base + piece_off[pc] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = p;
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

// mode 0: counts (segments, kept bases) per record + validation; mode 1: write the segment table
template <int MODE>
__global__ __launch_bounds__(256) void k_read_scan(const uint8_t *__restrict__ text,
                                                   const unsigned long long *__restrict__ line_end, FqParams fp,
                                                   uint64_t n_reads, unsigned long long *__restrict__ seg_cnt,
                                                   unsigned long long *__restrict__ base_cnt,
                                                   unsigned long long *__restrict__ stats /* [0]=bad flag [1]=input bases */,
                                                   uint32_t *__restrict__ seg_off, unsigned long long *__restrict__ seg_src) {
    unsigned long long my_in = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t s0, e0, s1, e1, s2, e2, s3, e3;
        line_span(text, line_end, fp, 4 * r + 1, s1, e1);
        line_span(text, line_end, fp, 4 * r + 3, s3, e3);
        if (MODE == 0) {
            line_span(text, line_end, fp, 4 * r, s0, e0);
            line_span(text, line_end, fp, 4 * r + 2, s2, e2);
            const bool ok = e0 > s0 && text[s0] == '@' && e2 > s2 && text[s2] == '+' && (e1 - s1) == (e3 - s3);
            if (!ok) { stats[0] = 1; seg_cnt[r] = 0; base_cnt[r] = 0; continue; }
            my_in += e1 - s1;
        }
        const uint64_t L = e1 - s1;
        uint64_t run = 0, nseg = 0, nb = 0;
        uint64_t so = MODE == 1 ? seg_cnt[r] : 0, bo = MODE == 1 ? base_cnt[r] : 0;
        for (uint64_t i = 0; i <= L; i++) {
            bool valid = false;
            if (i < L) valid = base_code(text[s1 + i]) < 4 && (int)text[s3 + i] - 33 >= (int)fp.min_qual;
            if (valid) { run++; continue; }
            if (run >= fp.k) {
                if (MODE == 1) { seg_off[so + nseg] = (uint32_t)(bo + nb); seg_src[so + nseg] = s1 + i - run; }
                nseg++; nb += run;
            }
            run = 0;
        }
        if (MODE == 0) { seg_cnt[r] = nseg; base_cnt[r] = nb; }
    }
    if (MODE == 0) {
        for (int o = 32; o > 0; o >>= 1) my_in += __shfl_down(my_in, o);
        if ((threadIdx.x & 63) == 0 && my_in) atomicAdd(&stats[1], my_in);
    }
}

// one lane per output word: bases [16w, 16w+16) of the stream, gathered from the text
__global__ __launch_bounds__(256) void k_pack_words(const uint8_t *__restrict__ text, const uint32_t *__restrict__ seg_off,
                                                    const unsigned long long *__restrict__ seg_src, uint64_t n_seg,
                                                    uint64_t n_bases, uint64_t n_words, uint32_t *__restrict__ out) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b0 = w * 16;
        uint32_t word = 0;
        if (b0 < n_bases) {
            // segment holding base b0: last s with seg_off[s] <= b0
            uint64_t lo = 0, hi = n_seg;
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if ((uint64_t)seg_off[mid] <= b0) lo = mid; else hi = mid; }
            uint64_t s = lo;
            uint64_t seg_end = s + 1 < n_seg ? seg_off[s + 1] : n_bases;       // seg_off[n_seg] == n_bases is written later
            uint64_t src = seg_src[s] + (b0 - seg_off[s]);
            for (uint32_t i = 0; i < 16 && b0 + i < n_bases; i++) {
                if (b0 + i >= seg_end) { s++; seg_end = s + 1 < n_seg ? seg_off[s + 1] : n_bases; src = seg_src[s]; }
                word |= base_code(text[src++]) << (2 * i);
            }
        }
        out[w] = word;
    }
}

__global__ void k_set_u32(uint32_t *p, uint32_t v) { *p = v; }

// byte offset (inside its own file) after record `every*(j+1)` for the progress strings; bit 63 = second file
__global__ __launch_bounds__(256) void k_progress_marks(const unsigned long long *__restrict__ line_end, FqParams fp,
                                                        uint64_t every, uint64_t n_marks, unsigned long long *__restrict__ marks) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_marks; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = every * (j + 1) - 1;                  // last record of the batch
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
                   uint64_t every, void *stream_v, GpuPacked &out, std::string &err) {
    hipStream_t st = (hipStream_t)stream_v;
    out = GpuPacked();
    Scratch sc;
    // ---- framing that can be decided on the host: trailing blank lines are ignored, the last line of a
    // file may lack its newline (one is supplied between the files)
    const size_t e1 = trimmed_len(t1, n1), e2 = t2 ? trimmed_len(t2, n2) : 0;
    const bool unterm1 = e1 && t1[e1 - 1] != '\n', unterm2 = e2 && t2[e2 - 1] != '\n';
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
    uint8_t *text = sc.get<uint8_t>(e + 32, err);
    if (!text) return -4;
    hipEvent_t ev0, ev1, ev2;
    FQCHK(hipEventCreate(&ev0)); FQCHK(hipEventCreate(&ev1)); FQCHK(hipEventCreate(&ev2));
    struct EvGuard { hipEvent_t a, b, c; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(c); } } evg{ev0, ev1, ev2};
    FQCHK(hipEventRecord(ev0, st));
    if (e1) FQCHK(hipMemcpyAsync(text, t1, e1, hipMemcpyHostToDevice, st));
    if (off2 > e1) FQCHK(hipMemsetAsync(text + e1, '\n', 1, st));
    if (e2) FQCHK(hipMemcpyAsync(text + off2, t2, e2, hipMemcpyHostToDevice, st));
    FQCHK(hipMemsetAsync(text + e, 0, 32, st));
    FQCHK(hipEventRecord(ev1, st));

    const uint64_t n_chunks = (e + NL_CHUNK - 1) / NL_CHUNK;
    unsigned long long *chunk_cnt = sc.get<unsigned long long>(n_chunks, err);
    unsigned long long *d_tot = sc.get<unsigned long long>(8, err);
    if (!chunk_cnt || !d_tot) return -4;
    FQCHK(hipMemsetAsync(d_tot, 0, 64, st));
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
    hipLaunchKernelGGL(k_read_scan<0>, dim3(grid1(n_reads)), dim3(256), 0, st, text, line_end, fp, n_reads, seg_cnt, base_cnt,
                       d_tot + 2, (uint32_t *)nullptr, (unsigned long long *)nullptr);
    if (int rc = exclusive_scan(seg_cnt, n_reads, d_tot + 4, st, sc, err)) return rc;
    if (int rc = exclusive_scan(base_cnt, n_reads, d_tot + 5, st, sc, err)) return rc;
    unsigned long long h[8];
    FQCHK(hipMemcpyAsync(h, d_tot, 64, hipMemcpyDeviceToHost, st));
    FQCHK(hipStreamSynchronize(st));
    if (h[2]) return 1;                                   // a malformed record: the host parser reports it
    const uint64_t n_seg = h[4], n_bases = h[5];
    if (n_bases >= 0xFFFFFFF0ull) { err = "input exceeds 2^32 bases per batch"; return -1; }
    const uint64_t n_words = (n_bases >> 4) + 2;          // partial word + one spare word (as PackedReads::finish)
    uint32_t *seg_off = sc.get<uint32_t>(n_seg + 1, err);
    unsigned long long *seg_src = sc.get<unsigned long long>(n_seg + 1, err);
    uint32_t *bases = sc.get<uint32_t>(n_words, err);
    if (!seg_off || !seg_src || !bases) return -4;
    if (n_seg) {
        hipLaunchKernelGGL(k_read_scan<1>, dim3(grid1(n_reads)), dim3(256), 0, st, text, line_end, fp, n_reads, seg_cnt, base_cnt,
                           d_tot + 2, seg_off, seg_src);
        hipLaunchKernelGGL(k_pack_words, dim3(grid1(n_words)), dim3(256), 0, st, text, seg_off, seg_src, n_seg, n_bases, n_words,
                           bases);
    } else {
        FQCHK(hipMemsetAsync(bases, 0, n_words * 4, st));
    }
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, seg_off + n_seg, (uint32_t)n_bases);
    FQCHK(hipGetLastError());
    // progress marks
    if (every) {
        const uint64_t n_marks = n_reads / every;
        if (n_marks) {
            unsigned long long *marks = sc.get<unsigned long long>(n_marks, err);
            if (!marks) return -4;
            hipLaunchKernelGGL(k_progress_marks, dim3(grid1(n_marks)), dim3(256), 0, st, line_end, fp, every, n_marks, marks);
            out.progress_bytes.resize(n_marks);
            FQCHK(hipMemcpyAsync(out.progress_bytes.data(), marks, n_marks * 8, hipMemcpyDeviceToHost, st));
        }
    }
    FQCHK(hipEventRecord(ev2, st));
    FQCHK(hipStreamSynchronize(st));
    { float a = 0, b = 0; (void)hipEventElapsedTime(&a, ev0, ev1); (void)hipEventElapsedTime(&b, ev1, ev2); out.h2d_ms = a; out.kernels_ms = b; }
    out.bases_bytes = sc.keep(bases); out.seg_off_bytes = sc.keep(seg_off);
    out.d_bases = bases; out.d_seg_off = seg_off;
    out.n_seg = n_seg; out.n_bases = n_bases; out.n_reads = n_reads; out.n_input_bases = h[3];
    return 0;
}

void gpu_packed_free(GpuPacked &p) {
    if (p.d_bases) device_pool_release(p.d_bases, p.bases_bytes);
    if (p.d_seg_off) device_pool_release(p.d_seg_off, p.seg_off_bytes);
    p.d_bases = nullptr; p.d_seg_off = nullptr;
}

}  // namespace shk
