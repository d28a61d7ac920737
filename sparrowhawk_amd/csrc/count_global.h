// count_global.h — the first HIP path: one global-memory hash table with an atomic per k-mer instance;
// kept as SHK_COUNT_MODE_GLOBAL=1 for the counting-mode cross-check test and as the measured baseline (profiles/r01_baseline_global_atomics)
// (included by pipeline.hip inside namespace shk, after the device-side views and count_part.h)
#pragma once

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

