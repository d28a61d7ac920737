// Which VALU wave-instructions issue at what rate on one SIMD of gfx950 with 4 waves resident (1024-thread workgroup
// per CU, like k_partition / k_count_partitions)?  Sizes the instruction-issue floor of the count step.
//
// Round 3 ("weak" 6): every probed instruction is an inline-asm statement (nothing for the compiler to fold
// or re-associate: the ISA of the loop is exactly 64 probed instructions per trip, 8 independent chains), float and
// packed rows are measured beside the integer ones, and the clock is READ: cycles are shader-clock ticks (s_memtime)
// taken inside the kernel around the loop, the GHz is s_memtime / s_memrealtime (100 MHz), not a nominal 2.4.
// /opt/skills/guides/MI355X_MICROARCH.md:473 gives `v_fma_f32` 2 cycles per wave64 instruction with several waves
// resident (SIMD-32) and 4 for one wave alone; this tool says what the INTEGER instructions of the count step get.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CHK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); return 1; } } while (0)
static constexpr int REPS = 8;

#define OP1(str) asm volatile(str : "+v"(a[i]) : "v"(y), "v"(z))
template <int OP>
__global__ __launch_bounds__(1024) void k_valu(uint32_t *out, unsigned long long *clk, int iters, uint32_t seed) {
    uint32_t a[8];
    unsigned long long p[8];                               // 64-bit chains for the packed forms
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 31u + i; p[i] = ((unsigned long long)a[i] << 32) | (a[i] ^ 0x3f800000u); }
    const uint32_t y = seed | 1u, z = seed * 3u + 5u;
    const unsigned long long yy = 0x3f8000003f800000ull, zz = 0x3f0000003f000000ull;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) OP1("v_xor_b32 %0, %0, %1");
                else if (OP == 1) OP1("v_add_u32 %0, %0, %1");
                else if (OP == 2) OP1("v_alignbit_b32 %0, %0, %1, 31");
                else if (OP == 3) OP1("v_min_u32 %0, %0, %1");
                else if (OP == 4) OP1("v_and_or_b32 %0, %0, %1, %2");
                else if (OP == 5) OP1("v_bfe_u32 %0, %0, 3, 9");
                else if (OP == 6) OP1("v_mul_lo_u32 %0, %0, %1");
                else if (OP == 7) OP1("v_fma_f32 %0, %0, %1, %2");
                else if (OP == 8) OP1("v_add_f32 %0, %0, %1");
                else if (OP == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(yy), "v"(zz));
                else if (OP == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(yy));
                else if (OP == 11) OP1("v_pk_add_u16 %0, %0, %1");
                else if (OP == 12) OP1("v_lshl_add_u32 %0, %0, 3, %1");
                else if (OP == 13) OP1("v_xad_u32 %0, %0, %1, %2");
                else if (OP == 14) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(y), "v"(z) : "vcc");
                else if (OP == 15) OP1("v_min3_u32 %0, %0, %1, %2");
                else if (OP == 16) OP1("v_and_b32 %0, %0, %1");
                else if (OP == 17) OP1("v_or_b32 %0, %0, %1");
                else if (OP == 18) OP1("v_lshlrev_b32 %0, 3, %0");
                else if (OP == 19) OP1("v_lshrrev_b32 %0, 3, %0");
                else if (OP == 20) OP1("v_sub_u32 %0, %0, %1");
                else if (OP == 21) OP1("v_max_u32 %0, %0, %1");
                else if (OP == 22) OP1("v_cndmask_b32 %0, %0, %1, vcc");
                else if (OP == 23) OP1("v_mov_b32 %0, %1");
                else if (OP == 24) OP1("v_perm_b32 %0, %0, %1, %2");
                else if (OP == 25) OP1("v_or3_b32 %0, %0, %1, %2");
                else if (OP == 27) OP1("v_add3_u32 %0, %0, %1, %2");
                else if (OP == 28) OP1("v_lshl_or_b32 %0, %0, 3, %1");
                else if (OP == 29) OP1("v_min_f32 %0, %0, %1");
                else if (OP == 30) OP1("v_mul_f32 %0, %0, %1");
                else if (OP == 31) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(y) : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i] ^ (uint32_t)p[i] ^ (uint32_t)(p[i] >> 32);
    if (r == 0x12345u) out[0] = r;
}
template <int OP> int run(const char *name, int ninstr, uint32_t *d, unsigned long long *dclk, int cus, int waves_per_simd) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_valu<OP>, dim3(cus), dim3(256 * waves_per_simd), 0, 0, d, dclk, iters, 7u);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(2 * cus);
    CHK(hipMemcpy(h.data(), dclk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, ghz;
    for (int b = 0; b < cus; b++) { cyc.push_back((double)h[2 * b]); ghz.push_back(h[2 * b + 1] ? (double)h[2 * b] / (double)h[2 * b + 1] * 0.1 : 0.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double per_simd = (double)iters * 8 * REPS * ninstr * waves_per_simd;      // wave-instructions one SIMD issued
    // (the ticks of one wave's own loop say how fast THAT wave ran — the oldest wave of a SIMD is served first — not how
    // fast the SIMD issued: the rate is the kernel's wall time times the clock the kernel itself read)
    const double g = ghz[cus / 2];
    printf("%-34s %d waves/SIMD  %.3f ms  in-kernel clock %.2f GHz  %.2f clocks per wave-instruction per SIMD  (oldest wave alone: %.2f)\n",
           name, waves_per_simd, ms, g, ms * 1e6 * g / per_simd, cyc[cus / 2] / ((double)iters * 8 * REPS * ninstr));
    return 0;
}
int main() {
    uint32_t *d; CHK(hipMalloc(&d, 4));
    int cus = 0; CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned long long *dclk; CHK(hipMalloc(&dclk, (size_t)cus * 16));
    for (int w : {4, 1}) {
        run<7>("v_fma_f32", 1, d, dclk, cus, w);
        run<8>("v_add_f32", 1, d, dclk, cus, w);
        run<9>("v_pk_fma_f32", 1, d, dclk, cus, w);
        run<10>("v_pk_add_f32", 1, d, dclk, cus, w);
        run<0>("v_xor_b32", 1, d, dclk, cus, w);
        run<1>("v_add_u32", 1, d, dclk, cus, w);
        run<2>("v_alignbit_b32", 1, d, dclk, cus, w);
        run<3>("v_min_u32", 1, d, dclk, cus, w);
        run<15>("v_min3_u32", 1, d, dclk, cus, w);
        run<4>("v_and_or_b32", 1, d, dclk, cus, w);
        run<5>("v_bfe_u32", 1, d, dclk, cus, w);
        run<12>("v_lshl_add_u32", 1, d, dclk, cus, w);
        run<13>("v_xad_u32", 1, d, dclk, cus, w);
        run<11>("v_pk_add_u16", 1, d, dclk, cus, w);
        run<6>("v_mul_lo_u32", 1, d, dclk, cus, w);
        run<14>("v_cmp_lt_u32 + v_cndmask_b32 (2)", 2, d, dclk, cus, w);
        run<31>("v_cmp_lt_u32", 1, d, dclk, cus, w);
        run<22>("v_cndmask_b32", 1, d, dclk, cus, w);
        run<16>("v_and_b32", 1, d, dclk, cus, w);
        run<17>("v_or_b32", 1, d, dclk, cus, w);
        run<18>("v_lshlrev_b32", 1, d, dclk, cus, w);
        run<19>("v_lshrrev_b32", 1, d, dclk, cus, w);
        run<20>("v_sub_u32", 1, d, dclk, cus, w);
        run<21>("v_max_u32", 1, d, dclk, cus, w);
        run<23>("v_mov_b32", 1, d, dclk, cus, w);
        run<24>("v_perm_b32", 1, d, dclk, cus, w);
        run<25>("v_or3_b32", 1, d, dclk, cus, w);
        run<27>("v_add3_u32", 1, d, dclk, cus, w);
        run<28>("v_lshl_or_b32", 1, d, dclk, cus, w);
        run<29>("v_min_f32", 1, d, dclk, cus, w);
        run<30>("v_mul_f32", 1, d, dclk, cus, w);
    }
    return 0;
}
