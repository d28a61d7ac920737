// Which VALU wave-instructions issue at what rate on one SIMD of gfx950 with 4 waves resident (1024-thread workgroup
// per CU, like k_partition / k_count_partitions)?  Sizes the instruction-issue floor of the count step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); return 1; } } while (0)
static constexpr int REPS = 8;
template <int OP>
__global__ __launch_bounds__(1024) void k_valu(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 31u + i;
    const uint32_t c = seed | 1u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {              // REPS x 8 independent VALU per trip: the loop's scalar overhead is amortised
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t x = a[i], y = c + (uint32_t)(i + rep);   // 8 independent chains per wave
            if (OP == 0) x = x ^ y;                                            // v_xor_b32
            else if (OP == 1) x = x + y;                                       // v_add_u32
            else if (OP == 2) x = __builtin_amdgcn_alignbit(x, y, 31);         // v_alignbit_b32
            else if (OP == 3) x = (x << 3) | y;                                // v_lshl_or_b32
            else if (OP == 4) x = min(x, y);                                   // v_min_u32
            else if (OP == 5) x = (x & c) | y;                                 // v_and_or_b32
            else if (OP == 6) x = x * (c + rep);                               // v_mul_lo_u32
            else if (OP == 7) x = __builtin_amdgcn_ubfe(x, 3, 9) + y;          // v_bfe_u32 + add (2 instr)
            else if (OP == 8) x = (x < y) ? x : (y ^ c);                       // v_cmp + v_cndmask (+xor)
            a[i] = x;
        }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i];
    if (r == 0x12345u) out[0] = r;
}
template <int OP> int run(const char *name, int ninstr, uint32_t *d, int cus) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_valu<OP>, dim3(cus), dim3(1024), 0, 0, d, iters, 7u);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double per_simd = (double)iters * 8 * REPS * ninstr * 4;      // 4 waves per SIMD
    printf("%-28s %.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e6 * 2.4 / per_simd);
    return 0;
}
int main() {
    uint32_t *d; CHK(hipMalloc(&d, 4));
    int cus = 0; CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    // (only the forms the compiler cannot fold across the unrolled repetitions: check the ISA when adding one —
    // 64 v_alignbit / 64 v_bfe + 64 v_add / 64 v_cmp + 64 v_cndmask per trip)
    run<2>("v_alignbit_b32", 1, d, cus);
    run<7>("v_bfe_u32 + v_add_u32 (2)", 2, d, cus);
    run<8>("v_cmp + v_cndmask (2)", 2, d, cus);
    return 0;
}
