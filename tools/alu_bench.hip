// alu_bench.hip -- issue cost of the integer instructions the lookup kernels are made of (gfx950).
// Each test keeps 8 independent dependency chains per lane busy with one opcode; all SIMDs are filled
// with 8 waves. Prints cycles per wave-instruction per SIMD (2.0 = full rate on a SIMD-32, 8.0 = quarter).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096, CHAINS = 8;

template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t* out, uint32_t seed) {
    uint64_t a[CHAINS];
    uint32_t b[CHAINS];
    for (int j = 0; j < CHAINS; ++j) {
        a[j] = (uint64_t(threadIdx.x) << 32 | seed) * (2 * j + 1);
        b[j] = threadIdx.x * 2654435761u + j + seed;
    }
    const uint32_t s = (seed & 15) + 1;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < CHAINS; ++j) {
            if constexpr (OP == 0) b[j] = b[j] & (b[(j + 1) % CHAINS] | 0x55555555u);             // v_and/or
            if constexpr (OP == 1) b[j] = b[j] * seed;                                                // v_mul_lo_u32
            if constexpr (OP == 2) a[j] = uint64_t(uint32_t(a[j])) * seed + (a[j] >> 32);             // v_mad_u64_u32
            if constexpr (OP == 3) a[j] = (a[j] >> s) ^ a[j];                                         // v_lshrrev_b64 (+xor)
            if constexpr (OP == 4) b[j] = __builtin_amdgcn_alignbit(b[j], b[(j + 1) % CHAINS], s);    // v_alignbit_b32
            if constexpr (OP == 5) a[j] = a[j] < a[(j + 1) % CHAINS] ? a[j] + 1 : a[(j + 1) % CHAINS]; // v_cmp_lt_u64 + cndmask x2
            if constexpr (OP == 6) b[j] = __umulhi(b[j], seed);                                        // v_mul_hi_u32
            if constexpr (OP == 7) b[j] = b[j] == seed ? b[(j + 1) % CHAINS] : b[j] + 1;               // v_cmp_eq_u32 + cndmask
            if constexpr (OP == 8) a[j] = a[j] * 0x517cc1b727220a95ULL;                                 // 64-bit multiply by constant
        }
    }
    uint64_t r = 0;
    for (int j = 0; j < CHAINS; ++j) r += a[j] + b[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    const int blocks = 256 * 8;  // 8 blocks of 256 threads per CU = 8 waves per SIMD
    uint64_t* out;
    CHECK(hipMalloc(&out, size_t(blocks) * 256 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    int clock_khz = 0;
    CHECK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
    const char* names[] = {"v_and+v_or (2 ops)", "v_mul_lo_u32", "v_mad_u64_u32 (+shift)", "v_lshrrev_b64 + xor64", "v_alignbit_b32",
                           "v_cmp_lt_u64 + add64 + 2 cndmask", "v_mul_hi_u32", "v_cmp_eq_u32 + add + cndmask", "u64 * const (mad+2 mul_lo+add3)"};
    for (int op = 0; op < 9; ++op) {
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CHECK(hipEventRecord(e0));
            switch (op) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, 12345u + r); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 12345u + r); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, 12345u + r); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, 12345u + r); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, 12345u + r); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, out, 12345u + r); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, out, 12345u + r); break;
                case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, out, 12345u + r); break;
                case 8: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, out, 12345u + r); break;
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // wave-level "statement" executions per SIMD: 8 waves * ITERS * CHAINS
        const double per_simd = 8.0 * ITERS * CHAINS;
        const double cycles = best * 1e-3 * (clock_khz * 1e3);
        printf("{\"op\": \"%s\", \"ms\": %.3f, \"cycles_per_statement_per_simd\": %.2f, \"clock_MHz\": %d}\n", names[op], best,
               cycles / per_simd, clock_khz / 1000);
    }
    return 0;
}
