// valu_rate.hip -- issue rate of the integer instructions the table-key election is made of (gfx950): v_mul_lo_u32 against
// v_mul_u32_u24, v_alignbit_b32, v_min_u32, v_lshl_or_b32. Eight independent chains per lane, 256 lanes x many workgroups;
// prints wave-instructions per cycle and SIMD. hipcc --offload-arch=gfx950 -O3 tools/debug/valu_rate.hip -o tools/debug/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ void __launch_bounds__(256) chain(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a[8];
    for (int j = 0; j < 8; ++j) a[j] = seed + threadIdx.x * 8 + j;
    uint32_t c = seed | 1u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 6" : "+v"(a[j]) : "v"(c));
            if (OP == 3) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 4) asm volatile("v_lshl_or_b32 %0, %0, 6, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 7) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        }
    }
    uint32_t s = 0;
    for (int j = 0; j < 8; ++j) s ^= a[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void run(const char* name, uint32_t* out, int clock_khz) {
    const int iters = 4096, blocks = 256 * 8 * 4;  // 8 workgroups (32 waves) per CU, four rounds
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chain<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = double(blocks) * 4 * iters * 8;
    const double cycles = ms * 1e-3 * clock_khz * 1e3;
    printf("{\"op\": \"%s\", \"ms\": %.3f, \"wave_instr_per_cycle_per_simd\": %.3f, \"cycles_per_wave_instr\": %.2f}\n", name, ms,
           wave_instr / cycles / (256 * 4), cycles * 256 * 4 / wave_instr);
}

int main() {
    uint32_t* out;
    hipMalloc(&out, 256 * 8 * 4 * 256 * 4);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    printf("{\"clock_khz\": %d}\n", khz);
    run<5>("v_add_u32", out, khz);
    run<0>("v_mul_lo_u32", out, khz);
    run<7>("v_mul_hi_u32", out, khz);
    run<1>("v_mul_u32_u24", out, khz);
    run<6>("v_mad_u32_u24", out, khz);
    run<2>("v_alignbit_b32", out, khz);
    run<3>("v_min_u32", out, khz);
    run<4>("v_lshl_or_b32", out, khz);
    return 0;
}
