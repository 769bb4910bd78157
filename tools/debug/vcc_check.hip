// vcc_check.hip -- does an LDS-DMA load (global_load_lds_dwordx4) leave VCC alone? One asm block sets VCC from a per-lane
// predicate, issues the load (variant 1) or an ordinary global_load_dwordx4 (variant 0), waits, and reads VCC back.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; return x; }

template <int VARIANT>
__global__ void __launch_bounds__(256) k(const char* __restrict__ a, uint64_t n_lines, int rounds, unsigned long long* bad, unsigned long long* waves) {
    __shared__ uint4 lds[256 * 4];
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint32_t lds_base = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6))) * 4096u;
    uint32_t wrong = 0;
    uint4 sink = make_uint4(0, 0, 0, 0);
    for (int r = 0; r < rounds; ++r) {
        const uint64_t x = mix(tid * 0x9E3779B97F4A7C15ULL + r);
        const char* p = a + (uint64_t((__uint128_t(x) * n_lines) >> 64)) * 64 + 16 * (threadIdx.x & 3);
        const uint32_t pred = uint32_t(mix(x + 5) & 1);
        uint64_t before, after;
        if constexpr (VARIANT == 2) {
            /* the staging sequence of sk_stage_buckets as hipcc schedules it: DPP broadcasts, readfirstlane -> M0,
               four LDS-DMA loads, and VALU work in between */
            uint32_t t0, t1, t2, sm;
            const uint32_t ldsv = (threadIdx.x << 6) & 0x3000u;
            asm volatile(
                "v_cmp_ne_u32_e32 vcc, 0, %[pred]\n\t"
                "s_mov_b64 %[before], vcc\n\t"
                "v_mov_b32_dpp %[t0], %[pred] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp %[t1], %[pred] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_readfirstlane_b32 %[sm], %[ldsv]\n\t"
                "v_or_b32_e32 %[t2], 0x400, %[ldsv]\n\t"
                "s_mov_b32 m0, %[sm]\n\t"
                "v_add_u32_e32 %[t0], %[t0], %[t1]\n\t"
                "v_readfirstlane_b32 %[sm], %[t2]\n\t"
                "global_load_lds_dwordx4 %[addr], off\n\t"
                "v_or_b32_e32 %[t2], 0x800, %[ldsv]\n\t"
                "s_mov_b32 m0, %[sm]\n\t"
                "v_mov_b32_e32 %[t1], %[t0]\n\t"
                "global_load_lds_dwordx4 %[addr], off\n\t"
                "v_readfirstlane_b32 %[sm], %[t2]\n\t"
                "v_or_b32_e32 %[t2], 0xc00, %[ldsv]\n\t"
                "s_mov_b32 m0, %[sm]\n\t"
                "v_mov_b32_e32 %[t1], %[t0]\n\t"
                "global_load_lds_dwordx4 %[addr], off\n\t"
                "v_readfirstlane_b32 %[sm], %[t2]\n\t"
                "s_nop 0\n\t"
                "s_mov_b32 m0, %[sm]\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %[addr], off\n\t"
                "s_waitcnt vmcnt(0)\n\t"
                "s_mov_b64 %[after], vcc\n\t"
                : [before] "=&s"(before), [after] "=&s"(after), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [sm] "=&s"(sm)
                : [pred] "v"(pred), [ldsv] "v"(ldsv), [addr] "v"(p)
                : "vcc", "memory");
        } else if constexpr (VARIANT == 1) {
            asm volatile(
                "v_cmp_ne_u32_e32 vcc, 0, %[pred]\n\t"
                "s_mov_b64 %[before], vcc\n\t"
                "s_mov_b32 m0, %[lds]\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %[addr], off\n\t"
                "s_waitcnt vmcnt(0)\n\t"
                "s_mov_b64 %[after], vcc\n\t"
                : [before] "=&s"(before), [after] "=&s"(after)
                : [pred] "v"(pred), [lds] "s"(lds_base), [addr] "v"(p)
                : "vcc", "memory");
        } else {
            asm volatile(
                "v_cmp_ne_u32_e32 vcc, 0, %[pred]\n\t"
                "s_mov_b64 %[before], vcc\n\t"
                "global_load_dwordx4 %[dst], %[addr], off\n\t"
                "s_waitcnt vmcnt(0)\n\t"
                "s_mov_b64 %[after], vcc\n\t"
                : [before] "=&s"(before), [after] "=&s"(after), [dst] "=&v"(sink)
                : [pred] "v"(pred), [addr] "v"(p)
                : "vcc", "memory");
        }
        wrong += before != after;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(waves, (unsigned long long)rounds);
        if (wrong) atomicAdd(bad, (unsigned long long)wrong);
    }
    if (sink.x == 0x12345678 && lds[threadIdx.x].x == 77) bad[1] = 1;
}

int main(int argc, char** argv) {
    const uint64_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4096;
    const uint64_t lanes = argc > 2 ? strtoull(argv[2], nullptr, 10) : (1ull << 24);
    const int rounds = argc > 3 ? atoi(argv[3]) : 8;
    char* a = nullptr;
    unsigned long long* c = nullptr;
    CHECK(hipMalloc(&a, mib << 20));
    CHECK(hipMemset(a, 1, mib << 20));
    CHECK(hipMalloc(&c, 32));
    for (int variant = 0; variant < 3; ++variant) {
        CHECK(hipMemset(c, 0, 32));
        if (variant == 2) hipLaunchKernelGGL(k<2>, dim3(uint32_t(lanes / 256)), dim3(256), 0, 0, a, (mib << 20) / 64, rounds, c, c + 2);
        else if (variant) hipLaunchKernelGGL(k<1>, dim3(uint32_t(lanes / 256)), dim3(256), 0, 0, a, (mib << 20) / 64, rounds, c, c + 2);
        else hipLaunchKernelGGL(k<0>, dim3(uint32_t(lanes / 256)), dim3(256), 0, 0, a, (mib << 20) / 64, rounds, c, c + 2);
        unsigned long long h[4];
        CHECK(hipMemcpy(h, c, 32, hipMemcpyDeviceToHost));
        printf("{\"variant\": \"%s\", \"wave_rounds\": %llu, \"vcc_changed\": %llu}\n", variant == 2 ? "staging_sequence" : variant ? "global_load_lds_dwordx4" : "global_load_dwordx4", h[2], h[0]);
    }
    return 0;
}
