// m0_check.hip -- is M0 latched when an LDS-DMA load is issued, or may a following s_mov_b32 m0 redirect it?
// Four global_load_lds_dwordx4 into four regions with M0 rewritten right after each load (GAP s_nop between the load and
// the next M0 write), under memory pressure; afterwards every region must hold the line of its own round.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__host__ __device__ inline uint32_t word_of(uint64_t line, uint32_t w) {
    uint64_t x = line * 16 + w + 0x9E3779B97F4A7C15ULL; x ^= x >> 31; x *= 0xD6E8FEB86659FD93ULL; x ^= x >> 29; return uint32_t(x);
}
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; return x; }
__global__ void fill(uint32_t* a, uint64_t n_words) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n_words) a[i] = word_of(i / 16, uint32_t(i % 16));
}
template <int GAP>
__global__ void __launch_bounds__(256) k(const char* __restrict__ a, uint32_t n_lines, int rounds, unsigned long long* bad) {
    __shared__ uint4 lds[256 * 4];
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, sub = lane & 3u;
    const uint32_t base = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6))) * 4096u;
    const uint4* ws = lds + (threadIdx.x >> 6) * 256;
    uint32_t wrong = 0;
    for (int r = 0; r < rounds; ++r) {
        uint32_t line[4];
        const char* p[4];
        for (int o = 0; o < 4; ++o) {  // line of the quad's lane o, the same for the four lanes of the quad
            const uint64_t owner = (tid & ~uint64_t(3)) + o;
            line[o] = uint32_t((__uint128_t(mix(owner * 0x9E3779B97F4A7C15ULL + r)) * n_lines) >> 64);
            p[o] = a + uint64_t(line[o]) * 64 + 16 * sub;
        }
        const uint32_t m1 = base + 0x400, m2 = base + 0x800, m3 = base + 0xc00;
        if constexpr (GAP == 0) {
            asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                         "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                         "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                         "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                         "s_waitcnt vmcnt(0)\n\t"
                         :: "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "s"(base), "s"(m1), "s"(m2), "s"(m3) : "memory");
        } else {
            asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_nop 7\n\ts_nop 7\n\t"
                         "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_nop 7\n\ts_nop 7\n\t"
                         "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_nop 7\n\ts_nop 7\n\t"
                         "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                         "s_waitcnt vmcnt(0)\n\t"
                         :: "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "s"(base), "s"(m1), "s"(m2), "s"(m3) : "memory");
        }
        const uint4 v = ws[sub * 64 + (lane >> 2) * 4 + (r & 3)];  // my own line (round = my position in the quad), piece r & 3
        const uint32_t w = 4 * (r & 3);
        wrong += v.x != word_of(line[sub], w) || v.y != word_of(line[sub], w + 1) || v.z != word_of(line[sub], w + 2) || v.w != word_of(line[sub], w + 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (wrong) atomicAdd(bad, (unsigned long long)wrong);
}
int main(int argc, char** argv) {
    const uint64_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4096;
    const uint64_t lanes = argc > 2 ? strtoull(argv[2], nullptr, 10) : (1ull << 25);
    const int rounds = argc > 3 ? atoi(argv[3]) : 8;
    const uint64_t bytes = mib << 20;
    char* a = nullptr; unsigned long long* bad = nullptr;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&bad, 8));
    for (uint64_t off = 0; off < bytes / 4; off += (1ull << 30))
        hipLaunchKernelGGL(fill, dim3(uint32_t((std::min<uint64_t>(1ull << 30, bytes / 4 - off) + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(a) + off, std::min<uint64_t>(1ull << 30, bytes / 4 - off));
    CHECK(hipDeviceSynchronize());
    for (int gap = 0; gap < 2; ++gap) for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(bad, 0, 8));
        if (gap) hipLaunchKernelGGL(k<1>, dim3(uint32_t(lanes / 256)), dim3(256), 0, 0, a, uint32_t(bytes / 64), rounds, bad);
        else hipLaunchKernelGGL(k<0>, dim3(uint32_t(lanes / 256)), dim3(256), 0, 0, a, uint32_t(bytes / 64), rounds, bad);
        unsigned long long h = 0; CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        printf("{\"m0_rewritten\": \"%s\", \"rep\": %d, \"checked\": %llu, \"mismatches\": %llu}\n", gap ? "16 wait states after the load" : "right after the load", rep, (unsigned long long)(lanes * rounds), h);
    }
    return 0;
}
