// alloc_modes.hip -- does the random-line rate of a large hipMalloc block depend on WHICH allocation it is (round 6: two processes of one box
// gave the same replica 37.0 and 41.0 G lookups/s)? Blocks of <GiB> are allocated one after the other and HELD, each probed with 2^27 random
// 64-byte lines fetched by quads (the table's bucket fetch); then all are freed and the round repeats.   alloc_modes [GiB=37] [blocks=5] [rounds=2] [contiguous 0|1]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; return x; }
__global__ void __launch_bounds__(256) probe(const char* __restrict__ a, uint64_t n_lines, uint32_t* __restrict__ out, uint64_t salt) {
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t x = mix((tid >> 2) * 0x9E3779B97F4A7C15ULL + salt);
    const uint64_t line = uint64_t((__uint128_t(x) * n_lines) >> 64);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a + line * 64 + 16 * (tid & 3)));
    out[tid] = v.x ^ v.w;
}
int main(int argc, char** argv) {
    const uint64_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 37, blocks = argc > 2 ? strtoull(argv[2], nullptr, 10) : 5, rounds = argc > 3 ? strtoull(argv[3], nullptr, 10) : 2;
    const bool contiguous = argc > 4 && argv[4][0] == '1';  // hipExtMallocWithFlags(hipDeviceMallocContiguous): physically contiguous memory
    const uint64_t bytes = gib << 30, lanes = uint64_t(1) << 29;  // 2^27 lines x 4 lanes
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&out, lanes * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (uint64_t r = 0; r < rounds; ++r) {
        std::vector<char*> held;
        for (uint64_t b = 0; b < blocks; ++b) {
            char* p = nullptr;
            if (contiguous) CHECK(hipExtMallocWithFlags(reinterpret_cast<void**>(&p), bytes, hipDeviceMallocContiguous));
            else CHECK(hipMalloc(&p, bytes));
            CHECK(hipMemset(p, 1, bytes));
            held.push_back(p);
            float best = 1e9f;
            for (int t = 0; t < 4; ++t) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(probe, dim3(uint32_t(lanes / 256)), dim3(256), 0, 0, p, bytes / 64, out, 0x1234567ull * (t + 1));
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (t && ms < best) best = ms;
            }
            printf("%sround %llu block %llu at %p: %.2f G lines/s (%.3f ms per 2^27 random 64-byte lines)\n", contiguous ? "contiguous " : "", (unsigned long long)r, (unsigned long long)b, (void*)p,
                   double(lanes / 4) / best / 1e6, best);
            fflush(stdout);
        }
        for (char* p : held) CHECK(hipFree(p));
    }
    return 0;
}
