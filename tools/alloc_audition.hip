// alloc_audition.hip -- is the random-line rate a property of the ALLOCATION? tools/tlb_probe, one process per run, gave 39.3 and 43.8 G lines/s
// for the same hipMalloc on the same box (profiles/r04/tlb_probe_cooperative_allocations.txt). Here ONE process holds several allocations
// at once and probes each in turn, twice: if the regions differ from one another and each keeps its own rate, the engine can audition
// the table's allocation at upload and keep a good one.
//   hipcc --offload-arch=gfx950 -O3 tools/alloc_audition.hip -o tools/alloc_audition;  alloc_audition <GiB per region> <regions> [lines, default 2^26]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; return x; }

// four adjacent lanes read the four 16-byte pieces of one random 64-byte line with one load instruction; four rounds serve the four lanes' lines
__global__ void __launch_bounds__(256) probe(const char* __restrict__ a, uint64_t n_lines, uint32_t* __restrict__ out, uint64_t salt) {
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint32_t sub = threadIdx.x & 3;
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint64_t owner = (tid & ~uint64_t(3)) + r;
        const uint64_t line = uint64_t((__uint128_t(mix(owner * 0x9E3779B97F4A7C15ULL + salt)) * n_lines) >> 64);
        const uint4 v = *reinterpret_cast<const uint4*>(a + line * 64 + 16 * sub);
        acc += v.x ^ v.w;
    }
    if (acc == 0x12345678u) out[tid & 1023] = acc;  // (keeps the loads alive; practically never taken)
}

int main(int argc, char** argv) {
    const uint64_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 36;
    const int regions = argc > 2 ? atoi(argv[2]) : 4;
    const uint64_t lanes = argc > 3 ? strtoull(argv[3], nullptr, 10) : (1ull << 26);
    const uint64_t bytes = gib << 30, n_lines = bytes / 64;
    uint32_t* out;
    CHECK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<char*> region;
    for (int i = 0; i < regions; ++i) {
        char* p = nullptr;
        CHECK(hipMalloc(&p, bytes));
        CHECK(hipMemset(p, 0, bytes));
        region.push_back(p);
    }
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < regions; ++i) {
            float best = 1e30f;
            for (int r = 0; r < 4; ++r) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(probe, dim3(uint32_t(lanes / 256)), dim3(256), 0, 0, region[i], n_lines, out, uint64_t(17 * pass + r + 1));
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0 && ms < best) best = ms;
            }
            printf("{\"pass\": %d, \"region\": %d, \"at\": \"%p\", \"GiB\": %llu, \"ms_best\": %.3f, \"Glines_per_s\": %.2f}\n", pass, i, (void*)region[i],
                   (unsigned long long)gib, best, double(lanes) / (best * 1e-3) / 1e9);
        }
    return 0;
}
