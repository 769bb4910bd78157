// gather_bench.hip -- what random small reads cost on MI355X: the practical roofline of the lookup path.
//
// Every lane walks a chain of DEPTH dependent random reads of WIDTH bytes (8, 16 or 32) through an array
// far larger than the 256 MiB Infinity Cache: exactly the access pattern of a k-mer lookup (pilot ->
// codeword -> strings), without any of its arithmetic. Prints one JSON line with the sustained rate in
// G reads/s and in GB/s of 64-byte sectors. Used by DESIGN.md to relate achieved Lookups/s to what the
// memory system can deliver for this granularity.
//
//   gather_bench <array MiB> <lanes> <depth> <width: 8|16|32|64, or 65 = 16 B + dependent 8 B in the same sector> [repeats]
//                [window MiB: the workgroups in flight at one time confine their reads to one random window of this
//                 size (0 = whole array): what sorting a batch by table region could buy]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));             \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    return x;
}

template <int WIDTH>
__global__ void __launch_bounds__(256) chase(const uint64_t* __restrict__ a, uint64_t n_units, int depth, uint64_t* __restrict__ out,
                                             uint64_t window_units) {
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    uint64_t base = 0;
    if (window_units) {  // 8192 consecutive workgroups (about four times what is resident) share a window
        base = uint64_t((__uint128_t(mix((blockIdx.x >> 13) + 77)) * (n_units - window_units)) >> 64);
        n_units = window_units;
    }
    uint64_t x = mix(tid * 0x9E3779B97F4A7C15ULL + 1);
    uint64_t acc = 0;
    for (int d = 0; d < depth; ++d) {
        const uint64_t unit = base + uint64_t((__uint128_t(x) * n_units) >> 64);
        if constexpr (WIDTH == 8) {
            acc += a[unit];
        } else if constexpr (WIDTH == 16) {
            const uint4 v = reinterpret_cast<const uint4*>(a)[unit];
            acc += v.x ^ v.w;
        } else if constexpr (WIDTH == 32) {
            const uint4 v0 = reinterpret_cast<const uint4*>(a)[2 * unit];
            const uint4 v1 = reinterpret_cast<const uint4*>(a)[2 * unit + 1];
            acc += v0.x ^ v1.w;
        } else if constexpr (WIDTH == 64) {  // a whole sector as four 16-byte loads
            const uint4* p = reinterpret_cast<const uint4*>(a) + 4 * unit;
            const uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
            acc += (v0.x ^ v1.w) + (v2.y ^ v3.z);
        } else {  // 65: 16 bytes, then a DEPENDENT 8-byte read elsewhere in the same sector
            const uint4* p = reinterpret_cast<const uint4*>(a) + 4 * unit;
            const uint4 v0 = p[0];
            const uint64_t* q = reinterpret_cast<const uint64_t*>(p) + 2 + (v0.x % 6);
            acc += v0.w ^ q[0];
        }
        x = mix(x + acc + d);  // next address depends on the loaded value
    }
    out[tid] = acc;
}

int main(int argc, char** argv) {
    const uint64_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4096;
    const uint64_t lanes = argc > 2 ? strtoull(argv[2], nullptr, 10) : (1ull << 24);
    const int depth = argc > 3 ? atoi(argv[3]) : 4;
    const int width = argc > 4 ? atoi(argv[4]) : 16;
    const int repeats = argc > 5 ? atoi(argv[5]) : 5;
    const uint64_t window_mib = argc > 6 ? strtoull(argv[6], nullptr, 10) : 0;
    const uint64_t bytes = mib << 20;
    uint64_t* a = nullptr;
    uint64_t* out = nullptr;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&out, lanes * 8));
    {
        std::vector<uint64_t> h(1 << 20);
        uint64_t s = 12345;
        for (auto& v : h) v = (s = s * 6364136223846793005ULL + 1442695040888963407ULL);
        for (uint64_t off = 0; off < bytes; off += h.size() * 8)
            CHECK(hipMemcpy(reinterpret_cast<char*>(a) + off, h.data(), std::min<uint64_t>(h.size() * 8, bytes - off), hipMemcpyHostToDevice));
    }
    const uint64_t n_units = bytes / uint64_t(width == 65 ? 64 : width);
    const uint64_t window_units = (window_mib << 20) / uint64_t(width == 65 ? 64 : width);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const dim3 grid(uint32_t((lanes + 255) / 256)), block(256);
    float best = 1e30f;
    for (int r = 0; r < repeats + 1; ++r) {
        CHECK(hipEventRecord(e0));
        if (width == 8) hipLaunchKernelGGL(chase<8>, grid, block, 0, 0, a, n_units, depth, out, window_units);
        else if (width == 16) hipLaunchKernelGGL(chase<16>, grid, block, 0, 0, a, n_units, depth, out, window_units);
        else if (width == 32) hipLaunchKernelGGL(chase<32>, grid, block, 0, 0, a, n_units, depth, out, window_units);
        else if (width == 64) hipLaunchKernelGGL(chase<64>, grid, block, 0, 0, a, n_units, depth, out, window_units);
        else hipLaunchKernelGGL(chase<65>, grid, block, 0, 0, a, n_units, depth, out, window_units);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    const double reads = double(lanes) * depth;
    printf("{\"array_MiB\": %llu, \"window_MiB\": %llu, \"lanes\": %llu, \"depth\": %d, \"width\": %d, \"ms\": %.3f, \"Greads_per_s\": %.2f, "
           "\"GBps_useful\": %.1f, \"GBps_64B_sectors\": %.1f}\n",
           (unsigned long long)mib, (unsigned long long)window_mib, (unsigned long long)lanes, depth, width, best, reads / best / 1e6,
           reads * width / best / 1e6, reads * 64 / best / 1e6);
    return 0;
}
