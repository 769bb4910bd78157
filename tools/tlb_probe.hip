// tlb_probe.hip -- is the large-array random-read ceiling (38 G reads/s beyond 16 GiB vs 48 G/s inside 2 GiB,
// profiles/r01/gather_bench_large_arrays.jsonl) address translation or DRAM?
//
// The same independent random reads as gather_bench (depth 1), but the array can be allocated in three ways:
//   malloc        one hipMalloc (what the replica did in round 1)
//   vmm           HIP virtual-memory API: ONE virtual range reserved with a chosen alignment, backed by physical
//                 chunks of a chosen size (hipMemCreate), mapped back to back. The driver can only use a page-table
//                 fragment as large as the physical contiguity AND the virtual alignment allow, so
//                 (chunk, alignment) = (2 MiB, 2 MiB) ... (1 GiB, 1 GiB) sweeps the fragment size upward.
//   pieces        the array as N separate hipMallocs of `chunk` MiB, addressed through a table (what a replica made
//                 of many small arrays looks like)
// Prints one JSON line per run. Counters for the same runs: rocprofv3 --pmc TCP_UTCL1_* (see tools/jobs).
//
//   tlb_probe <array MiB> <width 8|16|32|64> <malloc|vmm|pieces> [chunk MiB] [align MiB] [lanes] [repeats] [lane|coop]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e = (x);                                                                       \
        if (e != hipSuccess) {                                                                    \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__);     \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    return x;
}

template <int WIDTH>
__global__ void __launch_bounds__(256) gather(const char* __restrict__ a, uint64_t n_units, uint64_t* __restrict__ out, uint64_t salt) {
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t x = mix(tid * 0x9E3779B97F4A7C15ULL + salt);
    const uint64_t unit = uint64_t((__uint128_t(x) * n_units) >> 64);
    const char* p = a + unit * WIDTH;
    uint64_t acc;
    if constexpr (WIDTH == 8) {
        acc = *reinterpret_cast<const uint64_t*>(p);
    } else if constexpr (WIDTH == 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        acc = v.x ^ v.w;
    } else if constexpr (WIDTH == 32) {
        const uint4 v0 = reinterpret_cast<const uint4*>(p)[0], v1 = reinterpret_cast<const uint4*>(p)[1];
        acc = v0.x ^ v1.w;
    } else {
        const uint4* q = reinterpret_cast<const uint4*>(p);
        const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
        acc = (v0.x ^ v1.w) + (v2.y ^ v3.z);
    }
    out[tid] = acc;
}

/* COOP lanes share one unit: a unit of COOP*16 bytes is read by COOP adjacent lanes, 16 bytes each, in ONE load
   instruction (the memory pipeline sees one request and one translation per unit instead of COOP); COOP rounds
   serve the units of all COOP lanes. What a lane would then fetch from its neighbours by DPP is left out: the
   point here is the memory side. */
template <int COOP, bool NT = false>
__global__ void __launch_bounds__(256) gather_coop(const char* __restrict__ a, uint64_t n_units, uint64_t* __restrict__ out, uint64_t salt) {
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint32_t sub = threadIdx.x & (COOP - 1);
    uint64_t acc = 0;
#pragma unroll
    for (int r = 0; r < COOP; ++r) {
        const uint64_t owner = (tid & ~uint64_t(COOP - 1)) + r;  // the lane of the group whose unit is read in round r
        const uint64_t x = mix(owner * 0x9E3779B97F4A7C15ULL + salt);
        const uint64_t unit = uint64_t((__uint128_t(x) * n_units) >> 64);
        uint32_t x0, x3;
        if (NT) {  // nontemporal: the policy the lookup kernels use for their bucket lines
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a + unit * (16 * COOP) + 16 * sub));
            x0 = v.x;
            x3 = v.w;
        } else {
            const uint4 v = *reinterpret_cast<const uint4*>(a + unit * (16 * COOP) + 16 * sub);
            x0 = v.x;
            x3 = v.w;
        }
        acc += x0 ^ x3;
    }
    out[tid] = acc;
}

/* the array as separate allocations: unit -> (piece, offset) */
template <int WIDTH>
__global__ void __launch_bounds__(256) gather_pieces(const char* const* __restrict__ pieces, uint64_t units_per_piece, uint64_t n_units,
                                                     uint64_t* __restrict__ out, uint64_t salt) {
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t x = mix(tid * 0x9E3779B97F4A7C15ULL + salt);
    const uint64_t unit = uint64_t((__uint128_t(x) * n_units) >> 64);
    const char* p = pieces[unit / units_per_piece] + (unit % units_per_piece) * WIDTH;
    uint64_t acc;
    if constexpr (WIDTH == 32) {
        const uint4 v0 = reinterpret_cast<const uint4*>(p)[0], v1 = reinterpret_cast<const uint4*>(p)[1];
        acc = v0.x ^ v1.w;
    } else {
        acc = *reinterpret_cast<const uint64_t*>(p);
    }
    out[tid] = acc;
}

int main(int argc, char** argv) {
    const uint64_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4096;
    const int width = argc > 2 ? atoi(argv[2]) : 32;
    const char* mode = argc > 3 ? argv[3] : "malloc";
    const uint64_t chunk_mib = argc > 4 ? strtoull(argv[4], nullptr, 10) : 1024;
    const uint64_t align_mib = argc > 5 ? strtoull(argv[5], nullptr, 10) : chunk_mib;
    const uint64_t lanes = argc > 6 ? strtoull(argv[6], nullptr, 10) : (1ull << 27);
    const int repeats = argc > 7 ? atoi(argv[7]) : 5;
    const char* kernel = argc > 8 ? argv[8] : "lane";  // "coop": 2 (width 32) or 4 (width 64) lanes share a unit
    const uint64_t bytes = mib << 20;

    char* a = nullptr;
    uint64_t* out = nullptr;
    const char** d_pieces = nullptr;
    uint64_t va = 0;
    size_t gran_min = 0, gran_rec = 0;
    CHECK(hipSetDevice(0));
    CHECK(hipMalloc(&out, lanes * 8));
    if (!strcmp(mode, "malloc")) {
        CHECK(hipMalloc(&a, bytes));
    } else if (!strcmp(mode, "vmm")) {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        CHECK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
        CHECK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
        const uint64_t chunk = chunk_mib << 20;
        if (bytes % chunk) {
            fprintf(stderr, "array must be a multiple of the chunk\n");
            return 1;
        }
        void* base = nullptr;
        /* hipMemAddressReserve ignores the alignment argument here (every range comes back 32 MiB aligned): reserve
           `alignment` bytes more and map at the aligned address inside */
        const uint64_t align = align_mib << 20;
        CHECK(hipMemAddressReserve(&base, bytes + align, align, nullptr, 0));
        base = reinterpret_cast<void*>((reinterpret_cast<uint64_t>(base) + align - 1) / align * align);
        for (uint64_t off = 0; off < bytes; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            CHECK(hipMemCreate(&h, chunk, &prop, 0));
            CHECK(hipMemMap(static_cast<char*>(base) + off, chunk, 0, h, 0));
            CHECK(hipMemRelease(h));  // the mapping keeps it alive
        }
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CHECK(hipMemSetAccess(base, bytes, &acc, 1));
        a = static_cast<char*>(base);
    } else if (!strcmp(mode, "pieces")) {
        const uint64_t chunk = chunk_mib << 20;
        std::vector<const char*> h_pieces;
        for (uint64_t off = 0; off < bytes; off += chunk) {
            char* p = nullptr;
            CHECK(hipMalloc(&p, chunk));
            CHECK(hipMemset(p, 0x5a, chunk));
            h_pieces.push_back(p);
        }
        CHECK(hipMalloc(&d_pieces, h_pieces.size() * 8));
        CHECK(hipMemcpy(d_pieces, h_pieces.data(), h_pieces.size() * 8, hipMemcpyHostToDevice));
    } else {
        fprintf(stderr, "mode must be malloc, vmm or pieces\n");
        return 1;
    }
    va = reinterpret_cast<uint64_t>(a);
    if (a) CHECK(hipMemset(a, 0x5a, bytes));
    CHECK(hipDeviceSynchronize());

    const uint64_t n_units = bytes / uint64_t(width);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const dim3 grid(uint32_t((lanes + 255) / 256)), block(256);
    float best = 1e30f, sum = 0;
    for (int r = 0; r < repeats + 1; ++r) {
        CHECK(hipEventRecord(e0));
        if (d_pieces) {
            const uint64_t upp = (chunk_mib << 20) / uint64_t(width);
            if (width == 32) hipLaunchKernelGGL(gather_pieces<32>, grid, block, 0, 0, d_pieces, upp, n_units, out, uint64_t(r) + 1);
            else hipLaunchKernelGGL(gather_pieces<8>, grid, block, 0, 0, d_pieces, upp, n_units, out, uint64_t(r) + 1);
        } else if (!strcmp(kernel, "coop") && width == 32) {
            hipLaunchKernelGGL(gather_coop<2>, grid, block, 0, 0, a, n_units, out, uint64_t(r) + 1);
        } else if (!strcmp(kernel, "coop") && width == 64) {
            hipLaunchKernelGGL(gather_coop<4>, grid, block, 0, 0, a, n_units, out, uint64_t(r) + 1);
        } else if (!strcmp(kernel, "coopnt") && width == 64) {
            hipLaunchKernelGGL((gather_coop<4, true>), grid, block, 0, 0, a, n_units, out, uint64_t(r) + 1);
        } else if (!strcmp(kernel, "coopnt") && width == 128) {
            hipLaunchKernelGGL((gather_coop<8, true>), grid, block, 0, 0, a, n_units, out, uint64_t(r) + 1);
        } else if (!strcmp(kernel, "coop") && width == 128) {
            hipLaunchKernelGGL(gather_coop<8>, grid, block, 0, 0, a, n_units, out, uint64_t(r) + 1);
        } else if (!strcmp(kernel, "coop") && width == 256) {
            hipLaunchKernelGGL(gather_coop<16>, grid, block, 0, 0, a, n_units, out, uint64_t(r) + 1);
        } else if (width == 8) hipLaunchKernelGGL(gather<8>, grid, block, 0, 0, a, n_units, out, uint64_t(r) + 1);
        else if (width == 16) hipLaunchKernelGGL(gather<16>, grid, block, 0, 0, a, n_units, out, uint64_t(r) + 1);
        else if (width == 32) hipLaunchKernelGGL(gather<32>, grid, block, 0, 0, a, n_units, out, uint64_t(r) + 1);
        else hipLaunchKernelGGL(gather<64>, grid, block, 0, 0, a, n_units, out, uint64_t(r) + 1);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) {
            sum += ms;
            if (ms < best) best = ms;
        }
    }
    printf("{\"array_MiB\": %llu, \"width\": %d, \"mode\": \"%s\", \"chunk_MiB\": %llu, \"align_MiB\": %llu, \"va\": \"0x%llx\", "
           "\"va_align_log2\": %d, \"gran_min\": %zu, \"gran_rec\": %zu, \"lanes\": %llu, \"ms_best\": %.3f, \"ms_avg\": %.3f, "
           "\"Greads_per_s\": %.2f, \"GBps_useful\": %.1f, \"kernel\": \"%s\"}\n",
           (unsigned long long)mib, width, mode, (unsigned long long)chunk_mib, (unsigned long long)align_mib, (unsigned long long)va,
           va ? __builtin_ctzll(va) : 0, gran_min, gran_rec, (unsigned long long)lanes, best, sum / repeats, double(lanes) / best / 1e6,
           double(lanes) * width / best / 1e6, kernel);
    return 0;
}
