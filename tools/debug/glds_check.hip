// glds_check.hip -- does the quad-cooperative LDS-DMA line fetch (lookup_device.hpp: sk_stage_buckets) always deliver the
// line before the DS reads that follow the compiler's s_waitcnt vmcnt(0)? Every lane fetches a random 64-byte line whose
// content is a function of its index, reads it back from LDS (mode 0: four ds_read_b128; mode 1: b32 + b64 + b128 as the
// is_member instance does) and compares. Prints mismatches per variant.
//   glds_check <array MiB> <lanes> <rounds per lane> <need percent>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* global_ptr;

__host__ __device__ inline uint32_t word_of(uint64_t line, uint32_t w) {
    uint64_t x = line * 16 + w + 0x9E3779B97F4A7C15ULL;
    x ^= x >> 31; x *= 0xD6E8FEB86659FD93ULL; x ^= x >> 29;
    return uint32_t(x);
}
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; return x; }

__global__ void fill(uint32_t* a, uint64_t n_words) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n_words) a[i] = word_of(i / 16, uint32_t(i % 16));
}

template <int OWNER>
__device__ __forceinline__ uint32_t quad_broadcast(uint32_t v) { return uint32_t(__builtin_amdgcn_mov_dpp(int(v), OWNER * 0x55, 0xf, 0xf, true)); }

template <int P>
__device__ __forceinline__ void stage_round(char const* a, uint32_t line, uint32_t need, uint32_t sub, uint4* ws) {
    const uint32_t ob = quad_broadcast<P>(line), on = quad_broadcast<P>(need);
    if (on) __builtin_amdgcn_global_load_lds((global_ptr)(a + uint64_t(ob) * 64 + 16 * sub), (lds_ptr)(ws + P * 64), 16, 0, 0);
}

template <int MODE, int FIX>
__global__ void __launch_bounds__(256) check(const char* __restrict__ a, uint32_t n_lines, int rounds, uint32_t need_percent,
                                             unsigned long long* __restrict__ bad, uint64_t salt, uint32_t share) {
    __shared__ uint4 lds[256 * 4];
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, sub = lane & 3u;
    uint4* ws = lds + (threadIdx.x >> 6) * 256;
    const uint4* mine = ws + sub * 64 + (lane >> 2) * 4;
    uint32_t wrong = 0;
    for (int r = 0; r < rounds; ++r) {
        const uint64_t x = mix((tid / share) * 0x9E3779B97F4A7C15ULL + salt + r);  // `share` consecutive lanes fetch the same line
        const uint32_t line = uint32_t((__uint128_t(x) * n_lines) >> 64);
        const bool need = (mix(x + 17) % 100) < need_percent;
        const uint32_t n32 = need ? 1u : 0u, b = need ? line : 0u;
        stage_round<0>(a, b, n32, sub, ws);
        stage_round<1>(a, b, n32, sub, ws);
        stage_round<2>(a, b, n32, sub, ws);
        stage_round<3>(a, b, n32, sub, ws);
        __builtin_amdgcn_wave_barrier();
        if constexpr (FIX == 1) { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_s_sleep(1); }
        if constexpr (FIX == 2) __syncthreads();
        if constexpr (FIX == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (need) {
            if constexpr (MODE == 0) {
                const uint4 q0 = mine[0], q1 = mine[1], q2 = mine[2], q3 = mine[3];
                wrong += q0.x != word_of(line, 0) || q0.w != word_of(line, 3) || q1.y != word_of(line, 5) || q2.z != word_of(line, 10) || q3.w != word_of(line, 15);
            } else {
                const uint32_t* w = reinterpret_cast<const uint32_t*>(mine);
                const uint32_t m0 = w[0];
                const uint2 zw = *reinterpret_cast<const uint2*>(w + 2);
                const uint4 q1 = mine[1];
                bool ok = m0 == word_of(line, 0) && zw.x == word_of(line, 2) && zw.y == word_of(line, 3) && q1.x == word_of(line, 4) && q1.w == word_of(line, 7);
                if (ok && (m0 & 3) == 1) {  // data-dependent second half, as the table probe does
                    const uint32_t m1 = w[8];
                    const uint4 q3 = mine[3];
                    ok = m1 == word_of(line, 8) && q3.y == word_of(line, 13);
                }
                wrong += !ok;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr (FIX == 2) __syncthreads();
    }
    if (wrong) atomicAdd(bad, (unsigned long long)wrong);
}

int main(int argc, char** argv) {
    const uint64_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4096;
    const uint64_t lanes = argc > 2 ? strtoull(argv[2], nullptr, 10) : (1ull << 26);
    const int rounds = argc > 3 ? atoi(argv[3]) : 4;
    const uint32_t need_percent = argc > 4 ? atoi(argv[4]) : 90;
    const uint32_t share = argc > 5 ? atoi(argv[5]) : 1;
    const uint64_t bytes = mib << 20, n_lines = bytes / 64;
    char* a = nullptr;
    unsigned long long* bad = nullptr;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&bad, 8));
    hipLaunchKernelGGL(fill, dim3(uint32_t((bytes / 4 + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(a), bytes / 4);
    CHECK(hipDeviceSynchronize());
    const dim3 grid(uint32_t((lanes + 255) / 256)), block(256);
    auto run = [&](const char* name, auto kernel) -> int {
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(bad, 0, 8));
            hipLaunchKernelGGL(kernel, grid, block, 0, 0, a, uint32_t(n_lines), rounds, need_percent, bad, uint64_t(rep) * 1000003 + 1, share);
            unsigned long long h = 0;
            CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
            printf("{\"variant\": \"%s\", \"rep\": %d, \"checked\": %llu, \"mismatches\": %llu}\n", name, rep,
                   (unsigned long long)(lanes * rounds * need_percent / 100), h);
        }
        return 0;
    };
    if (run("b128_reads", check<0, 0>)) return 1;
    if (run("b32_b64_b128_reads", check<1, 0>)) return 1;
    if (run("b32_b64_b128_reads+waitcnt0_sleep", check<1, 1>)) return 1;
    if (run("b32_b64_b128_reads+syncthreads", check<1, 2>)) return 1;
    if (run("b32_b64_b128_reads+asm_vmcnt0", check<1, 3>)) return 1;
    if (run("b128_reads+asm_vmcnt0", check<0, 3>)) return 1;
    return 0;
}
