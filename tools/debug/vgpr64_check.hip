// vgpr64_check.hip -- round 4: does a wave whose VGPR allocation is exactly 64 compute correctly in v58..v63 under sparse EXEC?
//
// Background: the is_member instance of fast_lookup_kernel allocates exactly 64 VGPRs and keeps the
// temporaries of the slot comparison of sk_finish_in_wave in v56..v63; it reported ~0.18 % of the indexed k-mers absent,
// differently from launch to launch. The SAME machine code is correct (a) with .amdhsa_next_free_vgpr 72 in the kernel
// descriptor and (b) with v[56:63] and v[8:15] renamed into each other. This program runs the comparison's instruction sequence
// (ds_read_b128 into v[58:61], shift amounts in v62 / v63, 64-bit funnel shift) with nothing of the repository around it:
//   hipcc --offload-arch=gfx950 -O3 tools/debug/vgpr64_check.hip -o tools/debug/vgpr64_check && tools/debug/vgpr64_check
// Modes: where the data and the two shift amounts live, and how many registers the wave is given (table `modes` below).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__host__ __device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__host__ __device__ __forceinline__ uint32_t word_of(uint64_t line, uint32_t j) { return uint32_t(mix(line * 16 + j + 0x1234567)); }

__global__ void fill(uint4* t, uint64_t n_lines) {
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_lines * 4; i += uint64_t(gridDim.x) * blockDim.x) {
        const uint64_t line = i >> 2;
        const uint32_t sub = uint32_t(i & 3);
        t[i] = make_uint4(word_of(line, 4 * sub), word_of(line, 4 * sub + 1), word_of(line, 4 * sub + 2), word_of(line, 4 * sub + 3));
    }
}

/* the comparison's instruction sequence with the registers spelled out: data v[D0:D3] (from LDS), shift amounts vSH = 2a, vINV = 63 - 2a */
#define SEQ(D0, D1, D2, D3, SH, INV)                                             \
    "ds_read_b128 v[" #D0 ":" #D3 "], %[addr] offset:48\n\t"                      \
    "v_lshlrev_b32_e32 v" #SH ", 1, %[a]\n\t"                                     \
    "v_sub_u32_e32 v" #INV ", 63, v" #SH "\n\t"                                   \
    "s_waitcnt lgkmcnt(0)\n\t"                                                    \
    "v_lshlrev_b64 v[" #D2 ":" #D3 "], 1, v[" #D2 ":" #D3 "]\n\t"                 \
    "v_lshlrev_b64 v[" #D2 ":" #D3 "], v" #INV ", v[" #D2 ":" #D3 "]\n\t"         \
    "v_lshrrev_b64 v[" #D0 ":" #D1 "], v" #SH ", v[" #D0 ":" #D1 "]\n\t"          \
    "v_or_b32_e32 %[hi], v" #D3 ", v" #D1 "\n\t"                                  \
    "v_or_b32_e32 %[lo], v" #D2 ", v" #D0 "\n\t"
#define RUN(TEXT, ...) asm volatile(TEXT : [lo] "=&v"(lo), [hi] "=&v"(hi) : [addr] "v"(addr), [a] "v"(a) : __VA_ARGS__, "memory")

struct probe_mode {
    int allocation;
    const char* what;
};
constexpr int MODES = 19;
static const probe_mode modes[MODES] = {
    {64, "data v[58:61], 2a in v62, 63-2a in v63 (the kernel's own allocation): v_lshlrev_b64 shifts by v63"},
    {64, "data v[58:61], 2a in v63, 63-2a in v62: v_lshrrev_b64 shifts by v63"},
    {64, "data v[60:63], shift amounts in v58 / v59: the 64-bit VALUE in v[62:63], amounts elsewhere"},
    {64, "data v[56:59], shift amounts in v60 / v61; v62, v63 allocated and untouched"},
    {64, "data v[56:59], shift amounts in v61 / v62; v63 written (v_mov) and not otherwise used"},
    {72, "data v[66:69], shift amounts in v70 / v71: the last register of a 72-register allocation"},
    {72, "data v[58:61], shift amounts in v62 / v63, allocation 72 (v71 clobbered): the same instructions as mode 0"},
    {64, "32-bit only: v_lshlrev_b32 by v63, v_lshrrev_b32 by v62"},
    {56, "data v[50:53], shift amounts in v54 / v55: the last register of a 56-register allocation"},
    {64, "as mode 0 with s_nop 4 in front of each 64-bit shift"},
    {64, "as mode 0, the shift amount copied out of v63 (v_mov_b32) into the output register first, then shifted by that"},
    {64, "v_mad_u64_u32 with src0 = v63 (32-bit factor in the last register)"},
    {64, "v_mad_u64_u32 with src1 = v63"},
    {64, "v_ashrrev_i64 by v63"},
    {64, "v_lshl_add_u64 with the shift amount (src1) in v63"},
    {64, "v_cvt_f64_u32 of v63"},
    {64, "v_mul_hi_u32 with src0 = v63 (32-bit control)"},
    {64, "ds_read_b32 with the LDS address in v63"},
    {64, "v_mad_u64_u32 with the 64-bit addend (src2) in v[62:63]"},
};

template <int MODE>
__global__ void __launch_bounds__(256) probe(const uint4* __restrict__ table, uint64_t n_lines, int rounds, uint32_t density,
                                             unsigned long long* bad, unsigned long long* tried) {
    __shared__ uint4 lds[1024 + 1];
    uint4* wave_stage = lds + (threadIdx.x >> 6) * 256;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t wave_id = tid >> 6;
    uint32_t wrong = 0, did = 0;
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        /* quad q of the wave fetches line L(q): lane = piece, as the in-wave loop does */
        const uint64_t line = uint64_t((__uint128_t(mix(wave_id * 16 + (lane >> 2) + uint64_t(r) * 0x9E3779B97F4A7C15ULL)) * n_lines) >> 64);
        wave_stage[lane] = table[line * 4 + (lane & 3u)];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const uint64_t h = mix(tid * 0x632BE59BD9B4E019ULL + r);
        const bool served = (h & 63u) < density;
        if (served) {
            const uint32_t q = uint32_t(h >> 8) & 15u;       // whose line this lane examines
            const uint32_t a = uint32_t(h >> 16) % 32u;      // alignment: shift by 2a bits
            const uint64_t want_line = uint64_t((__uint128_t(mix(wave_id * 16 + q + uint64_t(r) * 0x9E3779B97F4A7C15ULL)) * n_lines) >> 64);
            const uint64_t w0 = uint64_t(word_of(want_line, 12)) | (uint64_t(word_of(want_line, 13)) << 32);
            const uint64_t w1 = uint64_t(word_of(want_line, 14)) | (uint64_t(word_of(want_line, 15)) << 32);
            uint64_t want = (w0 >> (2 * a)) | ((w1 << 1) << (63 - 2 * a));
            const uint32_t addr = uint32_t(reinterpret_cast<uintptr_t>(wave_stage + 4 * q));  // LDS byte address of the line
            uint32_t lo, hi;
            if constexpr (MODE == 0) RUN(SEQ(58, 59, 60, 61, 62, 63), "v58", "v59", "v60", "v61", "v62", "v63");
            if constexpr (MODE == 1) RUN(SEQ(58, 59, 60, 61, 63, 62), "v58", "v59", "v60", "v61", "v62", "v63");
            if constexpr (MODE == 2) RUN(SEQ(60, 61, 62, 63, 58, 59), "v58", "v59", "v60", "v61", "v62", "v63");
            if constexpr (MODE == 3) RUN(SEQ(56, 57, 58, 59, 60, 61), "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
            if constexpr (MODE == 4) RUN("v_mov_b32_e32 v63, 0\n\t" SEQ(56, 57, 58, 59, 61, 62), "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
            if constexpr (MODE == 5) RUN(SEQ(66, 67, 68, 69, 70, 71), "v66", "v67", "v68", "v69", "v70", "v71");
            if constexpr (MODE == 6) RUN(SEQ(58, 59, 60, 61, 62, 63), "v58", "v59", "v60", "v61", "v62", "v63", "v71");
            if constexpr (MODE == 7) {
                RUN("ds_read_b128 v[58:61], %[addr] offset:48\n\t"
                    "v_and_b32_e32 v62, 31, %[a]\n\t"
                    "v_sub_u32_e32 v63, 31, v62\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "v_lshlrev_b32_e32 %[hi], v63, v60\n\t"
                    "v_lshrrev_b32_e32 %[lo], v62, v58\n\t",
                    "v58", "v59", "v60", "v61", "v62", "v63");
                want = uint64_t(uint32_t(w0) >> (a & 31u)) | (uint64_t(uint32_t(w1) << (31 - (a & 31u))) << 32);
            }
            if constexpr (MODE == 8) RUN(SEQ(50, 51, 52, 53, 54, 55), "v50", "v51", "v52", "v53", "v54", "v55");
            if constexpr (MODE == 9)
                RUN("ds_read_b128 v[58:61], %[addr] offset:48\n\t"
                    "v_lshlrev_b32_e32 v62, 1, %[a]\n\t"
                    "v_sub_u32_e32 v63, 63, v62\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "s_nop 4\n\t"
                    "v_lshlrev_b64 v[60:61], 1, v[60:61]\n\t"
                    "s_nop 4\n\t"
                    "v_lshlrev_b64 v[60:61], v63, v[60:61]\n\t"
                    "s_nop 4\n\t"
                    "v_lshrrev_b64 v[58:59], v62, v[58:59]\n\t"
                    "s_nop 4\n\t"
                    "v_or_b32_e32 %[hi], v61, v59\n\t"
                    "v_or_b32_e32 %[lo], v60, v58\n\t",
                    "v58", "v59", "v60", "v61", "v62", "v63");
            if constexpr (MODE == 10)
                RUN("ds_read_b128 v[58:61], %[addr] offset:48\n\t"
                    "v_lshlrev_b32_e32 v62, 1, %[a]\n\t"
                    "v_sub_u32_e32 v63, 63, v62\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "v_mov_b32_e32 %[lo], v63\n\t"
                    "v_lshlrev_b64 v[60:61], 1, v[60:61]\n\t"
                    "v_lshlrev_b64 v[60:61], %[lo], v[60:61]\n\t"
                    "v_lshrrev_b64 v[58:59], v62, v[58:59]\n\t"
                    "v_or_b32_e32 %[hi], v61, v59\n\t"
                    "v_or_b32_e32 %[lo], v60, v58\n\t",
                    "v58", "v59", "v60", "v61", "v62", "v63");
            /* other instructions with a 32-bit operand in the last register: f = low word of w0, g = low word of w1 */
            const uint32_t f = uint32_t(w0), g = uint32_t(w1);
            if constexpr (MODE == 11 || MODE == 12) {
                if constexpr (MODE == 11)
                    RUN("ds_read_b128 v[58:61], %[addr] offset:48\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v63, v58\n\tv_mov_b32_e32 v62, v60\n\t"
                        "v_mad_u64_u32 v[58:59], vcc, v63, v62, 0\n\tv_mov_b32_e32 %[lo], v58\n\tv_mov_b32_e32 %[hi], v59\n\t",
                        "v58", "v59", "v60", "v61", "v62", "v63", "vcc");
                else
                    RUN("ds_read_b128 v[58:61], %[addr] offset:48\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v63, v58\n\tv_mov_b32_e32 v62, v60\n\t"
                        "v_mad_u64_u32 v[58:59], vcc, v62, v63, 0\n\tv_mov_b32_e32 %[lo], v58\n\tv_mov_b32_e32 %[hi], v59\n\t",
                        "v58", "v59", "v60", "v61", "v62", "v63", "vcc");
                want = uint64_t(f) * uint64_t(g);
            }
            if constexpr (MODE == 13) {
                RUN("ds_read_b128 v[58:61], %[addr] offset:48\n\tv_lshlrev_b32_e32 v63, 1, %[a]\n\ts_waitcnt lgkmcnt(0)\n\t"
                    "v_ashrrev_i64 v[58:59], v63, v[58:59]\n\tv_mov_b32_e32 %[lo], v58\n\tv_mov_b32_e32 %[hi], v59\n\t",
                    "v58", "v59", "v60", "v61", "v62", "v63");
                want = uint64_t(int64_t(w0) >> (2 * a));
            }
            if constexpr (MODE == 14) {
                RUN("ds_read_b128 v[58:61], %[addr] offset:48\n\tv_and_b32_e32 v63, 3, %[a]\n\ts_waitcnt lgkmcnt(0)\n\t"
                    "v_lshl_add_u64 v[58:59], v[58:59], v63, v[60:61]\n\tv_mov_b32_e32 %[lo], v58\n\tv_mov_b32_e32 %[hi], v59\n\t",
                    "v58", "v59", "v60", "v61", "v62", "v63");
                want = (w0 << (a & 3u)) + w1;
            }
            if constexpr (MODE == 15) {
                RUN("ds_read_b128 v[58:61], %[addr] offset:48\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v63, v58\n\t"
                    "v_cvt_f64_u32_e32 v[60:61], v63\n\tv_mov_b32_e32 %[lo], v60\n\tv_mov_b32_e32 %[hi], v61\n\t",
                    "v58", "v59", "v60", "v61", "v62", "v63");
                const double dv = double(f);
                want = __double_as_longlong(dv);
            }
            if constexpr (MODE == 16) {
                RUN("ds_read_b128 v[58:61], %[addr] offset:48\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v63, v58\n\t"
                    "v_mul_hi_u32 %[hi], v63, v60\n\tv_mul_lo_u32 %[lo], v63, v60\n\t",
                    "v58", "v59", "v60", "v61", "v62", "v63");
                want = uint64_t(f) * uint64_t(g);
            }
            if constexpr (MODE == 17) {
                RUN("v_add_u32_e32 v63, 48, %[addr]\n\tds_read_b32 %[lo], v63\n\tds_read_b32 %[hi], v63 offset:4\n\ts_waitcnt lgkmcnt(0)\n\t",
                    "v58", "v59", "v60", "v61", "v62", "v63");
                want = w0;
            }
            if constexpr (MODE == 18) {
                RUN("ds_read_b128 v[58:61], %[addr] offset:48\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v62, v60\n\tv_mov_b32_e32 v63, v61\n\t"
                    "v_mad_u64_u32 v[60:61], vcc, v58, v59, v[62:63]\n\tv_mov_b32_e32 %[lo], v60\n\tv_mov_b32_e32 %[hi], v61\n\t",
                    "v58", "v59", "v60", "v61", "v62", "v63", "vcc");
                want = uint64_t(f) * uint64_t(uint32_t(w0 >> 32)) + w1;
            }
            const uint64_t got = uint64_t(lo) | (uint64_t(hi) << 32);
            wrong += got != want;
            ++did;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    if (wrong) atomicAdd(bad, (unsigned long long)wrong);
    if (did) atomicAdd(tried, (unsigned long long)did);
}

template <int MODE>
static int run_mode(const uint4* table, uint64_t n_lines, int rounds, unsigned long long* counters) {
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&probe<MODE>)));
    for (uint32_t density : {2u, 64u}) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemset(counters, 0, 16));
            probe<MODE><<<1 << 16, 256>>>(table, n_lines, rounds, density, counters, counters + 1);
            CHECK(hipDeviceSynchronize());
            unsigned long long h[2];
            CHECK(hipMemcpy(h, counters, 16, hipMemcpyDeviceToHost));
            printf("{\"mode\": %d, \"what\": \"%s\", \"numRegs\": %d, \"lanes_served_of_64\": %u, \"rep\": %d, \"comparisons\": %llu, \"wrong\": %llu}\n", MODE,
                   modes[MODE].what, fa.numRegs, density, rep, h[1], h[0]);
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    const uint64_t n_lines = argc > 1 ? strtoull(argv[1], nullptr, 10) : (uint64_t(1) << 26);  // 4 GiB of 64-byte lines
    const int rounds = argc > 2 ? atoi(argv[2]) : 64;
    uint4* table;
    unsigned long long* counters;
    CHECK(hipMalloc(&table, n_lines * 64));
    CHECK(hipMalloc(&counters, 16));
    fill<<<4096, 256>>>(table, n_lines);
    CHECK(hipDeviceSynchronize());
    const bool only_new = argc > 3;  // (the shift modes 0-10 are in profiles/r04/vgpr64_check_shift_modes.jsonl)
    if (only_new) {
        if (run_mode<11>(table, n_lines, rounds, counters) || run_mode<12>(table, n_lines, rounds, counters) || run_mode<13>(table, n_lines, rounds, counters) ||
            run_mode<14>(table, n_lines, rounds, counters) || run_mode<15>(table, n_lines, rounds, counters) || run_mode<16>(table, n_lines, rounds, counters) ||
            run_mode<17>(table, n_lines, rounds, counters) || run_mode<18>(table, n_lines, rounds, counters))
            return 1;
        return 0;
    }
    if (run_mode<0>(table, n_lines, rounds, counters) || run_mode<1>(table, n_lines, rounds, counters) || run_mode<2>(table, n_lines, rounds, counters) ||
        run_mode<3>(table, n_lines, rounds, counters) || run_mode<4>(table, n_lines, rounds, counters) || run_mode<5>(table, n_lines, rounds, counters) ||
        run_mode<6>(table, n_lines, rounds, counters) || run_mode<7>(table, n_lines, rounds, counters) || run_mode<8>(table, n_lines, rounds, counters) ||
        run_mode<9>(table, n_lines, rounds, counters) || run_mode<10>(table, n_lines, rounds, counters) || run_mode<11>(table, n_lines, rounds, counters) ||
        run_mode<12>(table, n_lines, rounds, counters) || run_mode<13>(table, n_lines, rounds, counters) || run_mode<14>(table, n_lines, rounds, counters) ||
        run_mode<15>(table, n_lines, rounds, counters) || run_mode<16>(table, n_lines, rounds, counters) || run_mode<17>(table, n_lines, rounds, counters) ||
        run_mode<18>(table, n_lines, rounds, counters))
        return 1;
    return 0;
}
