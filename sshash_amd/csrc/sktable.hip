// sktable.hip -- construction of the super-k-mer table (device_layout.hpp (5)) on the GPU, at upload.
//
// Input: the atoms of (1) and the endpoints, already resident. Nothing of the host index's own
// minimizer structures is used: the table is keyed by strand-symmetric minimizers (sk_key) recomputed from
// the strings, so it serves regular and canonical dictionaries alike.
//
//   1. scan     one lane per k-mer start: the k-mer's key occurrence (key, position, strand); a lane
//               whose occurrence differs from its left neighbour's starts a super-k-mer and emits one
//               tuple. Two passes (count, exclusive scan, emit) keep the tuples in string order.
//   2. sort     stable radix sort of the tuples by key (hipCUB), run-length encode -> one run per key.
//   3. place    SK_CHOICES rounds, one per hashed bucket choice: a key claims the first free slot of the bucket with a
//               CAS on the slot's flag word; a key that finds the bucket full sets the bucket's "go on" flag and waits
//               for the next round. The first round runs in three turns, the items of long super-k-mers first: the 7 %
//               of items that cannot stay in their first bucket are then the ones the fewest queries ask for.
//   4. fill     the winner writes its slot: the 64 bases around one occurrence, how far the super-k-mer may extend
//               inside its string, the string id. A key with up to SK_INLINE_MAX occurrences takes one such slot per
//               occurrence (a lookup then never leaves the table); a key with more takes one slot pointing at its
//               marker, and every k-mer of its super-k-mers becomes an item of its own (sk_heavy_kmers_kernel).
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <cstdlib>

#include "replica.hpp"
#include "hooks.hpp"
#include "sktable.hpp"

namespace sshash_amd {

namespace {

constexpr uint32_t WAVE = 64;
/* a tuple's value: position of the key occurrence << 1 | strand (40 bits), and -- bits 56-61 -- how many consecutive k-mers
   elected this occurrence, as far as the scanning wave saw (a super-k-mer running into the next wave is cut short: it is a
   priority, not a fact anything relies on) */
constexpr uint32_t SK_LEN_SHIFT = 56;
constexpr uint64_t SK_VAL_MASK = (uint64_t(1) << 40) - 1;
constexpr uint32_t NEW_PER_WAVE = WAVE - 1;  // lane 0 only supplies its right neighbour's "previous"

template <int W, bool EMIT>
__global__ void __launch_bounds__(256)
sk_scan_kernel(const dict_view d, const uint64_t first_wave, const uint64_t num_waves, uint32_t* __restrict__ counts,
               const uint64_t* __restrict__ offsets, uint64_t* __restrict__ keys, uint64_t* __restrict__ vals) {
    const uint64_t gtid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t wave = first_wave + gtid / WAVE;  // (launched in pieces: a launch of 2^32 threads is not carried out)
    const uint32_t lane = uint32_t(gtid % WAVE);
    if (wave >= num_waves) return;  // whole waves only: the grid is sized in waves
    const int64_t i = int64_t(wave * NEW_PER_WAVE + lane) - 1;
    uint64_t key = INVALID_U64, val = INVALID_U64;
    if (i >= 0 && uint64_t(i) + d.k <= d.num_bases) {
        const window_t<W> w = read_window<W>(d.granules, uint64_t(i), d.k);
        if (!w.crosses) {
            const kmer_w<W> x = w.kmer;
            const kmer_w<W> x_rc = kmer_revcomp<W>(x, d.k);
            const sk_key_t kk = sk_key<W>(x, x_rc, d.k, d.sk.m);
            if (sk_usable(d, kk)) {  // (another table shard's key is left out like a tie)
                key = kk.key;
                const uint64_t p = uint64_t(i) + (kk.rc ? (d.k - d.sk.m) - kk.pos : kk.pos);
                val = (p << 1) | (kk.rc ? 1u : 0u);
            }
        }
    }
    const uint64_t prev_key = __shfl_up(key, 1), prev_val = __shfl_up(val, 1);
    const bool start = lane > 0 && key != INVALID_U64 && (key != prev_key || val != prev_val);
    const uint64_t ballot = __ballot(start);
    if constexpr (!EMIT) {
        if (lane == 0) counts[wave] = uint32_t(__popcll(ballot));
    } else {
        if (start) {
            const uint64_t at = offsets[wave] + uint64_t(__popcll(ballot & ((uint64_t(1) << lane) - 1)));
            const uint64_t later = lane < 63 ? ballot >> (lane + 1) : 0;
            const uint32_t run = later ? uint32_t(__ffsll((unsigned long long)later)) : WAVE - lane;  // k-mers up to the next start
            const uint32_t most = d.k - d.sk.m + 1;
            keys[at] = key;
            vals[at] = val | (uint64_t(run < most ? run : most) << SK_LEN_SHIFT);
        }
    }
}

/* 64*W bases starting at base `pos` (pos may be negative: the missing bases read as zero) */
template <int W>
__device__ __forceinline__ void read_bases(void const* __restrict__ blocks, int64_t pos, uint64_t (&out)[2 * W]) {
    const uint64_t from = pos < 0 ? 0 : uint64_t(pos);
    const uint32_t r = uint32_t(from) & 31u;
    uint64_t b[2 * W + 1];  // 32 bases each
    if constexpr (W == 1) {
        const uint4* A = reinterpret_cast<const uint4*>(blocks) + 2 * (from >> 5);
        const uint4 a0 = A[0], a1 = A[2];  // bases of atom `from/32` and of the next atom
        b[0] = uint64_t(a0.x) | (uint64_t(a0.y) << 32);
        b[1] = uint64_t(a0.z) | (uint64_t(a0.w) << 32);
        b[2] = uint64_t(a1.z) | (uint64_t(a1.w) << 32);
    } else {
        const uint4* G = reinterpret_cast<const uint4*>(blocks) + (from >> 5);
        for (int i = 0; i < 2 * W + 1; ++i) {
            const uint4 g = G[i];
            b[i] = uint64_t(g.z) | (uint64_t(g.w) << 32);
        }
    }
    for (int i = 0; i < 2 * W; ++i) out[i] = funnel_shr(b[i], b[i + 1], 2 * r);
    if (pos < 0) {  // shift the whole window up by -pos bases (1 .. k-m <= 62)
        const uint32_t bits = 2 * uint32_t(-pos);
        const int ws = int(bits >> 6);
        const uint32_t bs = bits & 63u;
        uint64_t t[2 * W];
        for (int i = 0; i < 2 * W; ++i) {
            const uint64_t lo = i - ws - 1 >= 0 ? out[i - ws - 1] : 0, hi = i - ws >= 0 ? out[i - ws] : 0;
            t[i] = (hi << bs) | (bs ? lo >> (64 - bs) : 0);
        }
        for (int i = 0; i < 2 * W; ++i) out[i] = t[i];
    }
}

/* ---- items: what is placed into the table, one slot each ----
   flags[t] of tuple t (sorted by key): 0 = an occurrence of a light key (<= SK_INLINE_MAX occurrences): its own inline
   slot; 1 = first occurrence of a heavy key: the key's marker slot; 2 = further occurrence of a heavy key: no slot
   under the key. The k-mers of the heavy keys' occurrences are items of their own (keys = sk_kmer_key). */
constexpr uint8_t SK_ITEM_INLINE = 0, SK_ITEM_MARKER = 1, SK_ITEM_NONE = 2;

/* bin of a key with `size` occurrences: 1, 2, 3, 4, 5-8, 9-16, 17-64, 65-1024, > 1024 */
constexpr uint32_t SK_HIST_BINS = 9;
__device__ __forceinline__ uint32_t sk_hist_bin(uint32_t size) {
    return size <= 4 ? size - 1 : size <= 8 ? 4 : size <= 16 ? 5 : size <= 64 ? 6 : size <= 1024 ? 7 : 8;
}

/* one lane per run: classify its tuples; histogram of the keys by number of occurrences (LDS per workgroup, then one
   atomic per bin and workgroup) */
__global__ void __launch_bounds__(256)
sk_classify_kernel(const uint64_t num_keys, const uint32_t* __restrict__ run_sizes, const uint32_t* __restrict__ run_begins,
                   uint8_t* __restrict__ flags, unsigned long long* __restrict__ stats, unsigned long long* __restrict__ histogram) {
    __shared__ unsigned long long local[2 * SK_HIST_BINS];
    if (threadIdx.x < 2 * SK_HIST_BINS) local[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t r = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    bool heavy = false;
    uint32_t size = 0;
    if (r < num_keys) {
        const uint32_t s = run_sizes[r];
        atomicAdd(&local[sk_hist_bin(s)], 1ull);
        atomicAdd(&local[SK_HIST_BINS + sk_hist_bin(s)], (unsigned long long)s);
    }
    __syncthreads();
    if (threadIdx.x < 2 * SK_HIST_BINS && local[threadIdx.x]) atomicAdd(histogram + threadIdx.x, local[threadIdx.x]);
    if (r < num_keys) {
        size = run_sizes[r];
        heavy = size > SK_INLINE_MAX;
        if (heavy) {
            const uint64_t begin = run_begins[r];
            flags[begin] = SK_ITEM_MARKER;
            for (uint32_t t = 1; t < size; ++t) flags[begin + t] = SK_ITEM_NONE;
        }
    }
    const uint64_t b = __ballot(heavy);
    uint64_t occs = heavy ? size : 0;
    for (int o = 32; o > 0; o >>= 1) occs += __shfl_down(occs, o, 64);
    if ((threadIdx.x & (WAVE - 1)) == 0 && b) {
        atomicAdd(stats + 4, (unsigned long long)__popcll(b));  // heavy keys
        atomicAdd(stats + 5, (unsigned long long)occs);         // their occurrences
    }
}

/* One lane per (occurrence of a heavy key, alignment a): the k-mer starting km - a bases before the occurrence, if it
   lies inside its string and elects this very occurrence as its key, becomes an item keyed by sk_kmer_key. EMIT off:
   count only. */
template <int W, bool EMIT>
__global__ void __launch_bounds__(256)
sk_heavy_kmers_kernel(const dict_view d, const uint64_t first_tuple, const uint64_t num_tuples, const uint8_t* __restrict__ flags,
                      const uint64_t* __restrict__ occ, unsigned long long* __restrict__ cursor, uint64_t* __restrict__ item_keys,
                      uint64_t* __restrict__ item_vals) {
    const uint32_t km = d.k - d.sk.m, per = km + 1;
    const uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t t = first_tuple + g / per;
    const uint32_t a = uint32_t(g % per);
    bool mine = false;
    uint64_t key = 0, v = 0;
    if (t < num_tuples && flags[t] != SK_ITEM_INLINE) {
        v = occ[t] & SK_VAL_MASK;
        const uint64_t p = v >> 1;
        if (p + a >= km && p + a - km + d.k <= d.num_bases) {
            const window_t<W> w = read_window<W>(d.granules, p + a - km, d.k);
            if (!w.crosses) {
                const kmer_w<W> x = w.kmer, x_rc = kmer_revcomp<W>(x, d.k);
                const sk_key_t kk = sk_key<W>(x, x_rc, d.k, d.sk.m);
                if (sk_usable(d, kk)) {
                    const uint64_t q = (p + a - km) + (kk.rc ? km - kk.pos : kk.pos);  // where this k-mer's key occurrence lies
                    mine = q == p && (kk.rc ? 1u : 0u) == uint32_t(v & 1);
                    key = sk_kmer_key<W>(x, x_rc);
                }
            }
        }
    }
    const uint64_t ballot = __ballot(mine);
    if (ballot == 0) return;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t leader = uint32_t(__ffsll((unsigned long long)ballot)) - 1u;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(cursor, (unsigned long long)__popcll(ballot));
    base = __shfl(base, int(leader), 64);
    if constexpr (EMIT) {
        if (mine) {
            const uint64_t at = base + uint64_t(__popcll(ballot & ((uint64_t(1) << lane) - 1)));
            item_keys[at] = key;
            item_vals[at] = v | (uint64_t(a) << SK_LEN_SHIFT);  // a: where the k-mer starts relative to the occurrence (p + a - km)
        }
    }
}

/* One lane per item, one launch per bucket choice: claim the first free slot of the item's bucket of this choice and
   fill it; an item that finds the bucket full sets the bucket's go-on flag and waits for the next round (after the
   last one: it is left to the complete path). flags == nullptr: every item is an inline slot (the heavy keys' k-mers). */
template <int W, bool COMPACT = false>
__global__ void __launch_bounds__(256)
sk_place_kernel(const dict_view d, const uint32_t choice, const uint64_t num_items, const uint64_t* __restrict__ item_keys,
                const uint64_t* __restrict__ item_vals, const uint8_t* __restrict__ flags, uint32_t* __restrict__ slots,
                const uint32_t num_buckets, uint8_t* __restrict__ placed, unsigned long long* __restrict__ stats,
                const uint32_t len_lo, const uint32_t len_hi) {
    const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    bool unplaced = false, used = false;
    if (t < num_items && !placed[t]) {
        const uint8_t kind = flags ? flags[t] : SK_ITEM_INLINE;
        /* who takes part in this launch: the items whose super-k-mer holds len_lo .. len_hi k-mers (a marker counts as
           the longest: every k-mer of its key passes through it) -- see the rounds in build_sk_table */
        const uint32_t len = kind == SK_ITEM_MARKER ? 63u : uint32_t(item_vals[t] >> SK_LEN_SHIFT);
        if (kind != SK_ITEM_NONE && (len < len_lo || len > len_hi)) {
            // not its turn
        } else if (kind == SK_ITEM_NONE) {
            placed[t] = 1;
        } else {
            const sk_hash_t h = sk_hash(item_keys[t], num_buckets);
            if constexpr (COMPACT && W == 1) {
                /* k <= 31, the heavy keys' k-mers: a bucket is one line of three 20-byte entries behind a flags word
                   (device_layout.hpp: SK_KMER_ENTRIES_NARROW; lookup_device.hpp: sk_examine_kmer_line) */
                uint32_t* B = slots + 16 * uint64_t(h.bucket[choice]);
                uint32_t mine = SK_KMER_ENTRIES_NARROW;
                uint32_t cur = __hip_atomic_load(B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (;;) {
                    const uint32_t in_use = cur & ((1u << SK_KMER_ENTRIES_NARROW) - 1u);
                    if (in_use == (1u << SK_KMER_ENTRIES_NARROW) - 1u) break;  // full
                    const uint32_t e = uint32_t(__ffs(int(~in_use))) - 1u;
                    const uint32_t seen = atomicCAS(B, cur, cur | (1u << e));
                    if (seen == cur) {
                        mine = e;
                        break;
                    }
                    cur = seen;
                }
                if (mine == SK_KMER_ENTRIES_NARROW) {
                    atomicOr(B, (SK_GO_ON << choice) | (choice == 0 ? 1u << (SK_FILTER_SHIFT + sk_filter_index(h.fingerprint)) : 0u));
                    unplaced = choice + 1 == SK_CHOICES;
                } else {
                    placed[t] = 1;
                    used = true;
                    const uint64_t start = ((item_vals[t] & SK_VAL_MASK) >> 1) + (item_vals[t] >> SK_LEN_SHIFT) - (d.k - d.sk.m);
                    const window_t<1> w = read_window<1>(d.granules, start, d.k);
                    uint32_t* E = B + 1 + SK_KMER_ENTRY_WORDS * mine;
                    E[0] = uint32_t(w.kmer.w[0]);
                    E[1] = uint32_t(w.kmer.w[0] >> 32);
                    E[2] = uint32_t(start);
                    E[3] = w.string_id;
                    E[4] = uint32_t(start >> 32) & 0xFFu;
                }
            } else {
            /* COMPACT (k <= 63, the heavy keys' k-mers): 32-byte entries, two to a 64-byte bucket -- meta, string id, position of the
               K-MER | fingerprint, the k-mer itself (lookup_device.hpp: sk_examine_kmer_entry) */
            constexpr uint32_t SLOT_WORDS = COMPACT ? 8u : 8u * W;
            uint32_t* B = slots + (SK_BUCKET_SLOTS * SLOT_WORDS) * uint64_t(h.bucket[choice]);  // slot 0 of the bucket: carries the flags
            /* claim the first free slot of the bucket: set its valid bit unless somebody holds it (other lanes may be
               OR-ing flags into slot 0's word) */
            uint32_t* S = nullptr;
            for (uint32_t slot = 0; slot < SK_BUCKET_SLOTS && !S; ++slot) {
                uint32_t* T = B + slot * SLOT_WORDS;
                uint32_t cur = __hip_atomic_load(T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (!(cur & SK_VALID)) {
                    const uint32_t seen = atomicCAS(T, cur, cur | SK_VALID);
                    if (seen == cur) {
                        S = T;
                        break;
                    }
                    cur = seen;
                }
            }
            if (!S) {
                /* this item lives further along -- or, after the last choice, nowhere */
                atomicOr(B, (SK_GO_ON << choice) | (choice == 0 ? 1u << (SK_FILTER_SHIFT + sk_filter_index(h.fingerprint)) : 0u));
                unplaced = choice + 1 == SK_CHOICES;
            } else {
                placed[t] = 1;
                used = true;
                if (S != B) {
                    atomicOr(B, SK_SECOND_USED);
                    /* k <= 63: slot 1 lives in the bucket's second line; its fingerprint is kept in slot 0's spare bytes, so
                       that a probe fetches that line only when its own key is there */
                    if constexpr (W == 2 && !COMPACT) B[SK_SECOND_FINGERPRINT_WORD] = h.fingerprint;
                }
                uint32_t meta, d1;
                uint64_t w1;
                uint64_t body[2 * W];
                for (int i = 0; i < 2 * W; ++i) body[i] = 0;
                const uint64_t v = item_vals[t] & SK_VAL_MASK;
                const uint64_t p = v >> 1;
                if constexpr (COMPACT) {
                    const uint64_t start = p + (item_vals[t] >> SK_LEN_SHIFT) - (d.k - d.sk.m);
                    const window_t<W> w = read_window<W>(d.granules, start, d.k);
                    meta = 0;
                    d1 = w.string_id;
                    w1 = start | (uint64_t(h.fingerprint) << 40);
                    for (int i = 0; i < W; ++i) body[i] = w.kmer.w[i];
                } else if (kind == SK_ITEM_INLINE) {
                    const uint32_t km = d.k - d.sk.m;
                    const uint32_t sid = read_window<W>(d.granules, p, 1).string_id;
                    const uint64_t s_begin = d.endpoints[sid], s_end = d.endpoints[sid + 1];
                    const uint64_t left = p - s_begin < km ? p - s_begin : km;
                    const uint64_t right = s_end - (p + d.sk.m) < km ? s_end - (p + d.sk.m) : km;
                    meta = (uint32_t(v & 1) ? SK_STRAND : 0u) | (uint32_t(left) << SK_LEFT_SHIFT) | (uint32_t(right) << SK_RIGHT_SHIFT);
                    d1 = sid;
                    w1 = p | (uint64_t(h.fingerprint) << 40);
                    read_bases<W>(d.granules, int64_t(p) - int64_t(km), body);
                } else {
                    meta = SK_MARKER;
                    d1 = 0;
                    w1 = uint64_t(h.fingerprint) << 40;
                }
                S[1] = d1;
                reinterpret_cast<uint64_t*>(S)[1] = w1;
                for (int i = 0; i < (COMPACT ? W : 2 * W); ++i) reinterpret_cast<uint64_t*>(S)[2 + i] = body[i];
                if (meta) atomicOr(S, meta);
            }
            }
        }
    }
    /* statistics: one atomic per wave and counter, not per item */
    const uint64_t b2 = __ballot(unplaced), b3 = __ballot(used);
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        if (b2) atomicAdd(stats + 2, (unsigned long long)__popcll(b2));
        if (b3) atomicAdd(stats + 3, (unsigned long long)__popcll(b3));
    }
}

struct temp_buffers {  // freed on every exit path
    std::vector<void*> owned;
    template <typename T>
    T* alloc(uint64_t n) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<uint64_t>(n, 1) * sizeof(T)));
        owned.push_back(p);
        return static_cast<T*>(p);
    }
    void release(void* p) {
        for (auto& q : owned)
            if (q == p) {
                (void)hipFree(q);
                q = nullptr;
            }
    }
    /* hand a buffer over to the replica */
    void keep(void* p) {
        for (auto& q : owned)
            if (q == p) q = nullptr;
    }
    ~temp_buffers() {
        for (void* p : owned)
            if (p) (void)hipFree(p);
    }
};

}  // namespace

/* why a replica has no table (device_stats; sshash_device_stats out[11]): lookups then take the directory / MPHF path,
   about half as fast -- never silently: the caller can see it */
static void absent(device_replica& rep, uint32_t reason) {
    rep.sk_absent_reason = reason;
    if (std::getenv("SSHASH_AMD_VERBOSE")) fprintf(stderr, "[sshash_amd] device %d: no super-k-mer table (reason %u, sshash_amd.h)\n", rep.device, reason);
}

/* SSHASH_AMD_SK_DENSITY: how full the table is packed -- the footprint / rate trade of a replica, a documented switch.
     (unset), "default"   2.5 slots per item in the keys' region (3.0 at k <= 63), 1.75 (2.5) places per k-mer of a heavy key
     "compact"            1.6 / 1.35 (k <= 63: 2.0 / 1.75)
     a number x           x slots per item, 1.2 <= x <= 4.0; the k-mers' region scaled along
   C3, one box (profiles/r06/density_sweep_c3.txt): 2.5 -> 2.0 -> 1.6 -> 1.4 -> 1.25 slots per item = 16.8 -> 14.0 -> 11.8 -> 10.6 -> 9.8 B/k-mer
   in HBM for 40.2 -> 37.3 -> 35.8 -> 33.9 -> 29.7 G lookups/s: what a fuller table costs is second choices (a second line) and, past a load
   factor of 0.6, keys that find no slot in five choices and take the complete path (0.05 % of the keys at 2.5, 0.9 % at 1.6, 3.7 % at 1.25).
   Round 5 had folded the two knobs of round 2 (SSHASH_AMD_SK_SLOTS_PER_KEY / _PER_KMER) into the tests' hooks while INTEGRATION.md still
   offered them to deployments: they are named once on stderr if set, and this is what replaces them. */
static void sk_density(bool wide, double& slots_per_key, double& slots_per_kmer) {
    const double key_default = wide ? SK_SLOTS_PER_KEY_WIDE : SK_SLOTS_PER_KEY, kmer_default = wide ? SK_SLOTS_PER_KMER_WIDE : SK_SLOTS_PER_KMER_NARROW;
    slots_per_key = key_default;
    slots_per_kmer = kmer_default;
    for (char const* gone : {"SSHASH_AMD_SK_SLOTS_PER_KEY", "SSHASH_AMD_SK_SLOTS_PER_KMER"}) {
        static std::atomic<bool> said{false};
        if (std::getenv(gone) && !said.exchange(true))
            fprintf(stderr, "[sshash_amd] %s is no longer read (round 5): use SSHASH_AMD_SK_DENSITY=compact or =<slots per item> (INTEGRATION.md)\n", gone);
    }
    char const* e = std::getenv("SSHASH_AMD_SK_DENSITY");
    if (!e || !*e || std::strcmp(e, "default") == 0) return;
    if (std::strcmp(e, "compact") == 0) {
        slots_per_key = wide ? 2.0 : 1.6;
        slots_per_kmer = wide ? 1.75 : 1.35;
        return;
    }
    char* end = nullptr;
    const double x = std::strtod(e, &end);
    if (end == e || *end || !(x >= 1.2 && x <= 4.0)) {
        fprintf(stderr, "[sshash_amd] SSHASH_AMD_SK_DENSITY=%s: expected default, compact or a number of slots per item in [1.2, 4.0]; using the default\n", e);
        return;
    }
    slots_per_key = x;
    slots_per_kmer = std::min(kmer_default, std::max(1.2, x * kmer_default / key_default));
}

uint64_t hbm_budget() {
    if (const char* e = std::getenv("SSHASH_AMD_HBM_BUDGET")) return std::strtoull(e, nullptr, 10);
    return 0;
}

/* Length of the table's key m-mers (sk_view::m). The table elects its own key, so this need not be the dictionary's m:
     k <= 31  the dictionary's m. Same-box sweep on C3 (m = 21; profiles/r04/table_key_length_sweep.txt): shorter keys make fewer,
              longer super-k-mers but put more k-mers under heavy keys -- 19: 15.7 B/k-mer instead of 16.8 for -3 % (streaming
              query -10 %), 17: 15.5 for -8 %, 15: worse on both counts --, longer keys the opposite (23: 18.9 B/k-mer for +2-5 %);
     k > 31   at least k - 32, so that a super-k-mer holds at most 33 k-mers: C4 (k = 63, m = 25 -> 31) has 166 M k-mers under heavy
              keys instead of 271 M and 66 candidates to elect instead of 78 -- 13.35 B/k-mer instead of 13.86, lookups +2 %, the
              streaming query 31.5 -> 35.6 G k-mers/s (profiles/r04/table_key_length_sweep_c4.txt).
   SSHASH_AMD_SK_M asks for another one: between 12 (the election hashes an occurrence's first 12 bases) and min(k - 1, 31) (a key
   is one word, and all ones means no key), and long enough that a super-k-mer's 2k - m bases fit a slot's 64 (k <= 31) or 128
   (k <= 63). Every rank of a sharded lookup must use the same. */
uint32_t sk_table_m(uint32_t k, uint32_t m) {
    const uint32_t bases_in_slot = k <= 31 ? 64 : 128;
    uint32_t lowest = 12;
    if (2 * k > bases_in_slot + lowest) lowest = 2 * k - bases_in_slot;
    if (k - lowest > 62) lowest = k - 62;  // positions take six bits
    const uint32_t highest = k - 1 < 31 ? k - 1 : 31;
    uint32_t chosen = m;
    if (k > 31 && m + 32 < k) chosen = k - 32;
    if (chosen < lowest || chosen > highest) chosen = m;  // (cannot happen for k <= 63; a dictionary's own m is always usable)
    if (const char* e = std::getenv("SSHASH_AMD_SK_M")) {  // measurement knob, tests
        const uint32_t want = uint32_t(std::atoi(e));
        if (want >= lowest && want <= highest) chosen = want;
    }
    return chosen;
}

void build_sk_table(device_replica& rep, host_index const& idx, uint32_t table_shards, uint32_t table_shard_id) {
    dict_view& v = rep.view;
    v.sk.slots = nullptr;
    v.sk.num_buckets = 0;
    v.sk.kmer_buckets = 0;
    v.sk.enabled = 0;
    v.sk.m = sk_table_m(idx.k, idx.m);  // (before any return: the routing kernel of a table-sharded lookup elects keys on a replica without a table too)
    v.sk.num_shards = table_shards;  // read by the scan kernel's filter
    v.sk.shard_id = table_shard_id;
    const char* env = std::getenv("SSHASH_AMD_SKTABLE");
    if (env && env[0] == '0') return absent(rep, SK_ABSENT_DISABLED);
    if (idx.num_shards > 1 || idx.num_kmers == 0) return absent(rep, SK_ABSENT_MINIMIZER_SHARD);  // a shard holds only its own minimizers' buckets: keep its path
    if (idx.num_bases >= (uint64_t(1) << 39)) return absent(rep, SK_ABSENT_TOO_MANY_BASES);  // positions are stored in 40 bits with a strand bit

    const uint64_t positions = idx.num_bases - idx.k + 1;
    const uint64_t num_waves = (positions + 1 + NEW_PER_WAVE - 1) / NEW_PER_WAVE;
    const dim3 block(256);
    if (num_waves >= (uint64_t(1) << 31)) return absent(rep, SK_ABSENT_TOO_MANY_ITEMS);  // hipCUB item counts are int
    /* one lane per k-mer start, in pieces of 2^24 waves (2^30 threads) */
    auto scan = [&](auto kernel, uint32_t* counts, uint64_t const* offsets, uint64_t* keys, uint64_t* vals) {
        const uint64_t piece = uint64_t(1) << 24;
        for (uint64_t first = 0; first < num_waves; first += piece) {
            const uint64_t waves = std::min(piece, num_waves - first);
            hipLaunchKernelGGL(kernel, dim3(uint32_t((waves * WAVE + 255) / 256)), block, 0, 0, v, first, num_waves, counts, offsets, keys, vals);
        }
    };

    temp_buffers tmp;
    uint32_t* counts = tmp.alloc<uint32_t>(num_waves);
    uint64_t* offsets = tmp.alloc<uint64_t>(num_waves + 1);
    const bool wide = idx.k > 31;
    if (wide) scan(sk_scan_kernel<2, false>, counts, nullptr, nullptr, nullptr);
    else scan(sk_scan_kernel<1, false>, counts, nullptr, nullptr, nullptr);
    HIP_CHECK(hipGetLastError());
    {
        /* exclusive scan of the per-wave counts into 64-bit offsets */
        size_t bytes = 0;
        auto in = hipcub::TransformInputIterator<uint64_t, hipcub::CastOp<uint64_t>, uint32_t*>(counts, hipcub::CastOp<uint64_t>());
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, offsets, int(num_waves)));
        void* scratch = tmp.alloc<uint8_t>(bytes);
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scratch, bytes, in, offsets, int(num_waves)));
        tmp.release(scratch);
    }
    uint64_t last_offset = 0;
    uint32_t last_count = 0;
    HIP_CHECK(hipMemcpy(&last_offset, offsets + (num_waves - 1), 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(&last_count, counts + (num_waves - 1), 4, hipMemcpyDeviceToHost));
    const uint64_t T = last_offset + last_count;  // super-k-mers
    if (T == 0 || T >= (uint64_t(1) << 31)) return absent(rep, SK_ABSENT_TOO_MANY_ITEMS);

    uint64_t* keys = tmp.alloc<uint64_t>(T);
    uint64_t* vals = tmp.alloc<uint64_t>(T);
    if (wide) scan(sk_scan_kernel<2, true>, counts, offsets, keys, vals);
    else scan(sk_scan_kernel<1, true>, counts, offsets, keys, vals);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
    tmp.release(counts);
    tmp.release(offsets);

    uint64_t* keys_sorted = tmp.alloc<uint64_t>(T);
    uint64_t* occ = tmp.alloc<uint64_t>(T);
    {
        size_t bytes = 0;
        HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys, keys_sorted, vals, occ, int(T), 0, int(2 * v.sk.m)));
        void* scratch = tmp.alloc<uint8_t>(bytes);
        HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(scratch, bytes, keys, keys_sorted, vals, occ, int(T), 0, int(2 * v.sk.m)));
        HIP_CHECK(hipDeviceSynchronize());
        tmp.release(scratch);
    }
    tmp.release(vals);
    /* runs of equal keys; `keys` is reused for the distinct keys */
    uint32_t* run_sizes = tmp.alloc<uint32_t>(T);
    uint64_t* d_num_runs = tmp.alloc<uint64_t>(1);
    {
        size_t bytes = 0;
        HIP_CHECK(hipcub::DeviceRunLengthEncode::Encode(nullptr, bytes, keys_sorted, keys, run_sizes, d_num_runs, int(T)));
        void* scratch = tmp.alloc<uint8_t>(bytes);
        HIP_CHECK(hipcub::DeviceRunLengthEncode::Encode(scratch, bytes, keys_sorted, keys, run_sizes, d_num_runs, int(T)));
        HIP_CHECK(hipDeviceSynchronize());
        tmp.release(scratch);
    }
    uint64_t K = 0;
    HIP_CHECK(hipMemcpy(&K, d_num_runs, 8, hipMemcpyDeviceToHost));
    double slots_per_key, slots_per_kmer;
    sk_density(wide, slots_per_key, slots_per_kmer);
    slots_per_key = test_hook_f64("slots_per_key", slots_per_key, 1.2, 16.0);  // (tests: a packed table, where second choices and unplaced items are common)
    if (K == 0) return absent(rep, SK_ABSENT_TOO_MANY_ITEMS);
    uint32_t* run_begins = tmp.alloc<uint32_t>(K);
    {
        size_t bytes = 0;
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, run_sizes, run_begins, int(K)));
        void* scratch = tmp.alloc<uint8_t>(bytes);
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scratch, bytes, run_sizes, run_begins, int(K)));
        tmp.release(scratch);
    }
    /* stats: 2 unplaced items, 3 slots used, 4 heavy keys, 5 their occurrences, 6 / 7 cursors of the heavy k-mer passes */
    unsigned long long* stats = tmp.alloc<unsigned long long>(8 + 2 * SK_HIST_BINS);
    HIP_CHECK(hipMemset(stats, 0, 8 * (8 + 2 * SK_HIST_BINS)));
    uint8_t* flags = tmp.alloc<uint8_t>(T);
    HIP_CHECK(hipMemset(flags, SK_ITEM_INLINE, T));
    hipLaunchKernelGGL(sk_classify_kernel, dim3(uint32_t((K + 255) / 256)), block, 0, 0, K, run_sizes, run_begins, flags, stats, stats + 8);
    HIP_CHECK(hipGetLastError());
    tmp.release(run_begins);
    tmp.release(run_sizes);
    tmp.release(keys);
    /* the k-mers of the heavy keys, keyed one by one: count, then emit. A launch of 2^32 threads or more is not carried
       out (and reports no error), so the tuples are walked in pieces. */
    const uint64_t per = idx.k - v.sk.m + 1;
    const uint64_t tuples_per_launch = (uint64_t(1) << 30) / per;
    auto heavy_pass = [&](bool emit, unsigned long long* cursor, uint64_t* item_keys, uint64_t* item_vals) {
        for (uint64_t first = 0; first < T; first += tuples_per_launch) {
            const uint64_t count = std::min(tuples_per_launch, T - first);
            const dim3 pair_grid(uint32_t((count * per + 255) / 256));
            if (wide && emit) hipLaunchKernelGGL((sk_heavy_kmers_kernel<2, true>), pair_grid, block, 0, 0, v, first, first + count, flags, occ, cursor, item_keys, item_vals);
            else if (wide) hipLaunchKernelGGL((sk_heavy_kmers_kernel<2, false>), pair_grid, block, 0, 0, v, first, first + count, flags, occ, cursor, item_keys, item_vals);
            else if (emit) hipLaunchKernelGGL((sk_heavy_kmers_kernel<1, true>), pair_grid, block, 0, 0, v, first, first + count, flags, occ, cursor, item_keys, item_vals);
            else hipLaunchKernelGGL((sk_heavy_kmers_kernel<1, false>), pair_grid, block, 0, 0, v, first, first + count, flags, occ, cursor, item_keys, item_vals);
            HIP_CHECK(hipGetLastError());
        }
    };
    heavy_pass(false, stats + 6, nullptr, nullptr);
    unsigned long long h_stats[8];
    HIP_CHECK(hipMemcpy(h_stats, stats, 64, hipMemcpyDeviceToHost));
    const uint64_t heavy_keys = h_stats[4], heavy_occurrences = h_stats[5], heavy_kmers = h_stats[6];
    if (heavy_kmers >= (uint64_t(1) << 31)) return absent(rep, SK_ABSENT_TOO_MANY_ITEMS);
    uint64_t* kmer_keys = tmp.alloc<uint64_t>(heavy_kmers);
    uint64_t* kmer_vals = tmp.alloc<uint64_t>(heavy_kmers);
    if (heavy_kmers) {
        heavy_pass(true, stats + 7, kmer_keys, kmer_vals);
        HIP_CHECK(hipMemcpy(h_stats, stats, 64, hipMemcpyDeviceToHost));
        if (h_stats[7] != heavy_kmers) throw error(error_kind::hip, "super-k-mer table: the two passes over the heavy keys disagree");
    }
    /* slots asked for: one per occurrence of a light key, one marker per heavy key -- the keys' region -- and one per k-mer of
       a heavy key, in the region behind it (sk_view::kmer_buckets) */
    slots_per_kmer = test_hook_f64("slots_per_kmer", slots_per_kmer, 1.2, 16.0);
    const uint64_t wanted = (T - heavy_occurrences) + heavy_keys + heavy_kmers;
    const uint64_t key_buckets = uint64_t(double(wanted - heavy_kmers) * slots_per_key / SK_BUCKET_SLOTS) + 8;
    /* a bucket of the k-mers' region is one 64-byte line: two 32-byte entries at k <= 63, three 20-byte ones at k <= 31 (the k-mer
       itself instead of its super-k-mer's bases) */
    const uint64_t kmer_entries = wide ? SK_BUCKET_SLOTS : SK_KMER_ENTRIES_NARROW;
    const uint64_t kmer_buckets = heavy_kmers ? uint64_t(double(heavy_kmers) * slots_per_kmer / double(kmer_entries)) + 8 : 0;
    const uint64_t num_buckets = key_buckets + kmer_buckets;
    const uint64_t slot_bytes = wide ? 64 : 32;
    const uint64_t table_bytes = key_buckets * SK_BUCKET_SLOTS * slot_bytes + kmer_buckets * 64;
    if (num_buckets >= (uint64_t(1) << 32)) return absent(rep, SK_ABSENT_TOO_MANY_ITEMS);
    /* k <= 63: the in-wave hops post a bucket as a 32-bit LINE number -- two lines a bucket of the keys' region, one of the k-mers'
       (lookup_device.hpp: sk_finish_in_wave<2>): that number must fit too (a table of 256 GB would have wrapped silently) */
    if (wide && 2 * key_buckets + kmer_buckets >= (uint64_t(1) << 32)) return absent(rep, SK_ABSENT_TOO_MANY_ITEMS);
    {
        size_t free_bytes = 0, total_bytes = 0;
        HIP_CHECK(hipMemGetInfo(&free_bytes, &total_bytes));
        if (table_bytes > free_bytes / 2) return absent(rep, SK_ABSENT_NO_MEMORY);  // leave HBM for the caller's batches
        /* SSHASH_AMD_HBM_BUDGET (bytes): what ONE replica may hold -- how a deployment keeps room for several dictionaries,
           and how the tests make a dictionary "larger than the HBM" on a box whose HBM it would fit many times over */
        const uint64_t budget = hbm_budget();
        if (budget && rep.bytes + rep.bytes_still_to_come + table_bytes > budget) return absent(rep, SK_ABSENT_NO_MEMORY);
    }
    uint32_t* slots = tmp.alloc<uint32_t>(table_bytes / 4);
    if (const char* e = std::getenv("SSHASH_AMD_VERBOSE"); e && e[0] == '1')
        std::fprintf(stderr, "[sshash_amd] super-k-mer table: %.3f GB at %p (keys' region %llu buckets, k-mers' region %llu)\n", double(table_bytes) / 1e9,
                     (void*)slots, (unsigned long long)key_buckets, (unsigned long long)kmer_buckets);
    uint8_t* placed = tmp.alloc<uint8_t>(T + heavy_kmers);
    HIP_CHECK(hipMemset(slots, 0, table_bytes));
    HIP_CHECK(hipMemset(placed, 0, T + heavy_kmers));
    /* Rounds: one per bucket choice; the FIRST choice in three turns, long super-k-mers first. At load factor 0.4 with two
       slots per bucket 7.2 % of the items do not fit their first bucket whatever the order (Poisson), but WHICH items
       matters: a query is a k-mer, and an item answers as many queries as its super-k-mer holds k-mers (1 .. k-m+1, a
       quarter of them the full k-m+1). Served in order of length, the overflowing 7.2 % of the items hold 3.9 % of the
       k-mers instead of 7.2 % -- that many fewer positive queries need a second bucket (resume pass). */
    const uint32_t most = idx.k - v.sk.m + 1;
    const uint32_t turns[3][2] = {{(7 * most + 9) / 10, 63u}, {(35 * most + 99) / 100, (7 * most + 9) / 10 - 1}, {0u, (35 * most + 99) / 100 - 1}};
    uint32_t* kmer_slots = slots + key_buckets * SK_BUCKET_SLOTS * (slot_bytes / 4);  // the k-mers' region
    auto place = [&](uint32_t choice, uint32_t lo, uint32_t hi, bool with_heavy_kmers) {
        const dim3 g1(uint32_t((T + 255) / 256)), g2(uint32_t((heavy_kmers + 255) / 256));
        if (wide) {
            hipLaunchKernelGGL(sk_place_kernel<2>, g1, block, 0, 0, v, choice, T, keys_sorted, occ, flags, slots, uint32_t(key_buckets), placed, stats, lo, hi);
            if (heavy_kmers && with_heavy_kmers)
                hipLaunchKernelGGL((sk_place_kernel<2, true>), g2, block, 0, 0, v, choice, heavy_kmers, kmer_keys, kmer_vals, (const uint8_t*)nullptr, kmer_slots,
                                   uint32_t(kmer_buckets), placed + T, stats, 0u, 63u);
        } else {
            hipLaunchKernelGGL(sk_place_kernel<1>, g1, block, 0, 0, v, choice, T, keys_sorted, occ, flags, slots, uint32_t(key_buckets), placed, stats, lo, hi);
            if (heavy_kmers && with_heavy_kmers)
                hipLaunchKernelGGL((sk_place_kernel<1, true>), g2, block, 0, 0, v, choice, heavy_kmers, kmer_keys, kmer_vals, (const uint8_t*)nullptr, kmer_slots,
                                   uint32_t(kmer_buckets), placed + T, stats, 0u, 63u);
        }
        HIP_CHECK(hipGetLastError());
    };
    for (int turn = 0; turn < 3; ++turn)
        if (turns[turn][0] <= turns[turn][1]) place(0, turns[turn][0], turns[turn][1], turn == 2);  // (the heavy keys' k-mers: one k-mer each)
    for (uint32_t choice = 1; choice < SK_CHOICES; ++choice) place(choice, 0u, 63u, true);
    HIP_CHECK(hipMemcpy(h_stats, stats, 64, hipMemcpyDeviceToHost));
    HIP_CHECK(hipDeviceSynchronize());

    tmp.keep(slots);
    rep.allocations.push_back(slots);
    rep.bytes += table_bytes;
    {
        unsigned long long hist[2 * SK_HIST_BINS];
        HIP_CHECK(hipMemcpy(hist, stats + 8, sizeof(hist), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < 2 * SK_HIST_BINS; ++i) rep.sk_histogram[i] = hist[i];
        rep.sk_histogram[18] = T;
        rep.sk_histogram[19] = wanted;
    }
    rep.sk_keys = K;
    rep.sk_heavy_keys = heavy_keys;
    rep.sk_heavy_kmers = heavy_kmers;
    rep.sk_unplaced = h_stats[2];
    rep.sk_slots_used = h_stats[3];
    rep.sk_bytes = table_bytes;
    rep.sk_absent_reason = 0;
    v.sk.slots = slots;
    v.sk.num_buckets = uint32_t(key_buckets);
    v.sk.kmer_buckets = uint32_t(kmer_buckets);
    v.sk.enabled = 1;
}

}  // namespace sshash_amd
