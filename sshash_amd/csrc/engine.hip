// engine.hip -- device replicas, the batched lookup kernels and their launchers (gfx950 only).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <exception>
#include <stdexcept>
#include <thread>

#include "engine.hpp"
#include "hooks.hpp"
#include "replica.hpp"
#include "sktable.hpp"

namespace sshash_amd {

/* One lane per item, one launch: a launch of 2^32 threads or more is NOT carried out and reports no error (DESIGN.md
   section 6), so entry points that launch once say so instead of returning untouched output. The lookups themselves are
   split into pieces and take any n. */
static void check_single_launch(uint64_t lanes, char const* what) {
    if (lanes >= (uint64_t(1) << 32))
        throw error(error_kind::argument, std::string(what) + ": at most 2^32 - 1 lanes per call (split the batch)");
}


int visible_device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* strings + endpoints -> granules (k > 31) or atoms (k <= 31), device_layout.hpp (1) */
static std::vector<granule> make_granules(host_index const& idx, uint32_t num_threads) {
    const uint64_t G = idx.num_bases / GRANULE_BASES + 1 + GRANULE_PAD;
    std::vector<granule> g(G);
    detail::parallel_ranges(G, num_threads, [&](uint64_t b, uint64_t e, uint32_t) {
        for (uint64_t i = b; i < e; ++i) {
            g[i].rank = 0;
            g[i].marks = 0;
            g[i].bases = i < idx.strings.size() ? idx.strings[i] : 0;
        }
    });
    if (idx.endpoints.size() >= (uint64_t(1) << 32)) throw error(error_kind::build, "more than 2^32 strings");
    for (uint64_t e : idx.endpoints) g[e >> 5].marks |= 1u << (e & 31);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < G; ++i) {
        g[i].rank = acc;
        acc += uint32_t(__builtin_popcount(g[i].marks));
    }
    return g;
}

static std::vector<atom32> make_atoms(std::vector<granule> const& g, uint32_t num_threads) {
    const uint64_t A = g.size() - 1;  // every atom also carries the next block
    std::vector<atom32> a(A);
    detail::parallel_ranges(A, num_threads, [&](uint64_t b, uint64_t e, uint32_t) {
        for (uint64_t i = b; i < e; ++i) {
            a[i].bases[0] = g[i].bases;
            a[i].bases[1] = g[i + 1].bases;
            a[i].marks = uint64_t(g[i].marks) | (uint64_t(g[i + 1].marks) << 32);
            a[i].rank = g[i].rank;
            a[i].spare = 0;
        }
    });
    return a;
}

/* Widen the packed control codewords to u64 entries carrying the bucket minimizer's fingerprint
   (device_layout.hpp (3)). One lane per minimizer id; runs once per upload. */
__global__ void __launch_bounds__(256)
widen_codewords_kernel(const dict_view d, const uint64_t* __restrict__ packed, const uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t code = packed_get(packed, i, d.cw_width);
    uint64_t first;  // offset of the bucket's first minimizer position
    if ((code & 1) == 0) first = code >> 1;
    else if ((code & 3) == 1) {
        const uint32_t size = uint32_t((code >> 2) & (MAX_BUCKET_SMALL - 1)) + 2;
        first = packed_get(d.mid_load, uint64_t(d.begin_buckets_of_size[size]) + (code >> (2 + MIN_L)) * size, d.off_width);
    } else {
        first = packed_get(d.heavy_load, code >> 5, d.off_width);
    }
    const uint64_t mmer = read_mmer(d, first);
    out[i] = code | (minimizer_fingerprint(mmer, d.m, d.canonical != 0, d.cw_width) << d.cw_width);
}

/* ---- minimizer directory construction (device_layout.hpp (4)) --------------------------------- */

/* One lane per minimizer id: recover the key (the m-mer at the bucket's first offset; for canonical
   indexes whichever of {m-mer, its reverse complement} the MPHF maps back to this id) and claim a
   slot in its directory bucket. Keys beyond 4 per bucket are dropped: the finalize pass flags the bucket. */
__global__ void __launch_bounds__(256)
directory_insert_kernel(const dict_view d, const uint64_t n, uint64_t* __restrict__ buckets, uint32_t* __restrict__ claimed,
                        const uint32_t num_buckets) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t code = d.codewords[i] & low_mask(d.cw_width);
    uint64_t first;
    if ((code & 1) == 0) first = code >> 1;
    else if ((code & 3) == 1) {
        const uint32_t size = uint32_t((code >> 2) & (MAX_BUCKET_SMALL - 1)) + 2;
        first = packed_get(d.mid_load, uint64_t(d.begin_buckets_of_size[size]) + (code >> (2 + MIN_L)) * size, d.off_width);
    } else {
        first = packed_get(d.heavy_load, code >> 5, d.off_width);
    }
    const uint64_t mmer = read_mmer(d, first);
    uint64_t keys[2] = {mmer, mmer};
    int num = 1;
    if (d.canonical) {
        const uint64_t rc = mmer_revcomp(mmer, d.m);
        const bool a = mphf_eval(d.minimizers, city128_u64(mmer, d.minimizers.seed)) == i;
        const bool b = rc != mmer && mphf_eval(d.minimizers, city128_u64(rc, d.minimizers.seed)) == i;
        num = 0;
        if (a) keys[num++] = mmer;
        if (b) keys[num++] = rc;  // both only when a non-key collides with this id: the MPHF path would behave the same
    }
    for (int j = 0; j < num; ++j) {
        const uint64_t h = directory_hash(keys[j]);
        const uint32_t b = directory_bucket(h, num_buckets);
        const uint32_t slot = atomicAdd(claimed + b, 1u);
        if (slot < DIR_SLOTS) buckets[4 * uint64_t(b) + slot] = directory_entry(code, directory_fingerprint(h));
    }
}

/* One lane per bucket: drop entries whose fingerprint repeats inside the bucket (the dropped key stays
   reachable through the MPHF) and set the overflow flag where anything was lost. */
__global__ void __launch_bounds__(256)
directory_finalize_kernel(uint64_t* __restrict__ buckets, const uint32_t* __restrict__ claimed, const uint32_t num_buckets,
                          unsigned long long* __restrict__ stats) {
    const uint64_t s = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (s >= num_buckets) return;
    uint64_t* B = buckets + 4 * s;
    const uint32_t c = claimed[s];
    bool overflow = c > DIR_SLOTS;
    uint32_t count = c > DIR_SLOTS ? DIR_SLOTS : c;
    uint64_t e[DIR_SLOTS];
    for (uint32_t j = 0; j < DIR_SLOTS; ++j) e[j] = j < count ? B[j] : 0;
    for (uint32_t a = 0; a < count; ++a) {
        for (uint32_t b = a + 1; b < count;) {
            if (((e[a] >> 40) & 0xFFFF) == ((e[b] >> 40) & 0xFFFF)) {
                e[b] = e[count - 1];
                e[count - 1] = 0;
                --count;
                overflow = true;
            } else {
                ++b;
            }
        }
    }
    if (overflow) e[0] |= uint64_t(1) << 63;
    for (uint32_t j = 0; j < DIR_SLOTS; ++j) B[j] = e[j];
    if (overflow) atomicAdd(stats, 1ull);
    atomicAdd(stats + 1, (unsigned long long)count);
}

engine::engine(std::shared_ptr<host_index> idx) : m_idx(std::move(idx)) {}
engine::~engine() = default;

bool engine::on_device(int device) const {
    std::shared_lock<std::shared_mutex> lock(m_replicas_mutex);
    for (auto const& r : m_replicas)
        if (r->device == device) return true;
    return false;
}

std::vector<int> engine::devices() const {
    std::vector<int> d;
    std::shared_lock<std::shared_mutex> lock(m_replicas_mutex);
    for (auto const& r : m_replicas) d.push_back(r->device);
    return d;
}

uint64_t engine::device_bytes(int device) const { return replica(device)->bytes; }

void engine::device_stats(int device, uint64_t out[16]) const {
    device_replica const* r = replica(device);
    out[0] = r->bytes;
    out[1] = r->view.directory.enabled ? r->view.directory.num_buckets : 0;
    out[2] = r->directory_overflowed;
    out[3] = r->directory_entries;
    /* places an item can take: two slots per bucket of the keys' region; per bucket of the k-mers' region two entries (k <= 63) or three (k <= 31) */
    out[4] = r->view.sk.enabled ? uint64_t(r->view.sk.num_buckets) * SK_BUCKET_SLOTS +
                                      uint64_t(r->view.sk.kmer_buckets) * (r->view.k > 31 ? SK_BUCKET_SLOTS : SK_KMER_ENTRIES_NARROW)
                                : 0;
    out[5] = r->sk_keys;
    out[6] = r->sk_keys - r->sk_heavy_keys;
    out[7] = r->sk_unplaced;
    out[8] = r->sk_slots_used;
    out[9] = r->sk_heavy_keys;
    out[10] = r->sk_heavy_kmers;
    out[11] = r->view.sk.enabled ? 0 : r->sk_absent_reason;
    out[12] = r->sk_bytes;
    out[13] = r->view.sk.m;  // length of the table's keys (sk_table_m; the ranks of a sharded lookup compare it: sharded.cpp)
    out[14] = out[15] = 0;
}

void engine::device_table_histogram(int device, uint64_t out[32]) const {
    device_replica const* r = replica(device);
    for (int i = 0; i < 32; ++i) out[i] = i < 20 && r->view.sk.enabled ? r->sk_histogram[i] : 0;
}

device_replica const* engine::replica(int device) const {
    std::shared_lock<std::shared_mutex> lock(m_replicas_mutex);
    for (auto const& r : m_replicas)
        if (r->device == device) return r.get();
    throw error(error_kind::no_device, "dictionary is not resident on device " + std::to_string(device) +
                             " (call sshash_to_device first)");
}

/* How this library's device code was built (csrc/Makefile): through tools/isa_guard.py -- SSHASH_ISA_GUARDED is defined by that
   recipe's host-side compile and by nothing else -- or by a plain hipcc, whose register allocation may expose kernels to the
   last-VGPR hazard of gfx950 (HISTORY.md). */
char const* isa_guard_state() {
#if defined(SSHASH_ISA_GUARDED)
    return "guarded";
#else
    return "plain";
#endif
}

void engine::to_device(int device, uint32_t table_shards, uint32_t table_shard_id) {
#if !defined(SSHASH_ISA_GUARDED)
    {
        static std::once_flag once;
        std::call_once(once, [] {
            fprintf(stderr, "[sshash_amd] WARNING: this library was NOT built through tools/isa_guard.py (make -C sshash_amd/csrc): on gfx950 a "
                                "kernel that keeps a 64-bit shift amount in the last VGPR of its allocation computes wrongly some of the time "
                                "(HISTORY.md); tests/test_isa_guard.py names the kernels of a build that are exposed\n");
        });
    }
#endif
    if (table_shards == 0 || table_shard_id >= table_shards) throw error(error_kind::argument, "table shard id must be < number of table shards");
    std::lock_guard<std::mutex> one_upload_at_a_time(m_upload_mutex);
    {
        std::shared_lock<std::shared_mutex> lock(m_replicas_mutex);
        for (auto const& r : m_replicas) {
            if (r->device != device) continue;
            if (r->view.sk.num_shards != table_shards || r->view.sk.shard_id != table_shard_id)
                throw error(error_kind::argument, "the dictionary is already resident on device " + std::to_string(device) +
                                                      " with another table sharding (" + std::to_string(r->view.sk.shard_id) + " of " +
                                                      std::to_string(r->view.sk.num_shards) + ")");
            return;
        }
    }
    const int count = visible_device_count();
    if (count == 0) throw error(error_kind::no_device, "no HIP device visible: the lookup path requires an MI355X (no CPU fallback)");
    if (device < 0 || device >= count) throw error(error_kind::no_device, "invalid device ordinal " + std::to_string(device));
    device_guard guard(device);
    host_index const& idx = *m_idx;
    auto rep = std::make_unique<device_replica>();
    rep->device = device;
    dict_view& v = rep->view;
    v.k = idx.k;
    v.m = idx.m;
    v.canonical = idx.canonical;
    v.skew_parts = idx.skew_num_partitions;
    v.hash_magic = idx.hash_magic;
    v.num_kmers = idx.num_kmers;
    v.num_strings = idx.num_strings;
    v.num_bases = idx.num_bases;
    {
        const uint32_t nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<granule> g = make_granules(idx, nt);
        if (idx.k <= 31) {
            std::vector<atom32> a = make_atoms(g, nt);
            std::vector<granule>().swap(g);
            v.granules = rep->put(a);
        } else {
            v.granules = rep->put(g);
        }
    }
    v.endpoints = rep->put(idx.endpoints);
    v.minimizers = rep->put_mphf(idx.minimizers_mphf);
    v.cw_width = idx.control_codewords.width;
    v.off_width = idx.mid_load_buckets.width;
    v.begin_buckets_of_size = rep->put(idx.begin_buckets_of_size);
    v.mid_load = rep->put(idx.mid_load_buckets.words);
    v.heavy_load = rep->put(idx.heavy_load_buckets.words);
    v.heavy_size = idx.heavy_load_buckets.size;
    /* the super-k-mer table first (it needs the atoms and the endpoints only): what else is built depends on it */
    v.codewords = nullptr;
    v.cw_packed = 0;
    v.directory.buckets = nullptr;
    v.directory.num_buckets = 0;
    v.directory.enabled = 0;
    /* what is uploaded BEHIND the table (skew index, weights): the table's budget test must leave room for it, or a table that
       just fits would take the finished replica past SSHASH_AMD_HBM_BUDGET and the upload would fail where a table-less replica
       was possible */
    rep->bytes_still_to_come = 8 * sizeof(skew_part_dev);
    for (uint32_t p = 0; p < 8; ++p) {
        if (p < idx.skew_num_partitions && idx.skew_mphfs[p].num_keys) {
            mphf_host const& f = idx.skew_mphfs[p];
            rep->bytes_still_to_come += f.parts.size() * sizeof(f.parts[0]) + (f.pilots.size() + f.free_slots.size() + idx.skew_positions[p].words.size()) * 8;
        } else {
            rep->bytes_still_to_come += 64;
        }
    }
    if (idx.weighted()) rep->bytes_still_to_come += (idx.weight_starts.size() + idx.weight_values.size()) * 8;
    build_sk_table(*rep, idx, table_shards, table_shard_id);
    /* With the table resident only the deferred queries (ties, items that found no slot: ~0.05 %) and the
       `minimizer_found` byte of a miss come through the minimizer structures: they keep the host index's form -- bit-packed
       control codewords, no directory (device_layout.hpp (3), (4)); 11.8 GB of a human-scale replica. SSHASH_AMD_DIRECTORY=1
       builds the directory (and the widened codewords) all the same, =0 never builds it. */
    const char* dir_env = std::getenv("SSHASH_AMD_DIRECTORY");
    const bool lean = v.sk.enabled && !(dir_env && dir_env[0] == '1');
    if (lean) {
        v.codewords = rep->put(idx.control_codewords.words);
        v.cw_packed = 1;
    } else {
        const uint64_t n = idx.control_codewords.size;
        uint64_t* packed = nullptr;
        uint64_t* wide = nullptr;
        const size_t packed_bytes = idx.control_codewords.words.size() * sizeof(uint64_t);
        HIP_CHECK(hipMalloc(&packed, packed_bytes));
        HIP_CHECK(hipMemcpy(packed, idx.control_codewords.words.data(), packed_bytes, hipMemcpyHostToDevice));
        HIP_CHECK(hipMalloc(&wide, std::max<uint64_t>(n, 1) * sizeof(uint64_t)));
        rep->allocations.push_back(wide);
        rep->bytes += std::max<uint64_t>(n, 1) * sizeof(uint64_t);
        if (n) {
            check_single_launch(n, "upload (one lane per minimizer)");
            hipLaunchKernelGGL(widen_codewords_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, 0, v, packed, n, wide);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipDeviceSynchronize());
        }
        HIP_CHECK(hipFree(packed));
        v.codewords = wide;
    }
    /* minimizer directory (device_layout.hpp (4)) */
    if (!lean) {
        const bool want = !(dir_env && dir_env[0] == '0');
        const uint64_t n = idx.control_codewords.size;
        const uint64_t nb = uint64_t(double(n) / DIR_LOAD) + 1;
        if (want && n && nb < (uint64_t(1) << 32) && v.cw_width <= DIR_CODE_BITS) {
            uint64_t* buckets = nullptr;
            uint32_t* claimed = nullptr;
            unsigned long long* stats = nullptr;
            HIP_CHECK(hipMalloc(&buckets, nb * 32));
            HIP_CHECK(hipMalloc(&claimed, nb * 4));
            HIP_CHECK(hipMalloc(&stats, 16));
            HIP_CHECK(hipMemset(buckets, 0, nb * 32));
            HIP_CHECK(hipMemset(claimed, 0, nb * 4));
            HIP_CHECK(hipMemset(stats, 0, 16));
            rep->allocations.push_back(buckets);
            rep->bytes += nb * 32;
            hipLaunchKernelGGL(directory_insert_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, 0, v, n, buckets, claimed, uint32_t(nb));
            hipLaunchKernelGGL(directory_finalize_kernel, dim3(uint32_t((nb + 255) / 256)), dim3(256), 0, 0, buckets, claimed, uint32_t(nb), stats);
            HIP_CHECK(hipGetLastError());
            unsigned long long h_stats[2];
            HIP_CHECK(hipMemcpy(h_stats, stats, 16, hipMemcpyDeviceToHost));
            HIP_CHECK(hipFree(stats));
            HIP_CHECK(hipFree(claimed));
            rep->directory_overflowed = h_stats[0];
            rep->directory_entries = h_stats[1];
            v.directory.buckets = buckets;
            v.directory.num_buckets = uint32_t(nb);
            v.directory.enabled = 1;
        }
    }
    std::vector<skew_part_dev> skew(8);
    for (uint32_t p = 0; p < 8; ++p) {
        std::memset(&skew[p], 0, sizeof(skew_part_dev));
        if (p < idx.skew_num_partitions && idx.skew_mphfs[p].num_keys) {
            skew[p].f = rep->put_mphf(idx.skew_mphfs[p]);
            skew[p].positions = rep->put(idx.skew_positions[p].words);
            skew[p].pos_width = idx.skew_positions[p].width;
        } else {
            /* an absent key may still be routed here: give it a well-formed 1-key function */
            mphf_host dummy;
            dummy.pilot_width = 1;
            dummy.num_keys = 1;
            mphf_partition part{};
            part.num_keys = 1;
            part.table_size = 1;
            part.dense_buckets = 1;
            part.sparse_buckets = 1;
            dummy.parts.push_back(part);
            dummy.pilots.assign(2, 0);
            dummy.free_slots.assign(1, 0);
            skew[p].f = rep->put_mphf(dummy);
            skew[p].positions = rep->put(std::vector<uint64_t>(2, 0));
            skew[p].pos_width = 1;
        }
    }
    rep->d_skew = rep->put(skew);
    v.weight_starts = v.weight_values = nullptr;
    v.num_weight_intervals = idx.weight_values.size();
    if (idx.weighted()) {
        v.weight_starts = rep->put(idx.weight_starts);
        v.weight_values = rep->put(idx.weight_values);
    }
    rep->create_scratch_pool();
    if (const uint64_t budget = hbm_budget(); budget && rep->bytes > budget)
        throw error(error_kind::no_device, "the replica needs " + std::to_string(rep->bytes) + " bytes of HBM, SSHASH_AMD_HBM_BUDGET allows " +
                                               std::to_string(budget) + ": partition the dictionary (minimizer shards, table shards)");
    std::unique_lock<std::shared_mutex> lock(m_replicas_mutex);
    m_replicas.push_back(std::move(rep));
}

/* ---- kernels --------------------------------------------------------------------------- */

template <int W, bool CANON, int MODE, bool ASCII>
__global__ void __launch_bounds__(256)
lookup_kernel(const dict_view d, const skew_part_dev* __restrict__ skew, const void* __restrict__ queries,
              const uint64_t n, const bool check_rc, const result_view out, uint8_t* __restrict__ member,
              const uint8_t* __restrict__ lane_valid) {
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (lane_valid && !(lane_valid[i] & 1)) continue;  // no query in this place (streaming lookup)
        kmer_w<W> x;
        if constexpr (ASCII) {
            x = kmer_from_ascii<W>(static_cast<const char*>(queries) + i * d.k, d.k);
        } else {
            const uint64_t* q = static_cast<const uint64_t*>(queries) + i * W;
            for (int j = 0; j < W; ++j) x.w[j] = q[j];
            x = kmer_take_chars<W>(x, d.k);
        }
        /* the full-result kernel keeps the MPHF path: exact `minimizer_found` (device_layout.hpp (4)) */
        const hit_t h = lookup_one<W, CANON, MODE != int(out_mode::full)>(d, skew, x, check_rc);
        if constexpr (MODE == int(out_mode::member)) {
            member[i] = h.found ? 1 : 0;
        } else {
            store_result<MODE == int(out_mode::full)>(d, out, i, h);
        }
    }
}

constexpr uint32_t DEFER_SHARDS = 2048;  // power of two
constexpr uint32_t DEFER_BLOCKS_PER_SHARD = 4;

/* The per-stream scratch of one launch sequence: two queues sharded over DEFER_SHARDS counters.
     resume  queries the first table pass hands to the second (index | choice, and the packed k-mer); half a
             piece's worth of places -- a query that finds it full goes to `defer` instead;
     defer   queries for the complete path (index only); a whole piece's worth: a query is pushed at most once. */
struct pass_queues {
    uint32_t* defer_counts;
    uint32_t* resume_counts;
    uint32_t* defer_index;
    uint32_t* resume_index;
    uint64_t* resume_kmers;
    uint64_t* resume_meta;     // replicas without the table: scan_meta() of the entry (the bucket-scan pass reads it)
    uint32_t defer_capacity;   // places per shard
    uint32_t resume_capacity;
    /* the caller wants `minimizer_found`: a hit has it (true) and every other field of a miss is known as well, so the first
       two passes store complete results for everything; only the flag of a MISS needs the MPHF (for an absent minimizer the
       reference's flag depends on which arbitrary bucket the MPHF lands on): misses are queued with DEFER_FLAG_ONLY set
       and the last pass stores that one byte for them */
    uint32_t flag_misses;
};
constexpr uint32_t DEFER_FLAG_ONLY = 1u << 31;  // in a deferred-queue entry (query indices stay below 2^27)

template <int W, bool ASCII>
__device__ __forceinline__ kmer_w<W> load_query(const void* __restrict__ queries, uint64_t i, uint32_t k) {
    kmer_w<W> x;
    if constexpr (ASCII) {
        x = kmer_from_ascii<W>(static_cast<const char*>(queries) + i * k, k);
    } else {
        const uint64_t* q = static_cast<const uint64_t*>(queries) + i * W;
        for (int j = 0; j < W; ++j) x.w[j] = __builtin_nontemporal_load(q + j);
        x = kmer_take_chars<W>(x, k);
    }
    return x;
}

/* First pass of the multi-pass lookup (lookup_device.hpp): one query per lane, common case only. Queries it
   cannot settle are appended to `queue` (their index in the batch). SK: through the super-k-mer table
   (device_layout.hpp (5)) instead of directory + atoms; the four lanes of a quad fetch each other's buckets
   together, through LDS (sk_probe_wave), so no lane leaves before the probe. */
/* (k <= 63 through the table: eight waves per SIMD asked for -- the pass runs at the random-line rate and wants the waves; the compiler fits
   its 66 registers into 64 without scratch: 31.8 -> 32.2 G lookups/s on C4, profiles/r04/eight_waves_ab.txt. The k <= 31 kernel needs 67, spills
   three of them when held to 64, and loses 4 %: it keeps its seven waves.) */
template <int W, bool CANON, int MODE, bool ASCII, bool SK>
__global__ void __launch_bounds__(256, (W == 2 && SK) ? 8 : 1)
fast_lookup_kernel(const dict_view d, const void* __restrict__ queries, const uint64_t n, const bool check_rc,
                   const result_view out, uint8_t* __restrict__ member, const pass_queues q,
                   const uint8_t* __restrict__ lane_valid /* null, or bit 0 of entry i: place i holds a query */) {
    /* LDS: the ASCII tile (256 * k characters) and, afterwards, the staged bucket lines (64 * W bytes per lane) */
    constexpr uint32_t TILE_WORDS = ASCII ? 64 * (W == 1 ? 31 : 63) + 24 : 0;  // (+ slack: the packing reads whole groups of 16 characters)
    constexpr uint32_t STAGE_WORDS = SK ? 256 * 16 : 0;
    __shared__ uint4 lds[(TILE_WORDS > STAGE_WORDS ? TILE_WORDS : STAGE_WORDS) / 4 + 1];
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool active = i < n && (!lane_valid || (lane_valid[i] & 1));
    kmer_w<W> x = kmer_zero<W>();
    if constexpr (ASCII) {
        /* util::string_to_uint_kmer (include/util.hpp:207-213) for a whole workgroup: the 256*k
           characters of this workgroup's queries are one contiguous run of the input; it is staged in
           LDS with 16-byte loads and every lane then packs its own k characters, four at a time */
        uint32_t* tile = reinterpret_cast<uint32_t*>(lds);
        const uint64_t first = uint64_t(blockIdx.x) * blockDim.x;
        const uint32_t count = uint32_t(n - first < blockDim.x ? n - first : blockDim.x);
        const uint32_t bytes = count * d.k;
        const char* src = static_cast<const char*>(queries) + first * d.k;
        if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            for (uint32_t c = threadIdx.x; c * 16 < bytes; c += blockDim.x) {
                if (c * 16 + 16 <= bytes) {
                    reinterpret_cast<uint4*>(tile)[c] = sk_load_piece(src + 16 * c);  // nontemporal: read once
                } else {
                    for (uint32_t b = c * 16; b < bytes; ++b) reinterpret_cast<char*>(tile)[b] = src[b];
                }
            }
        } else {
            for (uint32_t b = threadIdx.x; b < bytes; b += blockDim.x) reinterpret_cast<char*>(tile)[b] = src[b];
        }
        __syncthreads();
        if (active) {
            /* sixteen characters -> one 32-bit word of 2-bit codes: four characters at a time are cut out of the tile (alignbyte), turned into
               four 2-bit codes in four bytes ((c >> 1) & 3) and gathered into one byte by a multiplication (the four partial products
               b0 << 24, b1 << 26, b2 << 28, b3 << 30 do not overlap); 19 instructions per 16 characters (rounds 1-3: 80). What lies
               beyond the k-th character (the next query's, or the tile's slack) is cut off at the end. */
            const uint32_t start = threadIdx.x * d.k, w0 = start >> 2, sh = start & 3;
            uint32_t acc[2 * W];
#pragma unroll
            for (int t = 0; t < 2 * W; ++t) {
                acc[t] = 0;
                if (16u * uint32_t(t) < d.k) {  // uniform
                    uint32_t prev = tile[w0 + 4 * t];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t next = tile[w0 + 4 * t + j + 1];
                        const uint32_t four = __builtin_amdgcn_alignbyte(next, prev, sh);
                        prev = next;
                        const uint32_t c = (((four >> 1) & 0x03030303u) * 0x01041040u) >> 24;
                        acc[t] |= c << (8 * j);
                    }
                }
            }
            x.w[0] = uint64_t(acc[0]) | (uint64_t(acc[1]) << 32);
            if constexpr (W == 2) x.w[1] = uint64_t(acc[2]) | (uint64_t(acc[3]) << 32);
            x = kmer_take_chars<W>(x, d.k);
        }
        if constexpr (SK) __syncthreads();  // the tile's LDS is reused for the bucket lines
    } else {
        if (active) x = load_query<W, false>(queries, i, d.k);
    }
    fast_t r;
    const uint32_t shard = blockIdx.x & (DEFER_SHARDS - 1);  // the queues are sharded by workgroup: one hot counter would
                                                             // serialise at ~90 atomics/us
    if constexpr (SK) {
        /* k <= 31: a probe that needs more than its key's first bucket is finished inside the wave (sk_finish_in_wave). k <= 63: it joins
           the resume pass -- finishing in the wave there measured 27.1 -> 25.6 G lookups/s (profiles/r04/inwave_k63_ab.txt: the loop's state
           takes the kernel from 60 to 86 registers, 8 to 5 waves per SIMD, and a pass that runs at the random-line rate wants the waves) */
        if constexpr (W == 1)
            r = sk_lookup_in_wave<W>(d, x, active, CANON || check_rc, (!CANON && check_rc) ? int8_t(-1) : int8_t(1), lds + (threadIdx.x >> 6) * (64 * 4));
        else
            r = sk_first_pass_wave<W>(d, x, active, CANON || check_rc, (!CANON && check_rc) ? int8_t(-1) : int8_t(1),
                                      lds + (threadIdx.x >> 6) * (64 * 4));
        /* whatever needs a second dependent read joins the compacted second pass, with its packed k-mer (no lane leaves
           before this: the queue places are handed out per wave) */
        bool resume = active && r.outcome == FAST_CONTINUE;
        const uint32_t place = wave_queue_place(resume, q.resume_counts + shard);
        if (resume) {
            if (place < q.resume_capacity) {
                const uint64_t at = uint64_t(shard) * q.resume_capacity + place;
                q.resume_index[at] = uint32_t(i) | uint32_t(r.kmer_offset);
                for (int j = 0; j < W; ++j) q.resume_kmers[at * W + j] = x.w[j];
            } else {
                r.outcome = FAST_DEFER;  // queue full (more than half of the batch resumes): the complete path takes it
            }
        }
        if (!active) return;
    } else {
        /* (32-byte units fetched by pairs of lanes: pair_load32 posts a unit as a 32-bit number, enough for 2^37 bases of strings --
           uniform; past that every lane reads its own units: ADVICE r4) */
        bool by_pairs = false;
        if constexpr (W == 1 && !CANON) by_pairs = (d.num_bases >> 37) == 0;
        if (by_pairs) {
            if constexpr (W == 1 && !CANON) r = fast_lookup_pairs(d, x, active, check_rc);
        } else {
            r = active ? fast_lookup_one<W, CANON>(d, x, check_rc) : fast_unsettled(false);
        }
        /* a MIDLOAD bucket whose first position did not settle the query: the rest of it is scanned by whole waves
           (scan_lookup_kernel), not by this lane while its 63 neighbours wait */
        const bool scan = active && r.outcome == FAST_SCAN;
        const uint32_t place = wave_queue_place(scan, q.resume_counts + shard);
        if (scan) {
            if (place < q.resume_capacity) {
                const uint64_t at = uint64_t(shard) * q.resume_capacity + place;
                q.resume_index[at] = uint32_t(i);
                q.resume_meta[at] = r.kmer_offset;
                for (int j = 0; j < W; ++j) q.resume_kmers[at * W + j] = x.w[j];
            } else {
                r.outcome = FAST_DEFER;
            }
        }
        if (!active) return;
    }
    /* every remaining lane stores first (a deferred or resumed lane's value is a placeholder that a later pass
       overwrites), the deferred-queue push comes last */
    if constexpr (MODE == int(out_mode::member)) {
        /* (This instance allocates exactly 64 VGPRs and hipcc keeps a 64-bit shift's amount in v63, the last of them: on this chip
           such a shift is wrong in 6-7 % of its executions -- the "0.15 % of the indexed k-mers absent, differently from launch to
           launch" of rounds 2 and 3, which an asm keep-alive of three dead registers used to hide by moving the allocation to 72.
           tools/isa_guard.py now pads such kernels when the library is built; HISTORY.md, tools/debug/vgpr64_check.hip.) */
        __builtin_nontemporal_store(uint8_t(r.outcome == FAST_HIT ? 1 : 0), member + i);
    } else {
        hit_t h;
        h.kmer_offset = r.kmer_offset;
        h.string_id = r.string_id;
        h.orientation = r.orientation;
        h.found = r.outcome == FAST_HIT;
        h.minimizer_found = true;  // a hit has it; a miss gets the real one from the last pass when the caller asked for it (pass_queues)
        store_result<MODE == int(out_mode::full)>(d, out, i, h);
    }
    if (r.outcome == FAST_DEFER) q.defer_index[uint64_t(shard) * q.defer_capacity + atomicAdd(q.defer_counts + shard, 1u)] = uint32_t(i);
    else if (MODE == int(out_mode::full) && SK && q.flag_misses && r.outcome == FAST_MISS)
        q.defer_index[uint64_t(shard) * q.defer_capacity + atomicAdd(q.defer_counts + shard, 1u)] = uint32_t(i) | DEFER_FLAG_ONLY;
}

/* Second pass of the table lookup: the queries the first pass could not settle with one bucket read (their key's
   first bucket was full when the key was placed, or the key is heavy: marker + one slot per k-mer), compacted, so that these
   dependent reads are issued by full waves instead of by the few lanes of every first-pass wave that need them.
   RESUME_PARTS workgroups walk one shard of the queue. */
constexpr uint32_t RESUME_PARTS = 4;

template <int W, bool CANON, int MODE>
__global__ void __launch_bounds__(256)
resume_lookup_kernel(const dict_view d, const bool check_rc, const result_view out, uint8_t* __restrict__ member, const pass_queues q) {
    __shared__ uint4 lds[256 * 4];
    const uint32_t shard = blockIdx.x & (DEFER_SHARDS - 1), part = blockIdx.x / DEFER_SHARDS;
    const uint32_t pushed = q.resume_counts[shard];
    const uint32_t total = pushed < q.resume_capacity ? pushed : q.resume_capacity;
    for (uint32_t base = part * blockDim.x; base < total; base += RESUME_PARTS * blockDim.x) {  // uniform over the workgroup
        const uint32_t j = base + threadIdx.x;
        const bool active = j < total;
        const uint64_t at = uint64_t(shard) * q.resume_capacity + (active ? j : 0u);
        const uint32_t entry = q.resume_index[at];
        const uint64_t i = entry & ((1u << RESUME_CHOICE_SHIFT) - 1);  // < 2^27: launch_piece_queries()
        kmer_w<W> x;
        for (int t = 0; t < W; ++t) x.w[t] = q.resume_kmers[at * W + t];
        const fast_t r = sk_second_pass_wave<W>(d, x, active, entry, CANON || check_rc, (!CANON && check_rc) ? int8_t(-1) : int8_t(1),
                                                lds + (threadIdx.x >> 6) * (64 * 4));
        if (!active) continue;
        if (r.outcome == FAST_DEFER) {
            q.defer_index[uint64_t(shard) * q.defer_capacity + atomicAdd(q.defer_counts + shard, 1u)] = uint32_t(i);
        } else if constexpr (MODE == int(out_mode::member)) {
            member[i] = r.outcome == FAST_HIT ? 1 : 0;
        } else {
            hit_t h;
            h.kmer_offset = r.kmer_offset;
            h.string_id = r.string_id;
            h.orientation = r.orientation;
            h.found = r.outcome == FAST_HIT;
            h.minimizer_found = true;
            store_result<MODE == int(out_mode::full)>(d, out, i, h);
            if (MODE == int(out_mode::full) && q.flag_misses && r.outcome == FAST_MISS)
                q.defer_index[uint64_t(shard) * q.defer_capacity + atomicAdd(q.defer_counts + shard, 1u)] = uint32_t(i) | DEFER_FLAG_ONLY;
        }
    }
}

/* Bucket-scan pass of a replica without the table (directory or MPHF path): the queries whose MIDLOAD bucket (2..64
   minimizer positions, include/sparse_and_skew_index.hpp:122-132) did not settle at its first position. What the reference
   does offset after offset (spectrum_preserving_string_set.hpp:41-44,68-70: copy the bucket's offsets, try one after the other)
   a wave does at once: the 64 entries of a wave are laid out in LDS (k-mer, where the bucket's offsets lie, how many are left),
   their candidates -- (entry, offset) pairs, a few hundred per wave -- are dealt to the lanes 64 at a time, every lane reads ITS
   candidate's offset out of mid_load and the window it points at, compares, and the one lane whose window equals the entry's k-mer
   (a k-mer occurs once in the strings) deposits the result in the entry's LDS place: two dependent reads for the whole bucket,
   all candidates in flight together, instead of a chain of `size` reads on one lane with 63 lanes waiting. */
template <int W, bool CANON, int MODE>
__global__ void __launch_bounds__(256)
scan_lookup_kernel(const dict_view d, const bool check_rc, const result_view out, uint8_t* __restrict__ member, const pass_queues q) {
    struct entry_t {
        uint64_t kmer[W];
        uint64_t meta;
        uint64_t offset;     // result: INVALID_U64 = not found (yet)
        uint32_t string_id;
        uint32_t incl;       // candidates of the wave's entries up to and including this one
        int32_t orientation;
        uint32_t pad;
    };
    __shared__ entry_t lds[256];
    entry_t* E = lds + (threadIdx.x & ~63u);  // this wave's 64 entries
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t shard = blockIdx.x & (DEFER_SHARDS - 1), part = blockIdx.x / DEFER_SHARDS;
    const uint32_t pushed = q.resume_counts[shard];
    const uint32_t total = pushed < q.resume_capacity ? pushed : q.resume_capacity;
    for (uint32_t base = part * blockDim.x; base < total; base += RESUME_PARTS * blockDim.x) {  // uniform over the workgroup
        const uint32_t j = base + threadIdx.x;
        const bool active = j < total;
        const uint64_t at = uint64_t(shard) * q.resume_capacity + (active ? j : 0u);
        const uint32_t i = q.resume_index[at];
        const uint64_t meta = active ? q.resume_meta[at] : 0;
        const bool rc_strand = (meta >> 44) & 1;
        kmer_w<W> x;
        for (int t = 0; t < W; ++t) x.w[t] = q.resume_kmers[at * W + t];
        /* the k-mer as the probing strand reads it (regular dictionaries probe the reverse complement on its own) */
        const kmer_w<W> y = (!CANON && rc_strand) ? kmer_revcomp<W>(x, d.k) : x;
        const uint32_t mine = active ? uint32_t((meta >> 32) & 63u) : 0u;  // offsets left to try
        uint32_t incl = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (int(lane) >= o) incl += up;
        }
        for (int t = 0; t < W; ++t) E[lane].kmer[t] = y.w[t];
        E[lane].meta = meta;
        E[lane].offset = INVALID_U64;
        E[lane].string_id = 0;
        E[lane].incl = incl;
        E[lane].orientation = 1;
        const uint32_t all = __shfl(incl, 63, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t c0 = 0; c0 < all; c0 += 64) {  // uniform over the wave
            const uint32_t t = c0 + lane;
            if (t < all) {
                /* the entry holding candidate t: the first whose inclusive count exceeds t */
                uint32_t lo = 0, hi = 63;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (E[mid].incl > t) hi = mid;
                    else lo = mid + 1;
                }
                const uint64_t em = E[lo].meta;
                const uint32_t left = uint32_t((em >> 32) & 63u), pos = uint32_t((em >> 38) & 63u);
                const uint32_t c = t - (E[lo].incl - left) + 1;  // 1 .. size - 1: position 0 was tried by the first pass
                const uint64_t p = packed_get(d.mid_load, (em & 0xFFFFFFFFull) + c, d.off_width);
                kmer_w<W> want;
                for (int w = 0; w < W; ++w) want.w[w] = E[lo].kmer[w];
                if constexpr (CANON) {
                    /* both alignments, both orientations (spectrum_preserving_string_set.hpp:237-275) */
                    const kmer_w<W> want_rc = kmer_revcomp<W>(want, d.k);
                    const uint32_t pos2 = d.k - d.m - pos;
                    const window_t<W> w1 = read_window<W>(d.granules, p >= pos ? p - pos : p, d.k);
                    const window_t<W> w2 = read_window<W>(d.granules, p >= pos2 ? p - pos2 : p, d.k);
                    const bool f1 = kmer_eq<W>(w1.kmer, want), b1 = kmer_eq<W>(w1.kmer, want_rc);
                    const bool f2 = kmer_eq<W>(w2.kmer, want), b2 = kmer_eq<W>(w2.kmer, want_rc);
                    if (p >= pos && (f1 || b1) && !w1.crosses) {
                        E[lo].offset = p - pos;
                        E[lo].string_id = w1.string_id;
                        E[lo].orientation = b1 ? -1 : 1;
                    } else if (p >= pos2 && (f2 || b2) && !w2.crosses) {
                        E[lo].offset = p - pos2;
                        E[lo].string_id = w2.string_id;
                        E[lo].orientation = b2 ? -1 : 1;
                    }
                } else {
                    if (p >= pos) {
                        const window_t<W> w = read_window<W>(d.granules, p - pos, d.k);
                        if (kmer_eq<W>(w.kmer, want) && !w.crosses) {
                            E[lo].offset = p - pos;
                            E[lo].string_id = w.string_id;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (active) {
            hit_t h;
            h.kmer_offset = E[lane].offset;
            h.string_id = E[lane].string_id;
            h.found = h.kmer_offset != INVALID_U64;
            h.minimizer_found = true;
            h.orientation = CANON ? int8_t(E[lane].orientation) : (rc_strand ? int8_t(-1) : int8_t(1));
            if (!h.found && !CANON && !rc_strand && check_rc) {
                /* the forward strand's bucket does not hold it: the reverse complement is still to be probed (src/dictionary.cpp:
                   70-75) -- rare enough (a positive on the other strand whose forward minimizer happens to be in the dictionary,
                   in a MIDLOAD bucket) for the complete path */
                q.defer_index[uint64_t(shard) * q.defer_capacity + atomicAdd(q.defer_counts + shard, 1u)] = i;
            } else if constexpr (MODE == int(out_mode::member)) {
                member[i] = h.found ? 1 : 0;
            } else {
                store_result<MODE == int(out_mode::full)>(d, out, i, h);
            }
        }
        __builtin_amdgcn_wave_barrier();  // the entries are rewritten by the next turn
    }
}

/* Last pass: the deferred queries, compacted, through the complete lookup. */
template <int W, bool CANON, int MODE, bool ASCII>
__global__ void __launch_bounds__(256)
deferred_lookup_kernel(const dict_view d, const skew_part_dev* __restrict__ skew, const void* __restrict__ queries,
                       const bool check_rc, const result_view out, uint8_t* __restrict__ member, const pass_queues q) {
    /* DEFER_BLOCKS_PER_SHARD workgroups share a shard: the deferred queries are the ones with the longest
       dependent chains, so they get as many lanes as there are entries rather than a few busy ones */
    const uint32_t shard = blockIdx.x & (DEFER_SHARDS - 1), part = blockIdx.x / DEFER_SHARDS;
    const uint32_t total = q.defer_counts[shard];
    const uint32_t* mine = q.defer_index + uint64_t(shard) * q.defer_capacity;
    for (uint32_t j = part * blockDim.x + threadIdx.x; j < total; j += DEFER_BLOCKS_PER_SHARD * blockDim.x) {
        const uint32_t entry = mine[j];
        const uint64_t i = entry & ~DEFER_FLAG_ONLY;
        const kmer_w<W> x = load_query<W, ASCII>(queries, i, d.k);
        if constexpr (MODE == int(out_mode::full)) {
            if (q.flag_misses) {
                /* the MPHF path without the directory: it alone reproduces the flag (launch()) */
                if (entry & DEFER_FLAG_ONLY) {  // everything else is stored already
                    out.minimizer_found[i] = minimizer_found_of_a_miss<W, CANON>(d, skew, x, check_rc) ? 1 : 0;
                } else {
                    store_result<true>(d, out, i, lookup_one<W, CANON, false>(d, skew, x, check_rc));
                }
                continue;
            }
        }
        const hit_t h = lookup_one<W, CANON, true>(d, skew, x, check_rc);
        if constexpr (MODE == int(out_mode::member)) {
            member[i] = h.found ? 1 : 0;
        } else {
            store_result<MODE == int(out_mode::full)>(d, out, i, h);
        }
    }
}

static result_view advance(result_view v, uint64_t at) {
    if (v.kmer_id) v.kmer_id += at;
    if (v.kmer_id_in_string) v.kmer_id_in_string += at;
    if (v.kmer_offset) v.kmer_offset += at;
    if (v.string_id) v.string_id += at;
    if (v.string_begin) v.string_begin += at;
    if (v.string_end) v.string_end += at;
    if (v.kmer_orientation) v.kmer_orientation += at;
    if (v.minimizer_found) v.minimizer_found += at;
    return v;
}

/* queries per launch sequence (first, resume, deferred): at most 2^27 (queue entries carry 27-bit indices); the tests cut a small batch
   into several (hooks.hpp) */
static uint64_t launch_piece_queries() { return test_hook_u64("piece", uint64_t(1) << 27, 4096, uint64_t(1) << 27); }

/* the resume queue holds half a launch piece; a query that finds it full takes the complete path instead (tests: a divisor of 64
   makes nearly every resumed query overflow, tests/test_gpu_parity.py) */
static uint32_t resume_capacity_divisor() { return uint32_t(test_hook_u64("resume_divisor", 2, 1, 4096)); }

/* Tail passes on an auxiliary stream (launch()): when the deferred pass is the only tail pass (k <= 31 with the table: the rest is
   finished in the wave) and the batch is cut into several launch sequences anyway. Measured on the calibrated stand-ins, same box,
   alternating runs (HISTORY.md): with a resume pass overlapping LOSES (C3 35.9 -> 35.6, C2 33.5-34.6 -> 31.6-32.9: that tail is more
   random line fetches for a memory system the first pass already saturates); deferred pass only: C3 40.4-40.5 -> 40.8-40.9 (+1.0 %);
   cutting a batch into pieces for the sake of it: C2 37.8 -> 36.9, hence no forcing. */
template <int W, bool CANON, int MODE, bool ASCII>
static void launch(device_replica const* rep, void const* q, uint64_t n, bool check_rc,
                   result_view const& out, uint8_t* member, hipStream_t stream, uint8_t const* lane_valid) {
    dict_view const& d = rep->view;
    skew_part_dev const* skew = rep->d_skew;
    const uint32_t block = 256;
    /* Multi-pass lookup through the table (or the minimizer directory). A caller that wants `minimizer_found` still gets
       everything from the table except the flag of the misses: for an absent minimizer the reference's flag depends on which
       (arbitrary) bucket the MPHF lands on, so only the MPHF path can reproduce it (device_layout.hpp (4)) -- the last pass
       computes that one byte for them. Without a table such a caller gets the MPHF kernel for everything. */
    {
        const bool wants_flag = MODE == int(out_mode::full) && out.minimizer_found;
        if (!(wants_flag && !d.sk.enabled)) {
            /* multi-pass: at most 2^27 queries per launch sequence (queue entries are 32-bit; the scratch
               queues stay below 1.3 GiB (2.1 for 128-bit k-mers) per set). When a batch takes several sequences and the deferred pass
               is the only tail pass, the tail (few, dependent, latency-bound reads) of piece i runs on an auxiliary stream while the
               caller's stream already runs the first pass of piece i + 1 -- two sets of queues, used in turn; events order
               first(i) -> tail(i) -> first(i + 2). The caller's stream waits for the last tail before the call returns control of it. */
            const uint64_t piece_max = launch_piece_queries();
            uint64_t pieces = (n + piece_max - 1) / piece_max;
            const bool in_wave = W == 1 && d.sk.enabled;
            const uint64_t piece = ((n + pieces - 1) / pieces + block - 1) / block * block;  // equal pieces, whole workgroups
            pieces = (n + piece - 1) / piece;
            const bool overlap = pieces > 1 && in_wave;
            const size_t qbytes = size_t(W) * 8, kbytes = d.k;
            const uint32_t nblocks_max = uint32_t((std::min(piece, n) + block - 1) / block);
            pass_queues shape{};
            shape.defer_capacity = ((nblocks_max + DEFER_SHARDS - 1) / DEFER_SHARDS) * block;
            shape.resume_capacity = (shape.defer_capacity + 1) / resume_capacity_divisor();  // (table: resumed probes; no table: bucket scans)
            const uint64_t defer_places = uint64_t(DEFER_SHARDS) * shape.defer_capacity, resume_places = uint64_t(DEFER_SHARDS) * shape.resume_capacity;
            const size_t set_bytes =
                (2 * DEFER_SHARDS * sizeof(uint32_t) + (defer_places + resume_places) * sizeof(uint32_t) + resume_places * (W + 1) * sizeof(uint64_t) + 255) & ~size_t(255);
            std::lock_guard<std::mutex> enqueue(rep->launch_mutex);
            device_replica::stream_scratch& sc = rep->scratch_for(stream, (overlap ? 2 : 1) * set_bytes, overlap);
            uint64_t index = 0;
            for (uint64_t at = 0; at < n; at += piece, ++index) {
                const uint64_t m = std::min(piece, n - at);
                const uint32_t nblocks = uint32_t((m + block - 1) / block);
                const uint32_t set = overlap ? uint32_t(index & 1) : 0u;
                pass_queues pq = shape;
                pq.flag_misses = wants_flag ? 1u : 0u;
                char* scratch = static_cast<char*>(sc.block) + set * set_bytes;
                if (overlap && index >= 2) HIP_CHECK(hipStreamWaitEvent(stream, sc.tail_done[set], 0));  // the set's queues are free again
                HIP_CHECK(hipMemsetAsync(scratch, 0, 2 * DEFER_SHARDS * sizeof(uint32_t), stream));
                pq.defer_counts = reinterpret_cast<uint32_t*>(scratch);
                pq.resume_counts = pq.defer_counts + DEFER_SHARDS;
                pq.resume_kmers = reinterpret_cast<uint64_t*>(pq.resume_counts + DEFER_SHARDS);  // 8-byte aligned: 2 * 2048 * 4 bytes in
                pq.resume_meta = pq.resume_kmers + resume_places * W;
                pq.defer_index = reinterpret_cast<uint32_t*>(pq.resume_meta + resume_places);
                pq.resume_index = pq.defer_index + defer_places;
                const void* qa = static_cast<const char*>(q) + at * (ASCII ? kbytes : qbytes);
                const result_view ids = advance(out, at);
                uint8_t* mem = member ? member + at : nullptr;
                hipStream_t tail = stream;
                if (d.sk.enabled) {
                    hipLaunchKernelGGL((fast_lookup_kernel<W, CANON, MODE, ASCII, true>), dim3(nblocks), dim3(block), 0, stream, d, qa,
                                       m, check_rc, ids, mem, pq, lane_valid ? lane_valid + at : nullptr);
                } else {
                    hipLaunchKernelGGL((fast_lookup_kernel<W, CANON, MODE, ASCII, false>), dim3(nblocks), dim3(block), 0, stream, d, qa,
                                       m, check_rc, ids, mem, pq, lane_valid ? lane_valid + at : nullptr);
                }
                if (overlap) {
                    HIP_CHECK(hipEventRecord(sc.first_done[set], stream));
                    HIP_CHECK(hipStreamWaitEvent(sc.aux, sc.first_done[set], 0));
                    tail = sc.aux;
                }
                if constexpr (W == 2) {  // (k <= 31 finishes every probe in the first pass's own waves)
                    if (d.sk.enabled)
                        hipLaunchKernelGGL((resume_lookup_kernel<W, CANON, MODE>), dim3(DEFER_SHARDS * RESUME_PARTS), dim3(block), 0, tail, d,
                                           check_rc, ids, mem, pq);
                }
                if (!d.sk.enabled)
                    hipLaunchKernelGGL((scan_lookup_kernel<W, CANON, MODE>), dim3(DEFER_SHARDS * RESUME_PARTS), dim3(block), 0, tail, d,
                                       check_rc, ids, mem, pq);
                hipLaunchKernelGGL((deferred_lookup_kernel<W, CANON, MODE, ASCII>), dim3(DEFER_SHARDS * DEFER_BLOCKS_PER_SHARD), dim3(block), 0, tail, d,
                                   skew, qa, check_rc, ids, mem, pq);
                if (overlap) HIP_CHECK(hipEventRecord(sc.tail_done[set], sc.aux));
                HIP_CHECK(hipGetLastError());
            }
            if (overlap) HIP_CHECK(hipStreamWaitEvent(stream, sc.tail_done[(index - 1) & 1], 0));  // (aux is in order: the last tail is the last of all)
            return;
        }
    }
    uint64_t blocks = (n + block - 1) / block;
    if (blocks > (uint64_t(1) << 22)) blocks = uint64_t(1) << 22;  // grid-stride beyond 2^30 queries
    hipLaunchKernelGGL((lookup_kernel<W, CANON, MODE, ASCII>), dim3(uint32_t(blocks)), dim3(block), 0, stream, d, skew,
                       q, n, check_rc, out, member, lane_valid);
    HIP_CHECK(hipGetLastError());
}

template <int W, bool CANON, bool ASCII>
static void launch_mode(out_mode mode, device_replica const* rep, void const* q, uint64_t n,
                        bool check_rc, result_view const& out, uint8_t* member, hipStream_t s, uint8_t const* lane_valid) {
    switch (mode) {
        case out_mode::ids: launch<W, CANON, 0, ASCII>(rep, q, n, check_rc, out, member, s, lane_valid); break;
        case out_mode::full: launch<W, CANON, 1, ASCII>(rep, q, n, check_rc, out, member, s, lane_valid); break;
        case out_mode::member: launch<W, CANON, 2, ASCII>(rep, q, n, check_rc, out, member, s, lane_valid); break;
    }
}

template <bool ASCII>
static void launch_any(out_mode mode, device_replica const* rep, void const* q, uint64_t n,
                       bool check_rc, result_view const& out, uint8_t* member, hipStream_t s, uint8_t const* lane_valid = nullptr) {
    dict_view const& d = rep->view;
    const bool wide = d.k > 31;
    if (!wide && !d.canonical) launch_mode<1, false, ASCII>(mode, rep, q, n, check_rc, out, member, s, lane_valid);
    else if (!wide && d.canonical) launch_mode<1, true, ASCII>(mode, rep, q, n, check_rc, out, member, s, lane_valid);
    else if (wide && !d.canonical) launch_mode<2, false, ASCII>(mode, rep, q, n, check_rc, out, member, s, lane_valid);
    else launch_mode<2, true, ASCII>(mode, rep, q, n, check_rc, out, member, s, lane_valid);
}


static void check_outputs(out_mode mode, result_view const& out, uint8_t* member) {
    if (mode == out_mode::member) {
        if (!member) throw error(error_kind::argument, "is_member output pointer is null");
    } else if (!out.kmer_id) {
        throw error(error_kind::argument, "kmer_id output pointer is null");
    }
}

void engine::lookup_packed_masked_device(int device, uint64_t const* d_kmers, uint8_t const* d_lane_valid, uint64_t n, bool check_rc,
                                         out_mode mode, result_view const& d_out, void* stream) const {
    device_replica const* rep = replica(device);
    check_outputs(mode, d_out, nullptr);
    if (n == 0) return;
    device_guard guard(device);
    launch_any<false>(mode, rep, d_kmers, n, check_rc, d_out, nullptr, hipStream_t(stream), d_lane_valid);
}

void engine::lookup_packed_device(int device, uint64_t const* d_kmers, uint64_t n, bool check_rc, out_mode mode,
                                  result_view const& d_out, uint8_t* d_member, void* stream) const {
    device_replica const* rep = replica(device);
    check_outputs(mode, d_out, d_member);
    if (n == 0) return;
    device_guard guard(device);
    launch_any<false>(mode, rep, d_kmers, n, check_rc, d_out, d_member, hipStream_t(stream));
}

void engine::lookup_ascii_device(int device, char const* d_kmers, uint64_t n, bool check_rc, out_mode mode,
                                 result_view const& d_out, uint8_t* d_member, void* stream) const {
    device_replica const* rep = replica(device);
    check_outputs(mode, d_out, d_member);
    if (n == 0) return;
    device_guard guard(device);
    launch_any<true>(mode, rep, d_kmers, n, check_rc, d_out, d_member, hipStream_t(stream));
}

/* ---- kmer_neighbours: expand every query into its 8 neighbours, then the ordinary batched lookup -------- */

template <int W>
__global__ void __launch_bounds__(256)
expand_neighbours_kernel(const uint64_t* __restrict__ kmers, const uint64_t n, const uint32_t k, uint64_t* __restrict__ out) {
    const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;  // one lane per neighbour: coalesced stores
    if (t >= 8 * n) return;
    const kmer_w<W> x = load_query<W, false>(kmers, t >> 3, k);
    const kmer_w<W> y = kmer_neighbour<W>(x, uint32_t(t & 7), k);
    for (int j = 0; j < W; ++j) out[t * W + j] = y.w[j];
}

void engine::neighbours_packed_device(int device, uint64_t const* d_kmers, uint64_t n, bool check_rc, out_mode mode,
                                      result_view const& d_out, void* stream) const {
    device_replica const* rep = replica(device);
    check_outputs(mode, d_out, nullptr);
    if (n == 0) return;
    device_guard guard(device);
    hipStream_t s = hipStream_t(stream);
    const uint32_t W = rep->view.k <= 31 ? 1 : 2;
    check_single_launch(8 * n, "kmer_neighbours");
    uint64_t* expanded = nullptr;
    expanded = static_cast<uint64_t*>(rep->stream_alloc(8 * n * W * sizeof(uint64_t), s));
    const dim3 grid(uint32_t((8 * n + 255) / 256)), block(256);
    if (W == 1) hipLaunchKernelGGL(expand_neighbours_kernel<1>, grid, block, 0, s, d_kmers, n, rep->view.k, expanded);
    else hipLaunchKernelGGL(expand_neighbours_kernel<2>, grid, block, 0, s, d_kmers, n, rep->view.k, expanded);
    HIP_CHECK(hipGetLastError());
    launch_any<false>(mode, rep, expanded, 8 * n, check_rc, d_out, nullptr, s);
    HIP_CHECK(hipFreeAsync(expanded, s));
}

void engine::neighbours_packed_host(uint64_t const* h_kmers, uint64_t n, bool check_rc, out_mode mode,
                                    result_view const& h_out) const {
    const uint32_t k = m_idx->k, W = m_idx->words_per_kmer();
    std::vector<uint64_t> expanded(8 * n * W);
    for (uint64_t i = 0; i < n; ++i) {
        for (uint32_t which = 0; which < 8; ++which) {
            if (W == 1) {
                kmer_w<1> x;
                x.w[0] = h_kmers[i];
                expanded[8 * i + which] = kmer_neighbour<1>(kmer_take_chars<1>(x, k), which, k).w[0];
            } else {
                kmer_w<2> x;
                x.w[0] = h_kmers[2 * i];
                x.w[1] = h_kmers[2 * i + 1];
                const kmer_w<2> y = kmer_neighbour<2>(kmer_take_chars<2>(x, k), which, k);
                expanded[2 * (8 * i + which)] = y.w[0];
                expanded[2 * (8 * i + which) + 1] = y.w[1];
            }
        }
    }
    lookup_packed_host(expanded.data(), 8 * n, check_rc, mode, h_out, nullptr);
}

void engine::string_neighbours_host(uint64_t const* h_string_ids, uint64_t n, bool check_rc, out_mode mode,
                                    result_view const& h_out) const {
    host_index const& idx = *m_idx;
    const uint32_t k = idx.k, W = idx.words_per_kmer();
    std::vector<uint64_t> expanded(8 * n * W);
    uint64_t first[2], last[2];
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t s = h_string_ids[i];
        if (s >= idx.num_strings) throw error(error_kind::argument, "string_id out of range");
        /* ids of the string's first and last k-mer (include/offsets.hpp:41-65 inverted) */
        const uint64_t first_id = idx.endpoints[s] - s * (k - 1), last_id = idx.endpoints[s + 1] - k - s * (k - 1);
        access_kmer_packed(idx, first_id, first);
        access_kmer_packed(idx, last_id, last);
        for (uint32_t which = 0; which < 8; ++which) {
            uint64_t const* from = which < 4 ? last : first;  // suffix of the string forward, prefix backward
            if (W == 1) {
                kmer_w<1> x;
                x.w[0] = from[0];
                expanded[8 * i + which] = kmer_neighbour<1>(x, which, k).w[0];
            } else {
                kmer_w<2> x;
                x.w[0] = from[0];
                x.w[1] = from[1];
                const kmer_w<2> y = kmer_neighbour<2>(x, which, k);
                expanded[2 * (8 * i + which)] = y.w[0];
                expanded[2 * (8 * i + which) + 1] = y.w[1];
            }
        }
    }
    lookup_packed_host(expanded.data(), 8 * n, check_rc, mode, h_out, nullptr);
}

/* ---- routing of queries to the owners of their minimizers (minimizer-sharded index) ------------ */

template <int W>
__global__ void __launch_bounds__(256)
route_kernel(const dict_view d, const uint64_t* __restrict__ kmers, const uint64_t n, const uint32_t num_shards,
             uint32_t* __restrict__ owner_fwd, uint32_t* __restrict__ owner_rc) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const kmer_w<W> x = load_query<W, false>(kmers, i, d.k);
    const kmer_w<W> x_rc = kmer_revcomp<W>(x, d.k);
    uint64_t f = compute_minimizer<W>(x, d.k, d.m, d.hash_magic).value;
    uint64_t r = compute_minimizer<W>(x_rc, d.k, d.m, d.hash_magic).value;
    if (d.canonical) f = r = (r < f ? r : f);  // src/dictionary.cpp:31-40: the smaller-valued minimizer is probed
    owner_fwd[i] = shard_of_minimizer(f, num_shards);
    owner_rc[i] = shard_of_minimizer(r, num_shards);
}

void engine::route_packed_device(int device, uint64_t const* d_kmers, uint64_t n, uint32_t num_shards,
                                 uint32_t* d_owner_fwd, uint32_t* d_owner_rc, void* stream) const {
    device_replica const* rep = replica(device);
    if (n == 0) return;
    check_single_launch(n, "route");
    device_guard guard(device);
    const dim3 grid(uint32_t((n + 255) / 256)), block(256);
    if (rep->view.k <= 31)
        hipLaunchKernelGGL(route_kernel<1>, grid, block, 0, hipStream_t(stream), rep->view, d_kmers, n, num_shards, d_owner_fwd, d_owner_rc);
    else
        hipLaunchKernelGGL(route_kernel<2>, grid, block, 0, hipStream_t(stream), rep->view, d_kmers, n, num_shards, d_owner_fwd, d_owner_rc);
    HIP_CHECK(hipGetLastError());
}

/* Bucketing of a batch by owner shard (sharded.py): one message per (query, distinct owner). Every workgroup
   counts its messages per shard in LDS, reserves one contiguous range per shard with a single global atomic
   and (SCATTER) writes its messages there: the packed k-mer and the index of its query. With SCATTER off the
   same reservations simply add up to the per-shard message counts. */
constexpr uint32_t ROUTE_MAX_SHARDS = 1024;

/* Place of this lane's message among those of its workgroup for the same owner: the lanes of a wave that name the same
   owner are ranked with a ballot and take ONE LDS atomic together (one per lane serialises on a handful of counters:
   with a single owner the kernel ran at 47 ps per query, slower than the lookup it feeds). Called by all 64 lanes. */
__device__ __forceinline__ uint32_t route_rank(uint32_t owner, bool has, uint32_t* local_count) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t rank = 0;
    uint64_t todo = __ballot(has);
    while (todo) {  // wave-uniform
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t o = uint32_t(__shfl(int(owner), leader, 64));
        const bool same = has && owner == o;
        const uint64_t mask = __ballot(same);
        uint32_t first = 0;
        if (int(lane) == leader) first = atomicAdd(&local_count[o], uint32_t(__popcll(mask)));
        first = uint32_t(__shfl(int(first), leader, 64));
        if (same) rank = first + uint32_t(__popcll(mask & ((uint64_t(1) << lane) - 1)));
        todo &= ~mask;
    }
    return rank;
}


/* One workgroup routes ROUTE_TILES tiles of 256 queries and makes ONE reservation per owner for all of them: a
   reservation per tile is 4 x 10^5 atomics on a single address when there is a single owner (~90 atomics/us: 4.3 of the
   kernel's 4.7 ms per 10^8 queries). The owners of a workgroup's queries wait in LDS between the count and the scatter. */
constexpr uint32_t ROUTE_TILES = 16;

template <int W, bool SCATTER, bool BY_KEY>
__global__ void __launch_bounds__(256)
route_bucket_kernel(const dict_view d, const uint64_t* __restrict__ kmers, const uint64_t n, const uint32_t num_shards,
                    const bool check_rc, unsigned long long* __restrict__ cursors, uint64_t* __restrict__ send,
                    uint32_t* __restrict__ slots, uint32_t* __restrict__ known_owners) {
    /* known_owners (optional, one word per query: forward owner | reverse-complement owner << 16): the counting launch
       leaves the owners it elected there and the scattering launch reads them back instead of electing them again --
       the election is what bounds the counting launch (0.51 ms per 10^8 queries), the scatter is bound by its bytes */
    __shared__ uint32_t local_count[ROUTE_MAX_SHARDS];
    __shared__ uint32_t local_fill[SCATTER ? ROUTE_MAX_SHARDS : 1];
    __shared__ unsigned long long base[ROUTE_MAX_SHARDS];
    __shared__ uint32_t owners[SCATTER ? 256 * ROUTE_TILES : 1];  // forward owner | reverse-complement owner << 16
    static_assert(ROUTE_MAX_SHARDS <= (1u << 16), "two owners share a word");
    for (uint32_t t = threadIdx.x; t < num_shards; t += blockDim.x) {
        local_count[t] = 0;
        if constexpr (SCATTER) local_fill[t] = 0;
    }
    __syncthreads();
    const uint64_t first = uint64_t(blockIdx.x) * (256 * ROUTE_TILES);
#pragma unroll 1
    for (uint32_t tile = 0; tile < ROUTE_TILES; ++tile) {  // uniform over the workgroup
        const uint64_t i = first + tile * 256 + threadIdx.x;
        const bool active = i < n;
        uint32_t owner_f = 0, owner_r = 0;
        if (SCATTER && known_owners) {  // uniform over the launch
            if (active) {
                const uint32_t both = __builtin_nontemporal_load(known_owners + i);
                owner_f = both & 0xFFFFu;
                owner_r = both >> 16;
            }
        } else if (active) {
            const kmer_w<W> x = load_query<W, false>(kmers, i, d.k);
            const kmer_w<W> x_rc = kmer_revcomp<W>(x, d.k);
            if constexpr (BY_KEY) {
                /* table shards: the owner of the k-mer's table key; a k-mer without a key (tie) can go to any
                   replica -- they all hold the complete path -- so it goes where its smaller strand hashes */
                const sk_key_t kk = sk_key<W>(x, x_rc, d.k, d.sk.m);
                owner_f = owner_r = sk_owner(kk.tie ? (kmer_less<W>(x_rc, x) ? x_rc.w[0] : x.w[0]) : kk.key, num_shards);
            } else {
                uint64_t f = compute_minimizer<W>(x, d.k, d.m, d.hash_magic).value;
                uint64_t r = compute_minimizer<W>(x_rc, d.k, d.m, d.hash_magic).value;
                if (d.canonical) f = r = (r < f ? r : f);
                if (!check_rc) r = f;
                owner_f = shard_of_minimizer(f, num_shards);
                owner_r = shard_of_minimizer(r, num_shards);
            }
            if (!SCATTER && known_owners) __builtin_nontemporal_store(owner_f | (owner_r << 16), known_owners + i);
        }
        if constexpr (SCATTER) owners[tile * 256 + threadIdx.x] = owner_f | (owner_r << 16);
        (void)route_rank(owner_f, active, local_count);
        (void)route_rank(owner_r, active && owner_r != owner_f, local_count);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < num_shards; t += blockDim.x) {
        const uint32_t c = local_count[t];
        base[t] = c ? atomicAdd(cursors + t, (unsigned long long)c) : 0ull;
    }
    if constexpr (SCATTER) {
        __syncthreads();
#pragma unroll 1
        for (uint32_t tile = 0; tile < ROUTE_TILES; ++tile) {
            const uint64_t i = first + tile * 256 + threadIdx.x;
            const bool active = i < n;
            const uint32_t both = owners[tile * 256 + threadIdx.x];
            const uint32_t owner_f = both & 0xFFFFu, owner_r = both >> 16;
            const uint32_t rank_f = route_rank(owner_f, active, local_fill);
            const uint32_t rank_r = route_rank(owner_r, active && owner_r != owner_f, local_fill);
            if (active) {
                const kmer_w<W> x = load_query<W, false>(kmers, i, d.k);
                const uint64_t at = base[owner_f] + rank_f;
                for (int j = 0; j < W; ++j) send[at * W + j] = x.w[j];
                slots[at] = uint32_t(i);
                if (owner_r != owner_f) {
                    const uint64_t at2 = base[owner_r] + rank_r;
                    for (int j = 0; j < W; ++j) send[at2 * W + j] = x.w[j];
                    slots[at2] = uint32_t(i);
                }
            }
        }
    }
}

void engine::route_bucket_device(int device, uint64_t const* d_kmers, uint64_t n, uint32_t num_shards, bool check_rc,
                                 bool by_table_key, uint64_t* d_cursors, uint64_t* d_send, uint32_t* d_slots, void* stream,
                                 uint32_t* d_known_owners) const {
    device_replica const* rep = replica(device);
    if (num_shards == 0 || num_shards > ROUTE_MAX_SHARDS) throw error(error_kind::argument, "num_shards must be in [1, 1024]");
    if (n >= (uint64_t(1) << 32)) throw error(error_kind::argument, "at most 2^32 - 1 queries per routed batch");
    if ((d_send == nullptr) != (d_slots == nullptr)) throw error(error_kind::argument, "send and slots go together");
    if (n == 0) return;
    device_guard guard(device);
    const dim3 grid(uint32_t((n + 256 * ROUTE_TILES - 1) / (256 * ROUTE_TILES))), block(256);
    auto* cursors = reinterpret_cast<unsigned long long*>(d_cursors);
    hipStream_t s = hipStream_t(stream);
    const bool wide = rep->view.k > 31, scatter = d_send != nullptr;
    auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, block, 0, s, rep->view, d_kmers, n, num_shards, check_rc, cursors, d_send, d_slots, d_known_owners); };
    if (by_table_key) {
        if (!wide && !scatter) go(route_bucket_kernel<1, false, true>);
        else if (!wide && scatter) go(route_bucket_kernel<1, true, true>);
        else if (wide && !scatter) go(route_bucket_kernel<2, false, true>);
        else go(route_bucket_kernel<2, true, true>);
    } else {
        if (!wide && !scatter) go(route_bucket_kernel<1, false, false>);
        else if (!wide && scatter) go(route_bucket_kernel<1, true, false>);
        else if (wide && !scatter) go(route_bucket_kernel<2, false, false>);
        else go(route_bucket_kernel<2, true, false>);
    }
    HIP_CHECK(hipGetLastError());
}

/* replies of the owners, aligned with `slots`: a reply that found its k-mer settles its query (two owners
   that both find it return the same id: a k-mer occurs once in the strings) */
template <bool EVERY>  // EVERY: one reply per query -- each reply IS its query's answer, found or not, and `out` needs no filling first
__global__ void __launch_bounds__(256)
route_combine_kernel(const uint64_t* __restrict__ replies, const uint32_t* __restrict__ slots, const uint64_t m,
                     uint64_t* __restrict__ out) {
    const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const uint64_t id = __builtin_nontemporal_load(replies + t);
    if (EVERY || id != INVALID_U64) out[__builtin_nontemporal_load(slots + t)] = id;
}

void engine::route_combine_device(int device, uint64_t const* d_replies, uint32_t const* d_slots, uint64_t m, uint64_t* d_out,
                                  void* stream, bool one_reply_per_query) const {
    (void)replica(device);
    if (m == 0) return;
    check_single_launch(m, "route_combine");
    device_guard guard(device);
    const dim3 grid(uint32_t((m + 255) / 256));
    if (one_reply_per_query)
        hipLaunchKernelGGL(route_combine_kernel<true>, grid, dim3(256), 0, hipStream_t(stream), d_replies, d_slots, m, d_out);
    else
        hipLaunchKernelGGL(route_combine_kernel<false>, grid, dim3(256), 0, hipStream_t(stream), d_replies, d_slots, m, d_out);
    HIP_CHECK(hipGetLastError());
}

/* ---- access(kmer_id) on the device: include/spectrum_preserving_string_set.hpp:114-118 with
        offsets::id_to_offset (include/offsets.hpp:41-65) as a binary search over the endpoints ---- */

template <int W>
__global__ void __launch_bounds__(256)
access_kernel(const dict_view d, const uint64_t* __restrict__ ids, const uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t id = ids[i];
    if (id >= d.num_kmers) {
        for (int j = 0; j < W; ++j) out[i * W + j] = INVALID_U64;
        return;
    }
    const uint64_t km1 = d.k - 1;
    uint64_t lo = 0, hi = d.num_strings - 1;  // largest s with endpoints[s] - s*(k-1) <= id
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (d.endpoints[mid] - mid * km1 <= id) lo = mid;
        else hi = mid - 1;
    }
    const window_t<W> w = read_window<W>(d.granules, id + lo * km1, d.k);
    for (int j = 0; j < W; ++j) out[i * W + j] = w.kmer.w[j];
}

void engine::access_packed_device(int device, uint64_t const* d_ids, uint64_t n, uint64_t* d_out, void* stream) const {
    device_replica const* rep = replica(device);
    if (n == 0) return;
    check_single_launch(n, "access");
    device_guard guard(device);
    const dim3 grid(uint32_t((n + 255) / 256)), block(256);
    if (rep->view.k <= 31) hipLaunchKernelGGL(access_kernel<1>, grid, block, 0, hipStream_t(stream), rep->view, d_ids, n, d_out);
    else hipLaunchKernelGGL(access_kernel<2>, grid, block, 0, hipStream_t(stream), rep->view, d_ids, n, d_out);
    HIP_CHECK(hipGetLastError());
}

/* ---- weight(kmer_id) on the device: include/weights.hpp:147-152, prev_leq as a binary search ---- */

__global__ void __launch_bounds__(256)
weight_kernel(const dict_view d, const uint64_t* __restrict__ ids, const uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t id = ids[i];
    if (id >= d.num_kmers) {
        out[i] = INVALID_U64;
        return;
    }
    uint64_t lo = 0, hi = d.num_weight_intervals - 1;  // largest interval start <= id
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (d.weight_starts[mid] <= id) lo = mid;
        else hi = mid - 1;
    }
    out[i] = d.weight_values[lo];
}

void engine::weight_device(int device, uint64_t const* d_ids, uint64_t n, uint64_t* d_out, void* stream) const {
    device_replica const* rep = replica(device);
    if (!rep->view.num_weight_intervals) throw error(error_kind::argument, "the dictionary does not store weights");
    if (n == 0) return;
    check_single_launch(n, "weight");
    device_guard guard(device);
    hipLaunchKernelGGL(weight_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, hipStream_t(stream), rep->view, d_ids, n, d_out);
    HIP_CHECK(hipGetLastError());
}

/* ---- host-buffer path -----------------------------------------------------------------------
   The caller's arrays are pageable host memory. Per replica the batch is cut into chunks that several
   *lanes* pull from a shared counter; a lane owns a HIP stream, a pinned staging block and a device block
   and runs copy-in -> H2D -> kernels -> D2H -> copy-out for one chunk at a time. Lanes overlap one another,
   so host copies, both PCIe directions and the kernels proceed concurrently without any lane having to be
   asynchronous inside. Lanes (streams, pinned and device memory) are pooled in the replica across calls. */

namespace {

constexpr uint64_t HOST_CHUNK = uint64_t(1) << 21;  // queries per lane round
constexpr uint32_t HOST_LANES_MAX = 8;

uint64_t align256(uint64_t x) { return (x + 255) & ~uint64_t(255); }

struct field_plan {  // where each output of a chunk lives inside the staging block
    uint64_t in_bytes = 0, out_bytes = 0;
    uint64_t at[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // sshash_results order; member shares slot 0
    bool wanted[8] = {false, false, false, false, false, false, false, false};
};

field_plan plan_fields(out_mode mode, result_view const& h_out, uint64_t chunk, uint64_t bytes_per_query) {
    field_plan p;
    p.in_bytes = align256(chunk * bytes_per_query);
    uint64_t at = 0;
    auto add = [&](int f, bool wanted, uint64_t width) {
        p.wanted[f] = wanted;
        if (!wanted) return;
        p.at[f] = at;
        at += align256(chunk * width);
    };
    if (mode == out_mode::member) {
        add(0, true, 1);
    } else {
        add(0, true, 8);
        const bool full = mode == out_mode::full;
        add(1, full && h_out.kmer_id_in_string, 8);
        add(2, full && h_out.kmer_offset, 8);
        add(3, full && h_out.string_id, 8);
        add(4, full && h_out.string_begin, 8);
        add(5, full && h_out.string_end, 8);
        add(6, full && h_out.kmer_orientation, 1);
        add(7, full && h_out.minimizer_found, 1);
    }
    p.out_bytes = at;
    return p;
}

}  // namespace

host_lane* device_replica::acquire_lane(size_t bytes) const {
    host_lane* lane = nullptr;
    {
        std::lock_guard<std::mutex> lock(lanes_mutex);
        if (!idle_lanes.empty()) {
            lane = idle_lanes.back();
            idle_lanes.pop_back();
        }
    }
    if (!lane) {
        lane = new host_lane();
        HIP_CHECK(hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking));
    }
    if (lane->capacity < bytes) {
        if (lane->pinned) HIP_CHECK(hipHostFree(lane->pinned));
        if (lane->device) HIP_CHECK(hipFree(lane->device));
        lane->pinned = lane->device = nullptr;
        lane->capacity = 0;
        /* with headroom: the batches of one query file differ by a few reads, and a lane re-allocated for every new maximum
           (page-locked memory: tens of milliseconds per block, times eight lanes) cost the end-to-end file query a third of
           its wall clock */
        const size_t want = (bytes + bytes / 4 + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
        HIP_CHECK(hipHostMalloc(&lane->pinned, want, hipHostMallocDefault));
        HIP_CHECK(hipMalloc(&lane->device, want));
        lane->capacity = want;
    }
    return lane;
}

void device_replica::release_lane(host_lane* lane) const {
    std::lock_guard<std::mutex> lock(lanes_mutex);
    idle_lanes.push_back(lane);
}

template <bool ASCII>
static void host_lookup(engine const& eng, std::vector<int> const& devs, void const* h_in, uint64_t bytes_per_query,
                        uint64_t n, bool check_rc, out_mode mode, result_view const& h_out, uint8_t* h_member) {
    check_outputs(mode, h_out, h_member);
    if (n == 0) return;
    if (devs.empty()) throw error(error_kind::no_device, "dictionary is not resident on any device (call sshash_to_device first)");
    const uint64_t G = devs.size();
    const uint64_t chunk = std::min<uint64_t>(test_hook_u64("host_chunk", HOST_CHUNK, 1024, uint64_t(1) << 26), n);
    const uint64_t max_lanes = test_hook_u64("host_lanes", HOST_LANES_MAX, 1, 64);
    const field_plan plan = plan_fields(mode, h_out, chunk, bytes_per_query);
    const uint64_t hw = std::max(1u, std::thread::hardware_concurrency());

    struct share {  // one per device: its slice of the batch, handed out chunk by chunk
        uint64_t lo = 0, hi = 0;
        std::atomic<uint64_t> next{0};
    };
    std::vector<share> shares(G);
    std::vector<std::pair<uint64_t, uint32_t>> lanes;  // (device index, lane index)
    for (uint64_t g = 0; g < G; ++g) {
        shares[g].lo = n * g / G;
        shares[g].hi = n * (g + 1) / G;
        shares[g].next = shares[g].lo;
        const uint64_t chunks = (shares[g].hi - shares[g].lo + chunk - 1) / chunk;
        const uint64_t want = std::min<uint64_t>({chunks, max_lanes, std::max<uint64_t>(1, hw / G)});
        for (uint32_t l = 0; l < want; ++l) lanes.emplace_back(g, l);
    }
    std::vector<std::exception_ptr> errors(lanes.size());

    /* Caller buffers that are page-locked already (hipHostMalloc / hipHostRegister; first and last byte checked) are
       copied from and to directly: the staging copies through this library's own pinned lanes are what bounds the
       pageable case (1.7 G lookups/s on 16 cores, far below the link) */
    auto pinned = [](void const* p, uint64_t bytes) {
        if (!p || bytes == 0) return true;
        for (void const* q : {p, static_cast<void const*>(static_cast<char const*>(p) + bytes - 1)}) {
            hipPointerAttribute_t a;
            if (hipPointerGetAttributes(&a, q) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            if (a.type != hipMemoryTypeHost) return false;
        }
        return true;
    };
    bool in_place = pinned(h_in, n * bytes_per_query);
    if (mode == out_mode::member) in_place = in_place && pinned(h_member, n);
    else
        in_place = in_place && pinned(h_out.kmer_id, n * 8) && (!plan.wanted[1] || pinned(h_out.kmer_id_in_string, n * 8)) &&
                   (!plan.wanted[2] || pinned(h_out.kmer_offset, n * 8)) && (!plan.wanted[3] || pinned(h_out.string_id, n * 8)) &&
                   (!plan.wanted[4] || pinned(h_out.string_begin, n * 8)) && (!plan.wanted[5] || pinned(h_out.string_end, n * 8)) &&
                   (!plan.wanted[6] || pinned(h_out.kmer_orientation, n)) && (!plan.wanted[7] || pinned(h_out.minimizer_found, n));

    /* Page-locked caller arrays that the devices can address (hipHostMalloc'ed, or registered with hipHostRegisterMapped): the KERNELS read
       the queries and write the results where they lie -- no copy engine, no staging block, no chunk. A copy pipeline moves a chunk in,
       runs the kernels, moves the results out, and what it achieved on this box was both directions together at the rate of one
       (47-52 GB/s, 2.9-3.2 G lookups/s: profiles/r05/host_buffer_entry_point_pcie_inclusive.txt); a kernel's coalesced query loads and
       id stores travel in both directions of the link at once by themselves, and the batch is one launch sequence per device, as on
       device buffers. (tests: host_staged_copies=1 keeps the copy pipeline below, which arrays that are page-locked but not
       mapped take anyway.) */
    if (in_place && !test_hook_u64("host_staged_copies", 0, 0, 1)) {
        struct mapped_share {
            void const* in = nullptr;
            result_view out{};
            uint8_t* member = nullptr;
        };
        std::vector<mapped_share> mapped(G);
        bool all_mapped = true;
        int before = 0;
        HIP_CHECK(hipGetDevice(&before));
        for (uint64_t g = 0; g < G && all_mapped; ++g) {
            HIP_CHECK(hipSetDevice(devs[g]));
            auto device_address = [&](auto* p) -> decltype(p) {
                if (!p) return nullptr;
                void* dptr = nullptr;
                if (hipHostGetDevicePointer(&dptr, const_cast<void*>(static_cast<void const*>(p)), 0) != hipSuccess || !dptr) {
                    (void)hipGetLastError();
                    all_mapped = false;
                    return nullptr;
                }
                return static_cast<decltype(p)>(dptr);
            };
            const uint64_t lo = shares[g].lo;
            mapped_share& ms = mapped[g];
            if (char const* in = device_address(static_cast<char const*>(h_in))) ms.in = in + lo * bytes_per_query;
            if (mode == out_mode::member) {
                if (uint8_t* q = device_address(h_member)) ms.member = q + lo;
            } else {
                auto place = [&](auto* host, bool wanted) -> decltype(host) {
                    if (!wanted) return nullptr;
                    auto* q = device_address(host);
                    return q ? q + lo : nullptr;
                };
                ms.out.kmer_id = place(h_out.kmer_id, true);
                ms.out.kmer_id_in_string = place(h_out.kmer_id_in_string, plan.wanted[1]);
                ms.out.kmer_offset = place(h_out.kmer_offset, plan.wanted[2]);
                ms.out.string_id = place(h_out.string_id, plan.wanted[3]);
                ms.out.string_begin = place(h_out.string_begin, plan.wanted[4]);
                ms.out.string_end = place(h_out.string_end, plan.wanted[5]);
                ms.out.kmer_orientation = place(h_out.kmer_orientation, plan.wanted[6]);
                ms.out.minimizer_found = place(h_out.minimizer_found, plan.wanted[7]);
            }
        }
        (void)hipSetDevice(before);
        if (all_mapped) {
            std::vector<std::exception_ptr> failed(G);
            auto run_device = [&](uint64_t g) {
                try {
                    const uint64_t m = shares[g].hi - shares[g].lo;
                    if (m == 0) return;
                    device_replica const* rep = eng.replica(devs[g]);
                    HIP_CHECK(hipSetDevice(devs[g]));
                    host_lane* lane = rep->acquire_lane(0);  // (for its stream)
                    struct give_back {
                        device_replica const* rep;
                        host_lane* lane;
                        ~give_back() { rep->release_lane(lane); }
                    } guard{rep, lane};
                    mapped_share const& ms = mapped[g];
                    if (ASCII) eng.lookup_ascii_device(devs[g], static_cast<char const*>(ms.in), m, check_rc, mode, ms.out, ms.member, lane->stream);
                    else eng.lookup_packed_device(devs[g], static_cast<uint64_t const*>(ms.in), m, check_rc, mode, ms.out, ms.member, lane->stream);
                    HIP_CHECK(hipStreamSynchronize(lane->stream));
                } catch (...) { failed[g] = std::current_exception(); }
            };
            if (G == 1) {
                run_device(0);
            } else {
                std::vector<std::thread> workers;
                for (uint64_t g = 0; g < G; ++g) workers.emplace_back(run_device, g);
                for (auto& w : workers) w.join();
            }
            (void)hipSetDevice(before);
            for (auto const& e : failed)
                if (e) std::rethrow_exception(e);
            return;
        }
    }

    auto run_lane = [&](size_t li) {
        try {
            const uint64_t g = lanes[li].first;
            device_replica const* rep = eng.replica(devs[g]);
            HIP_CHECK(hipSetDevice(devs[g]));
            host_lane* lane = rep->acquire_lane(plan.in_bytes + plan.out_bytes);
            struct give_back {
                device_replica const* rep;
                host_lane* lane;
                ~give_back() { rep->release_lane(lane); }
            } guard{rep, lane};
            hipStream_t s = lane->stream;
            char* hp = static_cast<char*>(lane->pinned);
            char* dp = static_cast<char*>(lane->device);
            char* h_out_block = hp + plan.in_bytes;
            char* d_out_block = dp + plan.in_bytes;
            result_view d_out{};
            uint8_t* d_member = nullptr;
            if (mode == out_mode::member) d_member = reinterpret_cast<uint8_t*>(d_out_block + plan.at[0]);
            else {
                d_out.kmer_id = reinterpret_cast<uint64_t*>(d_out_block + plan.at[0]);
                if (plan.wanted[1]) d_out.kmer_id_in_string = reinterpret_cast<uint64_t*>(d_out_block + plan.at[1]);
                if (plan.wanted[2]) d_out.kmer_offset = reinterpret_cast<uint64_t*>(d_out_block + plan.at[2]);
                if (plan.wanted[3]) d_out.string_id = reinterpret_cast<uint64_t*>(d_out_block + plan.at[3]);
                if (plan.wanted[4]) d_out.string_begin = reinterpret_cast<uint64_t*>(d_out_block + plan.at[4]);
                if (plan.wanted[5]) d_out.string_end = reinterpret_cast<uint64_t*>(d_out_block + plan.at[5]);
                if (plan.wanted[6]) d_out.kmer_orientation = reinterpret_cast<int8_t*>(d_out_block + plan.at[6]);
                if (plan.wanted[7]) d_out.minimizer_found = reinterpret_cast<uint8_t*>(d_out_block + plan.at[7]);
            }
            for (;;) {
                const uint64_t at = shares[g].next.fetch_add(chunk);
                if (at >= shares[g].hi) break;
                const uint64_t m = std::min(chunk, shares[g].hi - at);
                if (in_place) {
                    HIP_CHECK(hipMemcpyAsync(dp, static_cast<char const*>(h_in) + at * bytes_per_query, m * bytes_per_query, hipMemcpyHostToDevice, s));
                } else {
                    std::memcpy(hp, static_cast<char const*>(h_in) + at * bytes_per_query, m * bytes_per_query);
                    HIP_CHECK(hipMemcpyAsync(dp, hp, m * bytes_per_query, hipMemcpyHostToDevice, s));
                }
                if (ASCII) eng.lookup_ascii_device(devs[g], dp, m, check_rc, mode, d_out, d_member, s);
                else eng.lookup_packed_device(devs[g], reinterpret_cast<uint64_t const*>(dp), m, check_rc, mode, d_out, d_member, s);
                if (in_place) {
                    auto back = [&](void* dst, int f, uint64_t width) {
                        HIP_CHECK(hipMemcpyAsync(static_cast<char*>(dst) + at * width, d_out_block + plan.at[f], m * width, hipMemcpyDeviceToHost, s));
                    };
                    if (mode == out_mode::member) back(h_member, 0, 1);
                    else {
                        back(h_out.kmer_id, 0, 8);
                        if (plan.wanted[1]) back(h_out.kmer_id_in_string, 1, 8);
                        if (plan.wanted[2]) back(h_out.kmer_offset, 2, 8);
                        if (plan.wanted[3]) back(h_out.string_id, 3, 8);
                        if (plan.wanted[4]) back(h_out.string_begin, 4, 8);
                        if (plan.wanted[5]) back(h_out.string_end, 5, 8);
                        if (plan.wanted[6]) back(h_out.kmer_orientation, 6, 1);
                        if (plan.wanted[7]) back(h_out.minimizer_found, 7, 1);
                    }
                    HIP_CHECK(hipStreamSynchronize(s));
                    continue;
                }
                HIP_CHECK(hipMemcpyAsync(h_out_block, d_out_block, plan.out_bytes, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                if (mode == out_mode::member) std::memcpy(h_member + at, h_out_block + plan.at[0], m);
                else {
                    std::memcpy(h_out.kmer_id + at, h_out_block + plan.at[0], m * 8);
                    if (plan.wanted[1]) std::memcpy(h_out.kmer_id_in_string + at, h_out_block + plan.at[1], m * 8);
                    if (plan.wanted[2]) std::memcpy(h_out.kmer_offset + at, h_out_block + plan.at[2], m * 8);
                    if (plan.wanted[3]) std::memcpy(h_out.string_id + at, h_out_block + plan.at[3], m * 8);
                    if (plan.wanted[4]) std::memcpy(h_out.string_begin + at, h_out_block + plan.at[4], m * 8);
                    if (plan.wanted[5]) std::memcpy(h_out.string_end + at, h_out_block + plan.at[5], m * 8);
                    if (plan.wanted[6]) std::memcpy(h_out.kmer_orientation + at, h_out_block + plan.at[6], m);
                    if (plan.wanted[7]) std::memcpy(h_out.minimizer_found + at, h_out_block + plan.at[7], m);
                }
            }
        } catch (...) { errors[li] = std::current_exception(); }
    };

    int prev = 0;
    HIP_CHECK(hipGetDevice(&prev));
    if (lanes.size() == 1) {
        run_lane(0);  // small batch on one device: no thread at all
    } else {
        std::vector<std::thread> workers;
        for (size_t li = 0; li < lanes.size(); ++li) workers.emplace_back(run_lane, li);
        for (auto& w : workers) w.join();
    }
    (void)hipSetDevice(prev);
    for (auto const& e : errors)
        if (e) std::rethrow_exception(e);
}

void engine::lookup_packed_host(uint64_t const* h_kmers, uint64_t n, bool check_rc, out_mode mode,
                                result_view const& h_out, uint8_t* h_member) const {
    host_lookup<false>(*this, devices(), h_kmers, 8ull * m_idx->words_per_kmer(), n, check_rc, mode, h_out, h_member);
}

void engine::lookup_ascii_host(char const* h_kmers, uint64_t n, bool check_rc, out_mode mode, result_view const& h_out,
                               uint8_t* h_member) const {
    host_lookup<true>(*this, devices(), h_kmers, m_idx->k, n, check_rc, mode, h_out, h_member);
}

}  // namespace sshash_amd
