// mphf_build.hpp -- host-side construction of the PTHash-style MPHF described in mphf.hpp.
//
// Replaces what the reference obtains from `pthash::partitioned_phf::build_in_*_memory`
// (include/minimizers_control_map.hpp:6-34, src/builder/build_sparse_and_skew_index.cpp:347-356).
// Build parameters mirror the reference's: lambda = average bucket size (5, or 7 for the skew
// index), alpha = 0.94 load factor. Partitions are searched independently on a thread pool.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstring>
#include <stdexcept>
#include <mutex>
#include <exception>
#include <thread>
#include <vector>

#include "errors.hpp"
#include "mphf.hpp"

namespace sshash_amd {

struct mphf_host {
    uint64_t seed = 0;
    uint64_t num_keys = 0;
    uint32_t pilot_width = 1;
    std::vector<mphf_partition> parts;
    std::vector<uint64_t> pilots;  // packed + 1 padding word
    std::vector<uint32_t> free_slots;

    mphf_view view() const {
        mphf_view v;
        v.parts = parts.data();
        v.pilots = pilots.data();
        v.free_slots = free_slots.data();
        v.seed = seed;
        v.num_keys = num_keys;
        v.num_parts = uint32_t(parts.size());
        v.pilot_width = pilot_width;
        return v;
    }
    uint64_t num_bits() const {
        return 8 * (parts.size() * sizeof(mphf_partition) + pilots.size() * 8 + free_slots.size() * 4);
    }
};

struct mphf_build_config {
    double lambda = 5.0;
    double alpha = 0.94;
    uint64_t seed = 0;
    uint64_t avg_partition_size = uint64_t(1) << 20;
    uint32_t num_threads = 1;
    uint32_t max_pilot = 1u << 28;
};

inline void packed_set(std::vector<uint64_t>& data, uint64_t i, uint32_t w, uint64_t v) {
    const uint64_t bit = i * w;
    const uint64_t word = bit >> 6;
    const uint32_t sh = uint32_t(bit & 63);
    data[word] |= v << sh;
    if (sh + w > 64) data[word + 1] |= v >> (64 - sh);
}

inline uint32_t bits_for(uint64_t max_value) {  // width able to hold max_value, at least 1
    uint32_t w = 1;
    while (w < 64 && (max_value >> w) != 0) ++w;
    return w;
}

namespace detail {

struct duplicate_keys_error : error {
    duplicate_keys_error()
        : error(error_kind::build, "mphf: two keys are equal (a minimizer or k-mer occurs twice: the input is not a "
                                   "spectrum-preserving string set, every k-mer must occur once)") {}
};

struct part_build_result {
    std::vector<uint32_t> pilots;      // one per bucket
    std::vector<uint32_t> free_slots;  // table_size - num_keys entries
    uint32_t max_pilot = 0;
    bool ok = true;
    bool duplicate_keys = false;  // two keys with the same 128-bit hash: no seed can separate them
};

/* Search the pilots of one partition. `hashes` are the (h1,h2) of this partition's keys. */
inline part_build_result build_partition(std::vector<hash128> const& hashes,
                                         mphf_partition const& part, uint32_t max_pilot) {
    part_build_result res;
    const uint32_t n = part.num_keys;
    const uint32_t t = part.table_size;
    const uint32_t nb = part.dense_buckets + part.sparse_buckets;
    res.pilots.assign(nb, 0);
    res.free_slots.assign(t - n, 0);
    if (n == 0) return res;

    /* bucket -> its keys' h2 (counting sort by bucket id) */
    std::vector<uint32_t> bucket_begin(nb + 1, 0);
    std::vector<uint32_t> bucket_of(n);
    for (uint32_t i = 0; i < n; ++i) {
        bucket_of[i] = mphf_bucket_of(hashes[i], part.dense_buckets, part.sparse_buckets);
        ++bucket_begin[bucket_of[i] + 1];
    }
    for (uint32_t b = 0; b < nb; ++b) bucket_begin[b + 1] += bucket_begin[b];
    std::vector<uint64_t> h2(n);
    {
        std::vector<uint32_t> cursor(bucket_begin.begin(), bucket_begin.end() - 1);
        for (uint32_t i = 0; i < n; ++i) h2[cursor[bucket_of[i]]++] = hashes[i].second;
    }
    /* two keys of one bucket with equal h2 can never be separated: that is a repeated key (for the skew index, a
       k-mer that occurs twice in the input, which is then not a spectrum-preserving string set). Fail at once
       instead of searching 2^28 pilots under 16 seeds. */
    for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t begin = bucket_begin[b], size = bucket_begin[b + 1] - begin;
        if (size < 2) continue;
        std::sort(h2.begin() + begin, h2.begin() + begin + size);
        for (uint32_t j = 1; j < size; ++j) {
            if (h2[begin + j] == h2[begin + j - 1]) {
                res.ok = false;
                res.duplicate_keys = true;
                return res;
            }
        }
    }
    /* buckets in non-increasing size order (counting sort by size) */
    uint32_t max_size = 0;
    for (uint32_t b = 0; b < nb; ++b) max_size = std::max(max_size, bucket_begin[b + 1] - bucket_begin[b]);
    std::vector<uint32_t> order;
    order.reserve(nb);
    {
        std::vector<std::vector<uint32_t>> by_size(max_size + 1);
        for (uint32_t b = 0; b < nb; ++b) by_size[bucket_begin[b + 1] - bucket_begin[b]].push_back(b);
        for (uint32_t s = max_size; s >= 1; --s)
            for (uint32_t b : by_size[s]) order.push_back(b);
    }

    std::vector<uint64_t> taken((uint64_t(t) + 63) / 64, 0);
    std::vector<uint32_t> pos(max_size);
    for (uint32_t b : order) {
        const uint32_t begin = bucket_begin[b], size = bucket_begin[b + 1] - begin;
        uint32_t pilot = 0;
        for (;; ++pilot) {
            if (pilot > max_pilot) {
                res.ok = false;
                return res;
            }
            bool good = true;
            for (uint32_t j = 0; j < size; ++j) {
                const uint32_t p = mphf_position(h2[begin + j], pilot, t);
                if ((taken[p >> 6] >> (p & 63)) & 1) {
                    good = false;
                    break;
                }
                pos[j] = p;
            }
            if (!good) continue;
            if (size > 1) { /* in-bucket collisions */
                std::sort(pos.begin(), pos.begin() + size);
                for (uint32_t j = 1; j < size; ++j) {
                    if (pos[j] == pos[j - 1]) {
                        good = false;
                        break;
                    }
                }
                if (!good) continue;
            }
            break;
        }
        for (uint32_t j = 0; j < size; ++j) taken[pos[j] >> 6] |= uint64_t(1) << (pos[j] & 63);
        res.pilots[b] = pilot;
        res.max_pilot = std::max(res.max_pilot, pilot);
    }

    /* minimal output: every taken position p >= n is redirected to a free position < n */
    uint32_t next_free = 0;
    for (uint32_t p = n; p < t; ++p) {
        if ((taken[p >> 6] >> (p & 63)) & 1) {
            while ((taken[next_free >> 6] >> (next_free & 63)) & 1) ++next_free;
            res.free_slots[p - n] = next_free++;
        }
    }
    return res;
}

/* Workers are raw std::threads: an exception leaving one would terminate the process. The first one thrown is
   kept and rethrown on the calling thread once all workers have joined; the others stop at their next item. */
template <typename Fn>
inline void parallel_for(uint64_t n, uint32_t num_threads, Fn&& fn) {
    if (num_threads <= 1 || n <= 1) {
        for (uint64_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<uint64_t> next{0};
    std::atomic<bool> failed{false};
    std::exception_ptr first;
    std::mutex guard;
    std::vector<std::thread> pool;
    const uint32_t nt = uint32_t(std::min<uint64_t>(num_threads, n));
    for (uint32_t t = 0; t < nt; ++t) {
        pool.emplace_back([&] {
            try {
                for (;;) {
                    const uint64_t i = next.fetch_add(1);
                    if (i >= n || failed.load(std::memory_order_relaxed)) break;
                    fn(i);
                }
            } catch (...) {
                std::lock_guard<std::mutex> lock(guard);
                if (!first) first = std::current_exception();
                failed = true;
            }
        });
    }
    for (auto& th : pool) th.join();
    if (first) std::rethrow_exception(first);
}

/* [begin,end) ranges, one per thread */
template <typename Fn>
inline void parallel_ranges(uint64_t n, uint32_t num_threads, Fn&& fn) {
    const uint32_t nt = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(num_threads, n)));
    if (nt == 1) {
        fn(uint64_t(0), n, 0u);
        return;
    }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> thrown(nt);
    const uint64_t chunk = (n + nt - 1) / nt;
    for (uint32_t t = 0; t < nt; ++t) {
        const uint64_t b = std::min(n, t * chunk), e = std::min(n, b + chunk);
        pool.emplace_back([=, &fn, &thrown] {
            try {
                fn(b, e, t);
            } catch (...) { thrown[t] = std::current_exception(); }
        });
    }
    for (auto& th : pool) th.join();
    for (auto const& e : thrown)
        if (e) std::rethrow_exception(e);
}

}  // namespace detail

/* Build over `n` keys; `hash_of(i)` returns the 128-bit base hash of key i under `seed`.
   Throws error(build) when a partition cannot be completed (caller re-seeds). */
template <typename HashOf>
inline void mphf_build_once(mphf_host& f, uint64_t n, HashOf&& hash_of, mphf_build_config const& cfg) {
    f = mphf_host();
    f.seed = cfg.seed;
    f.num_keys = n;
    uint32_t P = uint32_t(std::max<uint64_t>(1, (n + cfg.avg_partition_size / 2) / cfg.avg_partition_size));
    f.parts.assign(P, mphf_partition{});

    /* pass 1: hash + partition sizes */
    std::vector<hash128> hashes(n);
    std::vector<uint32_t> part_of(n);
    {
        std::vector<std::vector<uint64_t>> counts(std::max(1u, cfg.num_threads), std::vector<uint64_t>(P, 0));
        detail::parallel_ranges(n, cfg.num_threads, [&](uint64_t b, uint64_t e, uint32_t t) {
            auto& c = counts[t];
            for (uint64_t i = b; i < e; ++i) {
                hashes[i] = hash_of(i);
                part_of[i] = mphf_partition_of(hashes[i], P);
                ++c[part_of[i]];
            }
        });
        for (uint32_t p = 0; p < P; ++p) {
            uint64_t c = 0;
            for (auto& v : counts) c += v[p];
            if (c >= (uint64_t(1) << 31)) throw error(error_kind::build, "mphf: partition too large");
            if (c == 0 && n != 0) throw error(error_kind::build, "mphf: empty partition");
            f.parts[p].num_keys = uint32_t(c);
        }
    }
    /* partition geometry + prefix sums */
    uint64_t key_off = 0, pilot_off = 0, free_off = 0;
    for (uint32_t p = 0; p < P; ++p) {
        auto& part = f.parts[p];
        const uint64_t np = part.num_keys;
        uint64_t t = uint64_t(double(np) / cfg.alpha);
        if (t < np) t = np;
        if (t == 0) t = 1;
        if ((t & (t - 1)) == 0 && t > 1) ++t;  // avoid powers of two
        uint64_t nb = uint64_t(double(np) / cfg.lambda) + 1;
        if (nb < 2) nb = 2;
        uint32_t dense = uint32_t(double(nb) * 0.3);
        if (dense == 0) dense = 1;
        part.table_size = uint32_t(t);
        part.dense_buckets = dense;
        part.sparse_buckets = uint32_t(nb) - dense;
        part.key_offset = key_off;
        part.pilot_base = pilot_off;
        part.free_base = free_off;
        key_off += np;
        pilot_off += nb;
        free_off += t - np;
    }
    /* scatter hashes per partition */
    std::vector<std::vector<hash128>> per_part(P);
    for (uint32_t p = 0; p < P; ++p) per_part[p].reserve(f.parts[p].num_keys);
    for (uint64_t i = 0; i < n; ++i) per_part[part_of[i]].push_back(hashes[i]);
    std::vector<hash128>().swap(hashes);
    std::vector<uint32_t>().swap(part_of);

    /* pass 2: pilot search, one partition per task */
    std::vector<detail::part_build_result> results(P);
    detail::parallel_for(P, cfg.num_threads, [&](uint64_t p) {
        results[p] = detail::build_partition(per_part[p], f.parts[p], cfg.max_pilot);
        std::vector<hash128>().swap(per_part[p]);
    });
    uint32_t max_pilot = 0;
    for (auto& r : results) {
        if (r.duplicate_keys) throw detail::duplicate_keys_error();
        if (!r.ok) throw error(error_kind::build, "mphf: pilot search failed");
        max_pilot = std::max(max_pilot, r.max_pilot);
    }
    f.pilot_width = bits_for(max_pilot);
    f.pilots.assign((pilot_off * f.pilot_width + 63) / 64 + 1, 0);
    f.free_slots.assign(free_off + 1, 0);
    for (uint32_t p = 0; p < P; ++p) {
        auto& r = results[p];
        for (uint64_t b = 0; b < r.pilots.size(); ++b)
            packed_set(f.pilots, f.parts[p].pilot_base + b, f.pilot_width, r.pilots[b]);
        if (!r.free_slots.empty())
            std::memcpy(f.free_slots.data() + f.parts[p].free_base, r.free_slots.data(), r.free_slots.size() * 4);
    }
}

/* Re-seeding wrapper: seeds cfg.seed, cfg.seed+1, ... until one works. */
template <typename MakeHashOf>
inline void mphf_build(mphf_host& f, uint64_t n, MakeHashOf&& make_hash_of, mphf_build_config cfg) {
    for (int attempt = 0; attempt < 16; ++attempt) {
        try {
            mphf_build_once(f, n, make_hash_of(cfg.seed), cfg);
            return;
        } catch (detail::duplicate_keys_error const&) {
            throw;  // another seed cannot help
        } catch (error const&) { cfg.seed += 0x9E3779B97F4A7C15ULL; }
    }
    throw error(error_kind::build, "mphf: construction failed after 16 seeds");
}

}  // namespace sshash_amd
