// index.cpp -- host-side dictionary construction and (de)serialisation. See index.hpp.
#include "index.hpp"

#include <zlib.h>

#include <algorithm>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <sstream>

namespace sshash_amd {

namespace {

struct mini_tuple {
    uint64_t minimizer;
    uint64_t rest;  // pos_in_seq << 16 | pos_in_kmer << 8 | num_kmers_in_super_kmer
    uint64_t pos() const { return rest >> 16; }
    uint32_t pos_in_kmer() const { return uint32_t((rest >> 8) & 0xFF); }
    uint32_t num_kmers() const { return uint32_t(rest & 0xFF); }
    bool operator<(mini_tuple const& o) const {
        return minimizer != o.minimizer ? minimizer < o.minimizer : rest < o.rest;
    }
};

inline uint32_t base_at(uint64_t const* words, uint64_t pos) {
    return uint32_t((words[pos >> 5] >> ((pos & 31) * 2)) & 3);
}

/* k bases starting at base offset `off` (util::read_kmer_at, include/util.hpp:248-257) */
template <int W>
inline kmer_w<W> read_kmer(uint64_t const* words, uint64_t off, uint32_t k) {
    const uint64_t word = off >> 5;
    const uint32_t sh = uint32_t(off & 31) * 2;
    kmer_w<W> x;
    for (int i = 0; i < W; ++i) {
        uint64_t v = words[word + i] >> sh;
        if (sh) v |= words[word + i + 1] << (64 - sh);
        x.w[i] = v;
    }
    return kmer_take_chars<W>(x, k);
}

inline void atomic_packed_set(uint64_t* data, uint64_t i, uint32_t w, uint64_t v) {
    const uint64_t bit = i * w;
    const uint64_t word = bit >> 6;
    const uint32_t sh = uint32_t(bit & 63);
    __atomic_fetch_or(&data[word], v << sh, __ATOMIC_RELAXED);
    if (sh + w > 64) __atomic_fetch_or(&data[word + 1], v >> (64 - sh), __ATOMIC_RELAXED);
}

/* chunk-sort + pairwise merges */
template <typename T>
void parallel_sort(std::vector<T>& v, uint32_t num_threads) {
    const uint64_t n = v.size();
    if (num_threads <= 1 || n < (1u << 16)) {
        std::sort(v.begin(), v.end());
        return;
    }
    uint32_t chunks = 1;
    while (chunks < num_threads) chunks <<= 1;
    std::vector<uint64_t> bounds(chunks + 1);
    for (uint32_t c = 0; c <= chunks; ++c) bounds[c] = n * c / chunks;
    detail::parallel_for(chunks, num_threads,
                         [&](uint64_t c) { std::sort(v.begin() + bounds[c], v.begin() + bounds[c + 1]); });
    for (uint32_t width = 1; width < chunks; width <<= 1) {
        const uint32_t pairs = chunks / (2 * width);
        detail::parallel_for(pairs, num_threads, [&](uint64_t p) {
            const uint64_t lo = bounds[p * 2 * width], mid = bounds[p * 2 * width + width],
                           hi = bounds[p * 2 * width + 2 * width];
            std::inplace_merge(v.begin() + lo, v.begin() + mid, v.begin() + hi);
        });
    }
}

struct timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap() {
        auto t1 = std::chrono::steady_clock::now();
        double s = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        return s;
    }
};

/* Super-k-mer tuples of the strings [s_begin, s_end): one tuple per maximal run of consecutive
   k-mers sharing (minimizer, minimizer position) -- src/builder/compute_minimizer_tuples.cpp:55-108.
   Sliding minimum with the reference's tie rules: forward strand leftmost (strict <,
   include/minimizer_iterator.hpp:46,77), reverse strand leftmost in the reverse-complemented
   k-mer == rightmost in forward coordinates (<=, include/minimizer_iterator.hpp:128,159);
   canonical picks the smaller minimizer VALUE, ties to forward (compute_minimizer_tuples.cpp:80). */
void tuples_of_strings(host_index const& idx, uint64_t s_begin, uint64_t s_end, std::vector<mini_tuple>& out) {
    const uint32_t k = idx.k, m = idx.m;
    const uint64_t magic = idx.hash_magic;
    const uint64_t mask = low_mask(2 * m);
    const uint32_t win = k - m + 1;  // m-mers per k-mer
    uint64_t const* words = idx.strings.data();
    std::vector<uint64_t> hf, hr, vf, vr;  // per m-mer position of the current string
    for (uint64_t s = s_begin; s < s_end; ++s) {
        const uint64_t begin = idx.endpoints[s], end = idx.endpoints[s + 1];
        const uint64_t len = end - begin;
        const uint64_t num_mmers = len - m + 1;
        hf.resize(num_mmers);
        vf.resize(num_mmers);
        if (idx.canonical) {
            hr.resize(num_mmers);
            vr.resize(num_mmers);
        }
        uint64_t fwd = 0, rc = 0;
        for (uint64_t i = 0; i < len; ++i) {
            const uint64_t c = base_at(words, begin + i);
            fwd = (fwd >> 2) | (c << (2 * (m - 1)));
            rc = ((rc << 2) | (c ^ 2)) & mask;
            if (i + 1 >= m) {
                const uint64_t p = i + 1 - m;
                vf[p] = fwd;
                hf[p] = mmer_hash(fwd, magic);
                if (idx.canonical) {
                    vr[p] = rc;
                    hr[p] = mmer_hash(rc, magic);
                }
            }
        }
        const uint64_t num_kmers = len - k + 1;
        uint64_t best_f = 0, best_r = 0;  // argmin positions (relative to string begin)
        bool have = false;
        uint64_t cur_min = 0, cur_pos = 0, cur_first_pik = 0, cur_count = 0;
        for (uint64_t j = 0; j < num_kmers; ++j) {
            /* forward: leftmost minimum over [j, j+win) */
            if (j == 0 || best_f < j) {
                best_f = j;
                for (uint64_t p = j + 1; p < j + win; ++p)
                    if (hf[p] < hf[best_f]) best_f = p;
            } else if (hf[j + win - 1] < hf[best_f]) {
                best_f = j + win - 1;
            }
            uint64_t mini = vf[best_f], pos = best_f;
            if (idx.canonical) {
                /* reverse strand: rightmost minimum over [j, j+win) */
                if (j == 0 || best_r < j) {
                    best_r = j;
                    for (uint64_t p = j + 1; p < j + win; ++p)
                        if (hr[p] <= hr[best_r]) best_r = p;
                } else if (hr[j + win - 1] <= hr[best_r]) {
                    best_r = j + win - 1;
                }
                if (vr[best_r] < mini) {
                    mini = vr[best_r];
                    pos = best_r;
                }
            }
            const uint64_t abs_pos = begin + pos;
            if (!have || mini != cur_min || abs_pos != cur_pos) {
                if (have) out.push_back({cur_min, (cur_pos << 16) | (cur_first_pik << 8) | cur_count});
                have = true;
                cur_min = mini;
                cur_pos = abs_pos;
                cur_first_pik = pos - j;
                cur_count = 0;
            }
            ++cur_count;
        }
        if (have) out.push_back({cur_min, (cur_pos << 16) | (cur_first_pik << 8) | cur_count});
    }
}

template <int W>
void build_skew_index(host_index& idx, std::vector<mini_tuple> const& tuples,
                      std::vector<uint64_t> const& heavy_order /* bucket ids, size ascending */,
                      std::vector<uint64_t> const& bucket_begin, std::vector<uint64_t> const& bucket_end,
                      std::vector<uint32_t> const& bucket_size,
                      std::vector<uint32_t> const& heavy_partition /* per entry of heavy_order */,
                      build_options const& opt, uint64_t mphf_seed) {
    const uint32_t k = idx.k;
    uint64_t const* words = idx.strings.data();
    for (uint32_t part = 0; part < idx.skew_num_partitions; ++part) {
        std::vector<kmer_w<W>> kmers;
        std::vector<uint32_t> pos_in_bucket;
        uint32_t max_pos = 0;
        for (uint64_t h = 0; h < heavy_order.size(); ++h) {
            if (heavy_partition[h] != part) continue;
            const uint64_t b = heavy_order[h];
            uint64_t prev_pos = INVALID_U64;
            uint32_t pib = uint32_t(-1);
            for (uint64_t t = bucket_begin[b]; t < bucket_end[b]; ++t) {
                auto const& mt = tuples[t];
                if (mt.pos() != prev_pos) {
                    prev_pos = mt.pos();
                    ++pib;
                }
                const uint64_t start = mt.pos() - mt.pos_in_kmer();
                for (uint32_t i = 0; i < mt.num_kmers(); ++i) {
                    kmer_w<W> x = read_kmer<W>(words, start + i, k);
                    if (idx.canonical) { /* build_sparse_and_skew_index.cpp:462-466 */
                        kmer_w<W> r = kmer_revcomp<W>(x, k);
                        if (kmer_less<W>(r, x)) x = r;
                    }
                    kmers.push_back(x);
                    pos_in_bucket.push_back(pib);
                }
            }
            max_pos = std::max(max_pos, uint32_t(bucket_size[b] - 1));
        }
        if (kmers.empty()) continue;
        mphf_build_config cfg;
        cfg.lambda = opt.lambda + 2.0;  // build_sparse_and_skew_index.cpp:312-313
        cfg.alpha = 0.94;
        cfg.seed = mphf_seed;
        cfg.num_threads = opt.num_threads;
        auto& F = idx.skew_mphfs[part];
        mphf_build(F, kmers.size(),
                   [&](uint64_t seed) { return [&kmers, seed](uint64_t i) { return city128_kmer<W>(kmers[i], seed); }; },
                   cfg);
        auto& P = idx.skew_positions[part];
        P.resize(kmers.size(), bits_for(max_pos));
        const mphf_view fv = F.view();
        for (uint64_t i = 0; i < kmers.size(); ++i)
            P.set(mphf_eval(fv, city128_kmer<W>(kmers[i], F.seed)), pos_in_bucket[i]);
    }
}

}  // namespace

uint64_t host_index::num_bits() const {
    uint64_t b = 8 * (strings.size() * 8 + endpoints.size() * 8 + begin_buckets_of_size.size() * 4 +
                      control_codewords.num_bytes() + mid_load_buckets.num_bytes() +
                      heavy_load_buckets.num_bytes());
    b += minimizers_mphf.num_bits();
    for (uint32_t p = 0; p < skew_num_partitions; ++p) b += skew_mphfs[p].num_bits() + 8 * skew_positions[p].num_bytes();
    b += 64 * (weight_starts.size() + weight_values.size());
    return b;
}

void build_from_packed(host_index& idx, std::vector<uint64_t>&& packed_bases, std::vector<uint64_t>&& endpoints,
                       build_options const& opt) {
    if (opt.k < 1 || opt.k > 63 || (opt.k % 2) == 0) throw error(error_kind::build, "k must be odd and in [1,63]");
    if (opt.m < 1 || opt.m > 31 || opt.m > opt.k) throw error(error_kind::build, "m must be in [1,min(31,k)]");
    if (opt.k - opt.m + 1 >= 256) throw error(error_kind::build, "k-m+1 does not fit 8 bits");
    if (endpoints.size() < 2 || endpoints.front() != 0) throw error(error_kind::build, "bad endpoints");
    timer tm;
    idx = host_index();
    idx.k = opt.k;
    idx.m = opt.m;
    idx.canonical = opt.canonical;
    if (opt.num_shards == 0 || opt.shard_id >= opt.num_shards) throw error(error_kind::build, "shard_id must be < num_shards");
    idx.build_seed = opt.seed;
    idx.num_shards = opt.num_shards;
    idx.shard_id = opt.shard_id;
    idx.hash_magic = xxh64_of_u64(opt.seed, 0);  // include/hash_util.hpp:88
    idx.endpoints = std::move(endpoints);
    idx.num_strings = idx.endpoints.size() - 1;
    idx.num_bases = idx.endpoints.back();
    idx.num_kmers = 0;
    for (uint64_t s = 0; s < idx.num_strings; ++s) {
        if (idx.endpoints[s + 1] < idx.endpoints[s]) throw error(error_kind::build, "endpoints must not decrease");
        const uint64_t len = idx.endpoints[s + 1] - idx.endpoints[s];
        if (len < opt.k) throw error(error_kind::build, "input string shorter than k");
        idx.num_kmers += len - opt.k + 1;
    }
    if (packed_bases.size() < (2 * idx.num_bases + 63) / 64) throw error(error_kind::build, "fewer packed words than the endpoints announce");
    const uint32_t W = idx.words_per_kmer();
    idx.strings = std::move(packed_bases);
    /* zero sentinel of one k-mer word-width (src/builder/encode_strings.cpp:183-188) + slack
       so that W+1 consecutive words are addressable from any base offset */
    const uint64_t data_words = (2 * idx.num_bases + 63) / 64;
    idx.strings.resize(data_words);
    if ((2 * idx.num_bases) % 64) idx.strings.back() &= low_mask(uint32_t((2 * idx.num_bases) % 64));
    idx.strings_num_bits = 2 * idx.num_bases + 64 * W;
    idx.strings.resize(data_words + W + 2, 0);

    const uint32_t nt = std::max(1u, opt.num_threads);

    /* 1. super-k-mer tuples */
    std::vector<mini_tuple> tuples;
    {
        std::vector<std::vector<mini_tuple>> per_thread(nt);
        /* balance by bases, not by string count */
        std::vector<uint64_t> cut(nt + 1, idx.num_strings);
        cut[0] = 0;
        for (uint32_t t = 1; t < nt; ++t) {
            const uint64_t target = idx.num_bases / nt * t;
            cut[t] = std::lower_bound(idx.endpoints.begin(), idx.endpoints.end() - 1, target) - idx.endpoints.begin();
        }
        for (uint32_t t = 1; t <= nt; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
        detail::parallel_for(nt, nt, [&](uint64_t t) { tuples_of_strings(idx, cut[t], cut[t + 1], per_thread[t]); });
        uint64_t total = 0;
        for (auto& v : per_thread) total += v.size();
        tuples.reserve(total);
        for (auto& v : per_thread) {
            tuples.insert(tuples.end(), v.begin(), v.end());
            std::vector<mini_tuple>().swap(v);
        }
    }
    if (opt.verbose) fprintf(stderr, "[build] %zu super-k-mer tuples in %.2fs\n", tuples.size(), tm.lap());
    parallel_sort(tuples, nt);
    if (opt.verbose) fprintf(stderr, "[build] sorted in %.2fs\n", tm.lap());

    /* 2. buckets = runs of equal minimizer; size = number of DISTINCT positions
          (include/builder/util.hpp:61-78) */
    std::vector<uint64_t> keys;
    std::vector<uint64_t> bucket_begin, bucket_end;
    std::vector<uint32_t> bucket_size;
    for (uint64_t t = 0; t < tuples.size();) {
        uint64_t e = t;
        uint32_t distinct = 0;
        uint64_t prev = INVALID_U64;
        while (e < tuples.size() && tuples[e].minimizer == tuples[t].minimizer) {
            if (tuples[e].pos() != prev) {
                prev = tuples[e].pos();
                ++distinct;
            }
            ++e;
        }
        if (opt.num_shards == 1 || shard_of_minimizer(tuples[t].minimizer, opt.num_shards) == opt.shard_id) {
            keys.push_back(tuples[t].minimizer);
            bucket_begin.push_back(t);
            bucket_end.push_back(e);
            bucket_size.push_back(distinct);
        }
        t = e;
    }
    if (keys.empty()) throw error(error_kind::build, "no minimizer falls into this shard: the input is too small to be sharded this way");
    const uint64_t num_minimizers = keys.size();

    /* 3. minimizers MPHF (include/minimizers_control_map.hpp:6-34) */
    const uint64_t favourite = 1234567890ULL;  // include/util.hpp:197-200
    const uint64_t mphf_seed = opt.seed != favourite ? favourite : ~favourite;
    {
        mphf_build_config cfg;
        cfg.lambda = opt.lambda;
        cfg.alpha = 0.94;
        cfg.seed = mphf_seed;
        cfg.num_threads = nt;
        mphf_build(idx.minimizers_mphf, num_minimizers,
                   [&](uint64_t seed) { return [&keys, seed](uint64_t i) { return city128_u64(keys[i], seed); }; }, cfg);
    }
    if (opt.verbose)
        fprintf(stderr, "[build] MPHF over %lu minimizers (%u partitions, %u-bit pilots) in %.2fs\n",
                (unsigned long)num_minimizers, (unsigned)idx.minimizers_mphf.parts.size(), idx.minimizers_mphf.pilot_width,
                tm.lap());

    /* 4. classify buckets, lay out mid/heavy offset lists, assign control codewords
          (src/builder/build_sparse_and_skew_index.cpp:102-244) */
    const uint32_t bits_per_offset = bits_for(idx.num_bases > 1 ? idx.num_bases - 1 : 1);
    std::vector<uint64_t> count_of_size(MAX_BUCKET_SMALL + 1, 0);
    std::vector<uint64_t> heavy_order;
    uint32_t max_bucket_size = 0;
    uint64_t num_mid_positions = 0, num_heavy_positions = 0;
    for (uint64_t b = 0; b < num_minimizers; ++b) {
        const uint32_t s = bucket_size[b];
        max_bucket_size = std::max(max_bucket_size, s);
        if (s >= 2 && s <= MAX_BUCKET_SMALL) {
            ++count_of_size[s];
            num_mid_positions += s;
        } else if (s > MAX_BUCKET_SMALL) {
            heavy_order.push_back(b);
            num_heavy_positions += s;
        }
    }
    std::stable_sort(heavy_order.begin(), heavy_order.end(),
                     [&](uint64_t a, uint64_t b) { return bucket_size[a] < bucket_size[b]; });
    idx.begin_buckets_of_size.assign(MAX_BUCKET_SMALL + 1, 0);
    {
        uint64_t acc = 0;
        for (uint32_t s = 2; s <= MAX_BUCKET_SMALL; ++s) {
            if (acc >= (uint64_t(1) << 32)) throw error(error_kind::build, "mid_load_buckets exceeds 2^32 entries");
            idx.begin_buckets_of_size[s] = uint32_t(acc);
            acc += count_of_size[s] * s;
        }
    }
    /* skew partitions: (64,128], (128,256], ... last one open-ended (:141-147) */
    uint32_t log2_max = 0;
    while ((uint64_t(1) << log2_max) < max_bucket_size) ++log2_max;
    uint32_t num_partitions = MAX_L - MIN_L + 1;
    if (max_bucket_size <= MAX_BUCKET_SMALL) num_partitions = 0;
    else if (max_bucket_size < (1u << MAX_L)) num_partitions = log2_max - MIN_L;
    idx.skew_num_partitions = num_partitions;
    std::vector<uint32_t> heavy_partition(heavy_order.size(), 0);
    std::vector<uint64_t> heavy_begin(heavy_order.size(), 0);
    {
        uint64_t acc = 0;
        for (uint64_t h = 0; h < heavy_order.size(); ++h) {
            const uint32_t s = bucket_size[heavy_order[h]];
            uint32_t p = 0;
            while (p + 1 < num_partitions && s > (uint64_t(2 * MAX_BUCKET_SMALL) << p)) ++p;
            heavy_partition[h] = p;
            heavy_begin[h] = acc;
            acc += s;
        }
    }
    /* per-bucket codeword */
    std::vector<uint64_t> code(num_minimizers);
    {
        std::vector<uint64_t> next_list_id(MAX_BUCKET_SMALL + 1, 0);
        for (uint64_t b = 0; b < num_minimizers; ++b) {
            const uint32_t s = bucket_size[b];
            if (s == 1) code[b] = tuples[bucket_begin[b]].pos() << 1;
            else if (s <= MAX_BUCKET_SMALL) code[b] = ((((next_list_id[s]++) << MIN_L) | (s - 2)) << 2) | 1;
        }
        for (uint64_t h = 0; h < heavy_order.size(); ++h)
            code[heavy_order[h]] = (((heavy_begin[h] << 3) | heavy_partition[h]) << 2) | 3;
    }
    uint64_t max_code = (uint64_t(1) << bits_per_offset);  // width >= bits_per_offset + 1 (:58-61)
    for (uint64_t c : code) max_code = std::max(max_code, c);
    idx.control_codewords.resize(num_minimizers, bits_for(max_code));
    /* (the device's bucket-scan pass carries a bucket's place in this array in 32 bits: lookup_device.hpp scan_meta) */
    if (num_mid_positions >= (uint64_t(1) << 32)) throw error(error_kind::build, "mid_load_buckets exceeds 2^32 entries");
    idx.mid_load_buckets.resize(num_mid_positions, bits_per_offset);
    idx.heavy_load_buckets.resize(num_heavy_positions, bits_per_offset);
    {
        const mphf_view fv = idx.minimizers_mphf.view();
        uint64_t* cw = idx.control_codewords.words.data();
        const uint32_t cww = idx.control_codewords.width;
        detail::parallel_ranges(num_minimizers, nt, [&](uint64_t b0, uint64_t b1, uint32_t) {
            for (uint64_t b = b0; b < b1; ++b)
                atomic_packed_set(cw, mphf_eval(fv, city128_u64(keys[b], fv.seed)), cww, code[b]);
        });
        /* offsets of mid-load buckets, grouped by size then list id */
        for (uint64_t b = 0; b < num_minimizers; ++b) {
            const uint32_t s = bucket_size[b];
            if (s < 2 || s > MAX_BUCKET_SMALL) continue;
            const uint64_t list_id = code[b] >> (2 + MIN_L);
            uint64_t at = idx.begin_buckets_of_size[s] + list_id * s;
            uint64_t prev = INVALID_U64;
            for (uint64_t t = bucket_begin[b]; t < bucket_end[b]; ++t) {
                if (tuples[t].pos() != prev) {
                    prev = tuples[t].pos();
                    idx.mid_load_buckets.set(at++, prev);
                }
            }
        }
        for (uint64_t h = 0; h < heavy_order.size(); ++h) {
            const uint64_t b = heavy_order[h];
            uint64_t at = heavy_begin[h];
            uint64_t prev = INVALID_U64;
            for (uint64_t t = bucket_begin[b]; t < bucket_end[b]; ++t) {
                if (tuples[t].pos() != prev) {
                    prev = tuples[t].pos();
                    idx.heavy_load_buckets.set(at++, prev);
                }
            }
        }
    }
    if (opt.verbose)
        fprintf(stderr, "[build] sparse index: %lu mid positions, %lu heavy positions, max bucket %u, cw %u bits in %.2fs\n",
                (unsigned long)num_mid_positions, (unsigned long)num_heavy_positions, max_bucket_size,
                idx.control_codewords.width, tm.lap());

    /* 5. skew index (:261-478) */
    if (num_partitions) {
        if (W == 1) build_skew_index<1>(idx, tuples, heavy_order, bucket_begin, bucket_end, bucket_size, heavy_partition, opt, mphf_seed);
        else build_skew_index<2>(idx, tuples, heavy_order, bucket_begin, bucket_end, bucket_size, heavy_partition, opt, mphf_seed);
        if (opt.verbose) fprintf(stderr, "[build] skew index (%u partitions) in %.2fs\n", num_partitions, tm.lap());
    }
}

namespace {
void pack_append(std::vector<uint64_t>& words, uint64_t& num_bases, char const* s, uint64_t n) {
    words.resize((2 * (num_bases + n) + 63) / 64 + 1, 0);
    for (uint64_t i = 0; i < n; ++i, ++num_bases)
        words[num_bases >> 5] |= uint64_t(base_code(s[i])) << ((num_bases & 31) * 2);
}
}  // namespace

void build_from_sequences(host_index& idx, std::vector<std::string> const& seqs, build_options const& opt) {
    std::vector<uint64_t> words, endpoints{0};
    uint64_t num_bases = 0;
    for (auto const& s : seqs) {
        pack_append(words, num_bases, s.data(), s.size());
        endpoints.push_back(num_bases);
    }
    build_from_packed(idx, std::move(words), std::move(endpoints), opt);
}

void build_from_fasta(host_index& idx, std::string const& filename, build_options const& opt) {
    gzFile f = gzopen(filename.c_str(), "rb");  // transparently reads plain or gzip
    if (!f) throw error(error_kind::io, "error in opening the file '" + filename + "'");
    gzbuffer(f, 1 << 20);
    std::vector<uint64_t> words, endpoints{0};
    uint64_t num_bases = 0;
    std::vector<char> buf(1 << 16);
    /* getline: returns false on EOF-before-any-char; `hit_eof` tells whether the line was cut by EOF */
    auto getline = [&](std::string& line, bool& hit_eof) -> bool {
        line.clear();
        hit_eof = false;
        for (;;) {
            if (!gzgets(f, buf.data(), int(buf.size()))) {
                hit_eof = true;
                return !line.empty();
            }
            const size_t n = strlen(buf.data());
            if (n && buf[n - 1] == '\n') {
                line.append(buf.data(), n - 1);
                return true;
            }
            line.append(buf.data(), n);
        }
    };
    /* Cuttlefish segment files ("<id>\t<sequence>" per line, src/builder/encode_strings.cpp:79-80) by extension,
       as the reference does (src/builder/encode_strings.cpp:226-262) */
    auto ends_with = [&](char const* suffix) {
        const size_t n = strlen(suffix);
        return filename.size() >= n && filename.compare(filename.size() - n, n, suffix) == 0;
    };
    const bool cf_seg = ends_with(".cf_seg") || ends_with(".cf_seg.gz");
    if (cf_seg && opt.weighted) {
        gzclose(f);
        throw error(error_kind::build, "weights are read from FASTA headers: a .cf_seg input has none");
    }
    std::string header, seq;
    bool eof = false;
    /* run-length intervals of weights over the k-mers in file order (encode_strings.cpp:73-75,120-132,216-218) */
    std::vector<uint64_t> weight_starts, weight_values;
    uint64_t num_kmers = 0;
    auto malformed = [&](char const* what) {
        gzclose(f);
        throw error(error_kind::build, std::string("file is malformed: ") + what);
    };
    for (;;) {
        if (!getline(header, eof) && eof) break;  // unweighted: the header line is skipped (encode_strings.cpp:136)
        uint64_t declared_len = 0;
        if (opt.weighted) {
            /* '>[id] LN:i:[seq_len] ab:Z:[weight_seq]' with seq_len - k + 1 space-separated counters */
            if (header.empty()) break;
            if (header[0] != '>') malformed("header does not start with '>'");
            size_t i = header.find(' ');
            if (i == std::string::npos || header.compare(i + 1, 5, "LN:i:") != 0) malformed("expected 'LN:i:'");
            i += 6;
            const size_t j = header.find(' ', i);
            if (j == std::string::npos || header.compare(j + 1, 5, "ab:Z:") != 0) malformed("expected 'ab:Z:'");
            declared_len = std::strtoull(header.c_str() + i, nullptr, 10);
            if (declared_len < opt.k) malformed("sequence shorter than k");
            i = j + 6;
            for (uint64_t t = 0; t != declared_len - opt.k + 1; ++t) {
                if (i >= header.size()) malformed("fewer weights than k-mers");
                const uint64_t w = std::strtoull(header.c_str() + i, nullptr, 10);
                const size_t sp = header.find(' ', i);
                i = sp == std::string::npos ? header.size() : sp + 1;
                if (weight_values.empty() || weight_values.back() != w) {
                    weight_starts.push_back(num_kmers + t);
                    weight_values.push_back(w);
                }
            }
        }
        if (cf_seg) {  // the line just read is "<id>\t<sequence>"
            if (eof) break;
            const size_t tab = header.find('\t');
            seq = tab == std::string::npos ? std::string() : header.substr(tab + 1);
        } else {
            getline(seq, eof);
            if (eof) break;  // a last line without '\n' is dropped, as in encode_strings.cpp:139-140
        }
        if (!seq.empty() && seq.back() == '\r') seq.pop_back();
        if (seq.size() < opt.k) {
            gzclose(f);
            throw error(error_kind::build, "input sequence shorter than k");
        }
        if (opt.weighted && declared_len != seq.size()) malformed("sequence length differs from its LN:i: field");  // :151-155
        pack_append(words, num_bases, seq.data(), seq.size());
        endpoints.push_back(num_bases);
        num_kmers += seq.size() - opt.k + 1;
    }
    gzclose(f);
    if (endpoints.size() < 2) throw error(error_kind::build, "no sequences in '" + filename + "'");
    /* weights parsed from a header whose sequence line was dropped (EOF) do not belong to any k-mer */
    while (!weight_starts.empty() && weight_starts.back() >= num_kmers) {
        weight_starts.pop_back();
        weight_values.pop_back();
    }
    build_from_packed(idx, std::move(words), std::move(endpoints), opt);
    idx.weight_starts = std::move(weight_starts);
    idx.weight_values = std::move(weight_values);
}

uint64_t weight_of(host_index const& idx, uint64_t kmer_id) {
    if (!idx.weighted()) throw error(error_kind::argument, "the dictionary does not store weights");
    if (kmer_id >= idx.num_kmers) throw error(error_kind::argument, "kmer_id out of range");
    /* prev_leq over the interval starts, include/weights.hpp:148 */
    const auto it = std::upper_bound(idx.weight_starts.begin(), idx.weight_starts.end(), kmer_id);
    return idx.weight_values[size_t(it - idx.weight_starts.begin()) - 1];
}

/* ---- access ---------------------------------------------------------------------------- */

static uint64_t id_to_offset(host_index const& idx, uint64_t kmer_id) {
    /* largest s with endpoints[s] - s*(k-1) <= kmer_id (include/offsets.hpp:41-65) */
    uint64_t lo = 0, hi = idx.num_strings - 1;
    const uint64_t km1 = idx.k - 1;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (idx.endpoints[mid] - mid * km1 <= kmer_id) lo = mid;
        else hi = mid - 1;
    }
    return kmer_id + lo * km1;
}

void access_kmer_packed(host_index const& idx, uint64_t kmer_id, uint64_t* out) {
    if (kmer_id >= idx.num_kmers) throw error(error_kind::argument, "kmer_id out of range");
    const uint64_t off = id_to_offset(idx, kmer_id);
    if (idx.words_per_kmer() == 1) {
        out[0] = read_kmer<1>(idx.strings.data(), off, idx.k).w[0];
    } else {
        auto x = read_kmer<2>(idx.strings.data(), off, idx.k);
        out[0] = x.w[0];
        out[1] = x.w[1];
    }
}

void access_kmer(host_index const& idx, uint64_t kmer_id, char* out) {
    if (kmer_id >= idx.num_kmers) throw error(error_kind::argument, "kmer_id out of range");
    const uint64_t off = id_to_offset(idx, kmer_id);
    static const char alphabet[] = "ACTG";  // include/kmer.hpp:118
    for (uint32_t i = 0; i < idx.k; ++i) out[i] = alphabet[base_at(idx.strings.data(), off + i)];
}

/* ---- (de)serialisation ----------------------------------------------------------------- */
//
// File layout (all little-endian, every section 8-byte aligned):
//   "SSHAMD\x03\x00"  magic
//   u8 version[3], u8 canonical, u32 k, u32 m, u32 skew_num_partitions
//   u64 num_kmers, num_strings, num_bases, hash_magic, build_seed, strings_num_bits, num_shards | shard_id << 32
//   vec<u64> strings, vec<u64> endpoints
//   mphf minimizers, packed control_codewords, vec<u32> begin_buckets_of_size, packed mid_load_buckets
//   skew_num_partitions x { mphf, packed positions }, packed heavy_load_buckets
//   vec<u64> weight_starts, vec<u64> weight_values   (both empty for an unweighted dictionary)
// where vec<T> = u64 count + raw data padded to 8 bytes; packed = u64 size, u64 width, vec<u64>;
// mphf = u64 seed, u64 num_keys, u64 pilot_width, vec<partition 48B>, vec<u64> pilots, vec<u32> free_slots.
// The reference's own .sshash byte format is defined by essentials/bits/pthash, whose sources are
// absent from the reference checkout, so it cannot be reproduced here ("index file").

namespace {
struct writer {
    FILE* f;
    void raw(void const* p, size_t n) {
        if (n && fwrite(p, 1, n, f) != n) throw error(error_kind::io, "write error");
    }
    void u64(uint64_t v) { raw(&v, 8); }
    template <typename T>
    void vec(std::vector<T> const& v) {
        u64(v.size());
        raw(v.data(), v.size() * sizeof(T));
        const size_t pad = (8 - (v.size() * sizeof(T)) % 8) % 8;
        const uint64_t z = 0;
        raw(&z, pad);
    }
    void packed(packed_vec const& p) {
        u64(p.size);
        u64(p.width);
        vec(p.words);
    }
    void mphf(mphf_host const& m) {
        u64(m.seed);
        u64(m.num_keys);
        u64(m.pilot_width);
        vec(m.parts);
        vec(m.pilots);
        vec(m.free_slots);
    }
};
struct reader {
    FILE* f;
    uint64_t left;  // bytes of the file not read yet: a length field can never ask for more
    void raw(void* p, size_t n) {
        if (n > left || (n && fread(p, 1, n, f) != n)) throw error(error_kind::format, "index file truncated");
        left -= n;
    }
    uint64_t u64() {
        uint64_t v;
        raw(&v, 8);
        return v;
    }
    template <typename T>
    void vec(std::vector<T>& v) {
        const uint64_t n = u64();
        if (n > left / sizeof(T)) throw error(error_kind::format, "index file corrupt (a vector longer than the file)");
        v.resize(n);
        raw(v.data(), n * sizeof(T));
        const size_t pad = (8 - (n * sizeof(T)) % 8) % 8;
        uint64_t z;
        raw(&z, pad);
    }
    void packed(packed_vec& p) {
        p.size = u64();
        p.width = uint32_t(u64());
        vec(p.words);
        if (p.width < 1 || p.width > 64 || p.size > (uint64_t(1) << 56) || p.words.size() < (p.size * p.width + 63) / 64 + 1)
            throw error(error_kind::format, "index file corrupt (packed vector)");
    }
    void mphf(mphf_host& m) {
        m.seed = u64();
        m.num_keys = u64();
        m.pilot_width = uint32_t(u64());
        vec(m.parts);
        vec(m.pilots);
        vec(m.free_slots);
    }
};
const char FILE_MAGIC[8] = {'S', 'S', 'H', 'A', 'M', 'D', 3, 0};
}  // namespace

void save_index(host_index const& idx, std::string const& filename) {
    FILE* f = fopen(filename.c_str(), "wb");
    if (!f) throw error(error_kind::io, "cannot open '" + filename + "' for writing");
    try {
        writer w{f};
        w.raw(FILE_MAGIC, 8);
        uint8_t hdr[4] = {idx.version[0], idx.version[1], idx.version[2], uint8_t(idx.canonical)};
        w.raw(hdr, 4);
        uint32_t kms[3] = {idx.k, idx.m, idx.skew_num_partitions};
        w.raw(kms, 12);
        w.u64(idx.num_kmers);
        w.u64(idx.num_strings);
        w.u64(idx.num_bases);
        w.u64(idx.hash_magic);
        w.u64(idx.build_seed);
        w.u64(idx.strings_num_bits);
        w.u64(uint64_t(idx.num_shards) | (uint64_t(idx.shard_id) << 32));
        w.vec(idx.strings);
        w.vec(idx.endpoints);
        w.mphf(idx.minimizers_mphf);
        w.packed(idx.control_codewords);
        w.vec(idx.begin_buckets_of_size);
        w.packed(idx.mid_load_buckets);
        for (uint32_t p = 0; p < idx.skew_num_partitions; ++p) {
            w.mphf(idx.skew_mphfs[p]);
            w.packed(idx.skew_positions[p]);
        }
        w.packed(idx.heavy_load_buckets);
        w.vec(idx.weight_starts);
        w.vec(idx.weight_values);
    } catch (...) {
        fclose(f);
        throw;
    }
    fclose(f);
}

/* Everything the device upload and the kernels rely on without checking again: a file that passes may hold a wrong
   dictionary, but not one that makes the engine read or write out of bounds (make_granules indexes by
   endpoint, the kernels index by codeword, offset and MPHF geometry). */
static void validate_mphf(mphf_host const& f, char const* what) {
    auto bad = [&](char const* why) { throw error(error_kind::format, std::string("index file corrupt (") + what + ": " + why + ")"); };
    if (f.pilot_width < 1 || f.pilot_width > 32) bad("pilot width");
    if (f.num_keys == 0) {
        if (!f.parts.empty()) bad("partitions of an empty function");
        return;
    }
    if (f.parts.empty()) bad("no partition");
    uint64_t keys = 0, pilots = 0, frees = 0;
    for (mphf_partition const& p : f.parts) {
        if (p.key_offset != keys || p.pilot_base != pilots || p.free_base != frees) bad("partition offsets");
        if (p.num_keys == 0 || p.table_size < p.num_keys) bad("table size");
        if (p.dense_buckets == 0 || p.sparse_buckets == 0) bad("bucket counts");
        keys += p.num_keys;
        pilots += uint64_t(p.dense_buckets) + p.sparse_buckets;
        frees += p.table_size - p.num_keys;
    }
    if (keys != f.num_keys) bad("key count");
    if (f.pilots.size() < (pilots * f.pilot_width + 63) / 64 + 1) bad("pilot vector");
    if (f.free_slots.size() < frees) bad("free slots");
    for (mphf_partition const& p : f.parts)
        for (uint64_t i = 0; i < uint64_t(p.table_size - p.num_keys); ++i)
            if (f.free_slots[p.free_base + i] >= p.num_keys) bad("free slot value");
}

static void validate_packed(packed_vec const& v, char const* what) {
    if (v.width < 1 || v.width > 64 || v.words.size() < (v.size * v.width + 63) / 64 + 1)
        throw error(error_kind::format, std::string("index file corrupt (") + what + ")");
}

static void validate_index(host_index const& idx) {
    auto bad = [](char const* why) { throw error(error_kind::format, std::string("index file corrupt (") + why + ")"); };
    if (idx.endpoints.size() != idx.num_strings + 1 || idx.endpoints.empty() || idx.endpoints.front() != 0 || idx.endpoints.back() != idx.num_bases)
        bad("endpoints");
    uint64_t kmers = 0;
    for (uint64_t s = 0; s < idx.num_strings; ++s) {
        if (idx.endpoints[s + 1] < idx.endpoints[s] || idx.endpoints[s + 1] - idx.endpoints[s] < idx.k) bad("string shorter than k");
        kmers += idx.endpoints[s + 1] - idx.endpoints[s] - idx.k + 1;
    }
    if (kmers != idx.num_kmers) bad("num_kmers");
    const uint64_t W = idx.words_per_kmer();
    if (idx.strings.size() < (2 * idx.num_bases + 63) / 64 + W + 2 || idx.strings_num_bits != 2 * idx.num_bases + 64 * W) bad("strings");
    if (idx.begin_buckets_of_size.size() != MAX_BUCKET_SMALL + 1) bad("bucket table");
    validate_mphf(idx.minimizers_mphf, "minimizers");
    validate_packed(idx.control_codewords, "control codewords");
    validate_packed(idx.mid_load_buckets, "mid-load buckets");
    if (idx.mid_load_buckets.size >= (uint64_t(1) << 32)) bad("mid-load buckets (2^32 entries or more)");
    validate_packed(idx.heavy_load_buckets, "heavy-load buckets");
    if (idx.control_codewords.size != idx.minimizers_mphf.num_keys) bad("one control codeword per minimizer");
    if (idx.heavy_load_buckets.size && idx.mid_load_buckets.size && idx.mid_load_buckets.width != idx.heavy_load_buckets.width)
        bad("offset widths");  // the device reads both lists with one width
    for (uint64_t i = 0; i < idx.mid_load_buckets.size; ++i)
        if (idx.mid_load_buckets.get(i) >= idx.num_bases) bad("mid-load offset");
    for (uint64_t i = 0; i < idx.heavy_load_buckets.size; ++i)
        if (idx.heavy_load_buckets.get(i) >= idx.num_bases) bad("heavy-load offset");
    for (uint32_t p = 0; p < idx.skew_num_partitions; ++p) {
        validate_mphf(idx.skew_mphfs[p], "skew index");
        validate_packed(idx.skew_positions[p], "skew positions");
        if (idx.skew_positions[p].size != idx.skew_mphfs[p].num_keys) bad("one skew position per k-mer");
    }
    /* control codewords (src/builder/build_sparse_and_skew_index.cpp:110-124,204-235): every one must point inside
       the structure its two low bits select */
    for (uint64_t i = 0; i < idx.control_codewords.size; ++i) {
        const uint64_t code = idx.control_codewords.get(i);
        if ((code & 1) == 0) {
            if ((code >> 1) >= idx.num_bases) bad("singleton offset");
        } else if ((code & 3) == 1) {
            const uint64_t size = ((code >> 2) & (MAX_BUCKET_SMALL - 1)) + 2;
            if (size > MAX_BUCKET_SMALL) bad("mid-load bucket size");
            const uint64_t begin = uint64_t(idx.begin_buckets_of_size[size]) + (code >> (2 + MIN_L)) * size;
            if (begin + size > idx.mid_load_buckets.size) bad("mid-load bucket");
        } else {
            if (((code >> 2) & 7) >= idx.skew_num_partitions || (code >> 5) >= std::max<uint64_t>(idx.heavy_load_buckets.size, 1)) bad("heavy-load bucket");
        }
    }
    if (idx.weight_starts.size() != idx.weight_values.size() || (!idx.weight_starts.empty() && idx.weight_starts[0] != 0)) bad("weights");
    for (size_t i = 1; i < idx.weight_starts.size(); ++i)
        if (idx.weight_starts[i] <= idx.weight_starts[i - 1] || idx.weight_starts[i] >= idx.num_kmers) bad("weight intervals");
}

void load_index(host_index& idx, std::string const& filename) {
    FILE* f = fopen(filename.c_str(), "rb");
    if (!f) throw error(error_kind::io, "error in opening the file '" + filename + "'");
    try {
        uint64_t file_bytes = 0;
        if (fseek(f, 0, SEEK_END) == 0) {
            const long at = ftell(f);
            if (at > 0) file_bytes = uint64_t(at);
        }
        rewind(f);
        reader r{f, file_bytes};
        idx = host_index();
        char magic[8];
        r.raw(magic, 8);
        if (memcmp(magic, FILE_MAGIC, 8) != 0) throw error(error_kind::format, "not an sshash_amd index file");
        uint8_t hdr[4];
        r.raw(hdr, 4);
        /* util::check_version_number, include/util.hpp:191-195 */
        if (hdr[0] != 5) throw error(error_kind::version, "MAJOR index version mismatch: SSHash index needs rebuilding");
        idx.version[0] = hdr[0];
        idx.version[1] = hdr[1];
        idx.version[2] = hdr[2];
        idx.canonical = hdr[3] != 0;
        uint32_t kms[3];
        r.raw(kms, 12);
        idx.k = kms[0];
        idx.m = kms[1];
        idx.skew_num_partitions = kms[2];
        if (idx.k < 1 || idx.k > 63 || idx.m < 1 || idx.m > 31 || idx.m > idx.k || idx.skew_num_partitions > 8)
            throw error(error_kind::format, "index file corrupt (header)");
        idx.num_kmers = r.u64();
        idx.num_strings = r.u64();
        idx.num_bases = r.u64();
        idx.hash_magic = r.u64();
        idx.build_seed = r.u64();
        idx.strings_num_bits = r.u64();
        {
            const uint64_t sh = r.u64();
            idx.num_shards = uint32_t(sh);
            idx.shard_id = uint32_t(sh >> 32);
            if (idx.num_shards == 0 || idx.shard_id >= idx.num_shards) throw error(error_kind::format, "index file corrupt (shard)");
        }
        r.vec(idx.strings);
        r.vec(idx.endpoints);
        r.mphf(idx.minimizers_mphf);
        r.packed(idx.control_codewords);
        r.vec(idx.begin_buckets_of_size);
        r.packed(idx.mid_load_buckets);
        for (uint32_t p = 0; p < idx.skew_num_partitions; ++p) {
            r.mphf(idx.skew_mphfs[p]);
            r.packed(idx.skew_positions[p]);
        }
        r.packed(idx.heavy_load_buckets);
        r.vec(idx.weight_starts);
        r.vec(idx.weight_values);
        char extra;
        if (fread(&extra, 1, 1, f) != 0) throw error(error_kind::format, "index file has trailing bytes");
        validate_index(idx);
    } catch (...) {
        fclose(f);
        throw;
    }
    fclose(f);
}

void bucket_statistics(host_index const& idx, uint64_t* out) {
    for (uint32_t i = 0; i < BUCKET_STATS_WORDS; ++i) out[i] = 0;
    const uint64_t n = idx.control_codewords.size;
    std::vector<uint64_t> heavy_begins;
    uint64_t singletons = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t code = idx.control_codewords.get(i);
        if ((code & 1) == 0) {
            ++singletons;
        } else if ((code & 3) == 1) {
            const uint32_t size = uint32_t((code >> 2) & (MAX_BUCKET_SMALL - 1)) + 2;
            ++out[2];
            out[3] += size;
            if (size <= 16) ++out[16 + size - 1];
            out[7] = std::max<uint64_t>(out[7], size);
        } else {
            heavy_begins.push_back(code >> 5);
        }
    }
    out[0] = n;
    out[16] = singletons;
    if (singletons) out[7] = std::max<uint64_t>(out[7], 1);
    std::sort(heavy_begins.begin(), heavy_begins.end());
    out[4] = heavy_begins.size();
    out[5] = idx.heavy_load_buckets.size;
    for (size_t h = 0; h < heavy_begins.size(); ++h) {
        const uint64_t end = h + 1 < heavy_begins.size() ? heavy_begins[h + 1] : idx.heavy_load_buckets.size;
        out[7] = std::max<uint64_t>(out[7], end - heavy_begins[h]);
    }
    out[1] = singletons + out[3] + out[5];
    for (uint32_t p = 0; p < idx.skew_num_partitions && p < 8; ++p) {
        out[8 + p] = idx.skew_mphfs[p].num_keys;
        out[6] += idx.skew_mphfs[p].num_keys;
    }
    out[32] = idx.num_kmers;
    out[33] = idx.num_strings;
    out[34] = idx.num_bases;
    out[35] = idx.skew_num_partitions;
    for (uint64_t s = 0; s < idx.num_strings; ++s) out[36] = std::max(out[36], idx.endpoints[s + 1] - idx.endpoints[s]);
}

std::string index_summary(host_index const& idx) {
    std::ostringstream os;
    const double n = double(idx.num_kmers ? idx.num_kmers : 1);
    os << "k=" << idx.k << " m=" << idx.m << " canonical=" << (idx.canonical ? "true" : "false")
       << " num_kmers=" << idx.num_kmers << " num_strings=" << idx.num_strings << " num_bases=" << idx.num_bases
       << " num_minimizers=" << idx.num_minimizers() << " skew_partitions=" << idx.skew_num_partitions
       << " bits/kmer=" << double(idx.num_bits()) / n << " [strings " << idx.strings.size() * 64.0 / n
       << ", control_codewords " << idx.control_codewords.num_bytes() * 8.0 / n << " (w=" << idx.control_codewords.width
       << "), mphf " << idx.minimizers_mphf.num_bits() / n << ", mid_load " << idx.mid_load_buckets.num_bytes() * 8.0 / n
       << ", endpoints " << idx.endpoints.size() * 64.0 / n << "]";
    return os.str();
}

}  // namespace sshash_amd
