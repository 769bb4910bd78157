// index.hpp -- host-side SSHash dictionary: components, construction, (de)serialisation.
//
// Component-for-component mirror of the reference dictionary (include/dictionary.hpp:139-152):
//   strings (2-bit packed superstring)            include/spectrum_preserving_string_set.hpp:202
//   string endpoints                              include/offsets.hpp:115-155
//   minimizers MPHF + control codewords           include/minimizers_control_map.hpp:55-56
//   begin_buckets_of_size, mid_load_buckets       include/sparse_and_skew_index.hpp:148-150
//   skew index: <=8 k-mer MPHFs + positions + heavy_load_buckets   include/sparse_and_skew_index.hpp:66-68
// Control-codeword encoding is the reference's (src/builder/build_sparse_and_skew_index.cpp:110-124,
// 204-235). The builder here is single-file, in-memory and thread-parallel; the reference's
// external-memory pipeline (include/builder/*) is out of scope -- it exists here only because
// no prebuilt index ships with the reference and the engine needs something to query.
#pragma once

#include <string>
#include <vector>

#include "mphf_build.hpp"

namespace sshash_amd {

struct packed_vec {  // bits::compact_vector equivalent
    uint64_t size = 0;
    uint32_t width = 1;
    std::vector<uint64_t> words = std::vector<uint64_t>(1, 0);  // + 1 padding word
    void resize(uint64_t n, uint32_t w) {
        size = n;
        width = w;
        words.assign((n * w + 63) / 64 + 1, 0);
    }
    void set(uint64_t i, uint64_t v) { packed_set(words, i, width, v); }
    uint64_t get(uint64_t i) const { return packed_get(words.data(), i, width); }
    uint64_t num_bytes() const { return words.size() * 8; }
};

struct build_options {
    uint32_t k = 31;
    uint32_t m = 20;            // include/util.hpp:146
    uint64_t seed = 1;          // include/constants.hpp:7
    bool canonical = false;
    uint32_t num_threads = 1;
    double lambda = 5.0;        // include/constants.hpp:10
    bool verbose = false;
    /* minimizer-sharded build: keep only the buckets of minimizers owned by `shard_id`
       (shard_of_minimizer); strings and endpoints stay complete. num_shards = 1: the whole index. */
    uint32_t num_shards = 1;
    uint32_t shard_id = 0;
    /* FASTA headers carry k-mer abundances, '>[id] LN:i:[len] ab:Z:[w w w ...]' (build_configuration::weighted,
       src/builder/encode_strings.cpp:83-135); only build_from_fasta reads them */
    bool weighted = false;
};

struct host_index {
    /* header (include/dictionary.hpp:141-147) */
    uint8_t version[3] = {5, 1, 1};  // include/constants.hpp:22-26
    uint64_t num_kmers = 0;
    uint64_t num_strings = 0;
    uint64_t num_bases = 0;
    uint32_t k = 0, m = 0;
    bool canonical = false;
    uint64_t hash_magic = 0;  // mixer_64::m_magic
    uint64_t build_seed = 0;
    uint32_t num_shards = 1, shard_id = 0;  // which part of the minimizer space this index holds

    /* spectrum-preserving string set */
    std::vector<uint64_t> strings;  // 2 bits/base, LSB first, + zero sentinel words
    uint64_t strings_num_bits = 0;  // including the sentinel (include/util.hpp:253 bound)
    std::vector<uint64_t> endpoints;  // num_strings + 1 base offsets, endpoints[0] == 0

    /* sparse and skew index */
    mphf_host minimizers_mphf;
    packed_vec control_codewords;
    std::vector<uint32_t> begin_buckets_of_size;  // 65 entries
    packed_vec mid_load_buckets;
    uint32_t skew_num_partitions = 0;
    mphf_host skew_mphfs[8];
    packed_vec skew_positions[8];
    packed_vec heavy_load_buckets;

    /* weights (include/weights.hpp): run-length encoded over the k-mer ids -- interval i covers the ids
       [weight_starts[i], weight_starts[i+1]) and has weight weight_values[i]; empty when not weighted */
    std::vector<uint64_t> weight_starts;
    std::vector<uint64_t> weight_values;
    bool weighted() const { return !weight_values.empty(); }

    uint64_t num_minimizers() const { return control_codewords.size; }
    uint64_t num_bits() const;
    uint32_t words_per_kmer() const { return k <= 31 ? 1 : 2; }
};

/* Input: a set of strings already 2-bit packed (no sentinel), with their endpoints. */
void build_from_packed(host_index& idx, std::vector<uint64_t>&& packed_bases,
                       std::vector<uint64_t>&& endpoints, build_options const& opt);

/* Input: FASTA (one header line + one sequence line per record, optionally .gz) as consumed
   by the reference builder (src/builder/encode_strings.cpp:70-176). */
void build_from_fasta(host_index& idx, std::string const& filename, build_options const& opt);

/* Input: sequences in memory (ASCII). */
void build_from_sequences(host_index& idx, std::vector<std::string> const& seqs, build_options const& opt);

void save_index(host_index const& idx, std::string const& filename);
void load_index(host_index& idx, std::string const& filename);

/* access(kmer_id): include/spectrum_preserving_string_set.hpp:114-118 + include/offsets.hpp:41-65.
   Host-side helper used to generate positive queries; writes k chars. */
void access_kmer(host_index const& idx, uint64_t kmer_id, char* out);
/* Packed form: out[0..W) words. */
void access_kmer_packed(host_index const& idx, uint64_t kmer_id, uint64_t* out);

/* weight(kmer_id): include/weights.hpp:147-152 (dictionary::weight, src/dictionary.cpp:96-100) */
uint64_t weight_of(host_index const& idx, uint64_t kmer_id);

std::string index_summary(host_index const& idx);

/* The bucket statistics the reference's builder prints (src/builder/build_sparse_and_skew_index.cpp:64-99,
   include/buckets_statistics.hpp), recomputed from the finished index (so they are available for a loaded file too).
   out[0] minimizers, [1] minimizer positions (= super-k-mers with distinct positions), [2] buckets of 2..64 positions,
   [3] positions in them, [4] buckets in the skew index (> 64 positions), [5] positions in them, [6] k-mers in the skew index,
   [7] largest bucket, [8..15] k-mers per skew partition, [16..31] buckets of exactly 1..16 positions, [32] k-mers,
   [33] strings, [34] bases, [35] skew partitions, [36] longest string, [37..63] zero. */
constexpr uint32_t BUCKET_STATS_WORDS = 64;
void bucket_statistics(host_index const& idx, uint64_t* out);

}  // namespace sshash_amd
