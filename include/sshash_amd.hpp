// sshash_amd.hpp -- header-only C++17 facade over the C ABI (sshash_amd.h), shaped like the
// reference's `sshash::dictionary` (reference include/dictionary.hpp:10-181) so that drivers and
// checkers written against the reference read the same: same method names, same argument
// meaning, results by value, errors as std::runtime_error (the reference's only error channel:
// include/util.hpp:191-195, src/query.cpp:128), "not found" == constants::invalid_uint64.
//
// Additions over the reference interface are the batched overloads (the GPU engine wants
// batches) and `to_device`. Everything else is a one-element batch.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "sshash_amd.h"

namespace sshash_amd {

namespace constants {  // reference include/constants.hpp:5,17-18
constexpr uint64_t invalid_uint64 = uint64_t(-1);
constexpr int forward_orientation = 1;
constexpr int backward_orientation = -1;
}  // namespace constants

struct lookup_result {  // reference include/util.hpp:38-62
    uint64_t kmer_id = constants::invalid_uint64;
    uint64_t kmer_id_in_string = constants::invalid_uint64;
    uint64_t kmer_offset = constants::invalid_uint64;
    int64_t kmer_orientation = constants::forward_orientation;
    uint64_t string_id = constants::invalid_uint64;
    uint64_t string_begin = constants::invalid_uint64;
    uint64_t string_end = constants::invalid_uint64;
    bool minimizer_found = true;
};

struct streaming_query_report {  // reference include/util.hpp:21-36
    uint64_t num_kmers = 0, num_positive_kmers = 0, num_negative_kmers = 0, num_invalid_kmers = 0, num_searches = 0,
             num_extensions = 0;
};

struct build_configuration {  // reference include/util.hpp:143-159
    uint64_t k = 31, m = 20, seed = 1, num_threads = 1;
    double lambda = 5.0;
    bool canonical = false, verbose = false;
};

/* struct-of-arrays batch result */
struct lookup_results {
    std::vector<uint64_t> kmer_id, kmer_id_in_string, kmer_offset, string_id, string_begin, string_end;
    std::vector<int8_t> kmer_orientation;
    std::vector<uint8_t> minimizer_found;
    size_t size() const { return kmer_id.size(); }
    lookup_result operator[](size_t i) const {
        lookup_result r;
        r.kmer_id = kmer_id[i];
        r.kmer_id_in_string = kmer_id_in_string[i];
        r.kmer_offset = kmer_offset[i];
        r.kmer_orientation = kmer_orientation[i];
        r.string_id = string_id[i];
        r.string_begin = string_begin[i];
        r.string_end = string_end[i];
        r.minimizer_found = minimizer_found[i] != 0;
        return r;
    }
};

class dictionary {
public:
    dictionary() = default;
    dictionary(dictionary const&) = delete;
    dictionary& operator=(dictionary const&) = delete;
    dictionary(dictionary&& o) noexcept : m_h(o.m_h), m_info(o.m_info) { o.m_h = nullptr; }
    ~dictionary() { sshash_free(m_h); }

    /* dictionary::build(input_filename, build_config) -- include/dictionary.hpp:28 */
    void build(std::string const& input_filename, build_configuration const& c) {
        sshash_build_config cfg;
        sshash_build_config_default(&cfg);
        cfg.k = uint32_t(c.k);
        cfg.m = uint32_t(c.m);
        cfg.seed = c.seed;
        cfg.canonical = c.canonical;
        cfg.num_threads = uint32_t(c.num_threads);
        cfg.lambda = c.lambda;
        cfg.verbose = c.verbose;
        reset();
        check(sshash_build_from_fasta(input_filename.c_str(), &cfg, &m_h));
        check(sshash_get_info(m_h, &m_info));
    }
    /* essentials::load(dict, filename) / essentials::save -- tools/common.hpp:19-22, tools/build.cpp:90-95 */
    void load(std::string const& index_filename) {
        reset();
        check(sshash_load(index_filename.c_str(), &m_h));
        check(sshash_get_info(m_h, &m_info));
    }
    void save(std::string const& index_filename) const { check(sshash_save(m_h, index_filename.c_str())); }

    /* no reference counterpart: make the dictionary resident in the HBM of `device` */
    void to_device(int device = 0) { check(sshash_to_device(m_h, device)); }

    uint64_t num_kmers() const { return m_info.num_kmers; }
    uint64_t num_strings() const { return m_info.num_strings; }
    uint64_t k() const { return m_info.k; }
    uint64_t m() const { return m_info.m; }
    bool canonical() const { return m_info.canonical != 0; }
    uint64_t num_bits() const { return m_info.num_bits; }

    /* Lookup queries -- include/dictionary.hpp:40-42 */
    lookup_result lookup(char const* string_kmer, bool check_reverse_complement = true) const {
        return lookup_batch(string_kmer, 1, check_reverse_complement)[0];
    }
    lookup_result lookup_packed(uint64_t const* uint_kmer_words, bool check_reverse_complement = true) const {
        return lookup_batch(uint_kmer_words, 1, check_reverse_complement)[0];
    }
    /* batched: n k-mers of k chars back to back (no terminators) */
    lookup_results lookup_batch(char const* kmers, uint64_t n, bool check_reverse_complement = true) const {
        lookup_results r;
        sshash_results out = bind(r, n);
        check(sshash_lookup_ascii(m_h, kmers, n, check_reverse_complement, &out));
        return r;
    }
    /* batched: n packed k-mers, words_per_kmer words each */
    lookup_results lookup_batch(uint64_t const* kmers, uint64_t n, bool check_reverse_complement = true) const {
        lookup_results r;
        sshash_results out = bind(r, n);
        check(sshash_lookup_packed(m_h, kmers, n, check_reverse_complement, &out));
        return r;
    }
    std::vector<uint64_t> lookup_ids(uint64_t const* kmers, uint64_t n, bool check_reverse_complement = true) const {
        std::vector<uint64_t> ids(n);
        sshash_results out{};
        out.kmer_id = ids.data();
        check(sshash_lookup_packed(m_h, kmers, n, check_reverse_complement, &out));
        return ids;
    }

    /* Navigational query -- include/dictionary.hpp:59-61 (kmer_neighbours): ids of the forward neighbours
       suffix + A,C,T,G at [8*i .. 8*i+3] and of the backward neighbours A,C,T,G + prefix at [8*i+4 .. 8*i+7]
       (the reference's alphabet order "ACTG": index = 2-bit code of the character)
       of packed k-mer i; INVALID for a neighbour that is not in the dictionary. */
    std::vector<uint64_t> neighbours_ids(uint64_t const* kmers, uint64_t n, bool check_reverse_complement = true) const {
        std::vector<uint64_t> ids(8 * n);
        sshash_results out{};
        out.kmer_id = ids.data();
        check(sshash_neighbours_packed(m_h, kmers, n, check_reverse_complement, &out));
        return ids;
    }

    /* Return the number of kmers in string -- include/dictionary.hpp:44-46 */
    uint64_t string_size(uint64_t string_id) const {
        uint64_t size = 0;
        check(sshash_string_size(m_h, &string_id, 1, &size));
        return size;
    }

    /* Return the weight of the kmer given its id -- include/dictionary.hpp (weight), src/dictionary.cpp:96-100 */
    bool weighted() const { return m_info.weighted != 0; }
    uint64_t weight(uint64_t kmer_id) const {
        uint64_t w = 0;
        check(sshash_weight(m_h, &kmer_id, 1, &w));
        return w;
    }

    /* include/dictionary.hpp:62 (string_neighbours): same layout, forward neighbours of the string's last k-mer and
       backward neighbours of its first one */
    std::vector<uint64_t> string_neighbours_ids(uint64_t const* string_ids, uint64_t n, bool check_reverse_complement = true) const {
        std::vector<uint64_t> ids(8 * n);
        sshash_results out{};
        out.kmer_id = ids.data();
        check(sshash_string_neighbours(m_h, string_ids, n, check_reverse_complement, &out));
        return ids;
    }

    /* Membership queries -- include/dictionary.hpp:74-76 */
    bool is_member(char const* string_kmer, bool check_reverse_complement = true) const {
        uint8_t out = 0;
        check(sshash_is_member_ascii(m_h, string_kmer, 1, check_reverse_complement, &out));
        return out != 0;
    }
    std::vector<uint8_t> is_member_batch(char const* kmers, uint64_t n, bool check_reverse_complement = true) const {
        std::vector<uint8_t> out(n);
        check(sshash_is_member_ascii(m_h, kmers, n, check_reverse_complement, out.data()));
        return out;
    }

    /* Return the string of the kmer whose id is kmer_id -- include/dictionary.hpp:68-69 */
    void access(uint64_t kmer_id, char* string_kmer) const { check(sshash_access(m_h, kmer_id, string_kmer)); }

    /* include/dictionary.hpp:81-82 */
    streaming_query_report streaming_query_from_file(std::string const& filename, bool multiline) const {
        sshash_streaming_report s;
        check(sshash_streaming_query_from_file(m_h, filename.c_str(), multiline, &s));
        streaming_query_report r;
        r.num_kmers = s.num_kmers;
        r.num_positive_kmers = s.num_positive_kmers;
        r.num_negative_kmers = s.num_negative_kmers;
        r.num_invalid_kmers = s.num_invalid_kmers;
        r.num_searches = s.num_searches;
        r.num_extensions = s.num_extensions;
        return r;
    }

    sshash_dict* handle() const { return m_h; }
    uint32_t words_per_kmer() const { return m_info.words_per_kmer; }

private:
    static void check(sshash_status s) {
        if (s != SSHASH_OK) throw std::runtime_error(sshash_last_error());
    }
    void reset() {
        sshash_free(m_h);
        m_h = nullptr;
    }
    static sshash_results bind(lookup_results& r, uint64_t n) {
        r.kmer_id.resize(n);
        r.kmer_id_in_string.resize(n);
        r.kmer_offset.resize(n);
        r.string_id.resize(n);
        r.string_begin.resize(n);
        r.string_end.resize(n);
        r.kmer_orientation.resize(n);
        r.minimizer_found.resize(n);
        sshash_results out;
        out.kmer_id = r.kmer_id.data();
        out.kmer_id_in_string = r.kmer_id_in_string.data();
        out.kmer_offset = r.kmer_offset.data();
        out.string_id = r.string_id.data();
        out.string_begin = r.string_begin.data();
        out.string_end = r.string_end.data();
        out.kmer_orientation = r.kmer_orientation.data();
        out.minimizer_found = r.minimizer_found.data();
        return out;
    }
    sshash_dict* m_h = nullptr;
    sshash_info m_info{};
};

}  // namespace sshash_amd
