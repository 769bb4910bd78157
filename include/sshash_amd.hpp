// sshash_amd.hpp -- header-only C++17 facade over the C ABI (sshash_amd.h), shaped like the
// reference's `sshash::dictionary` (reference include/dictionary.hpp:10-181) so that drivers and
// checkers written against the reference read the same: same method names, same argument
// meaning, results by value, errors as std::runtime_error (the reference's only error channel:
// include/util.hpp:191-195, src/query.cpp:128), "not found" == constants::invalid_uint64.
//
// Additions over the reference interface are the batched overloads (the GPU engine wants
// batches) and `to_device`. Everything else is a one-element batch. `streaming_query` below mirrors
// reference include/streaming_query.hpp (one k-mer per call, the reference's shape) and adds the batched
// lookup_read().
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
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

/* the `Kmer` of the reference's typed overloads lookup(Kmer, bool) / is_member(Kmer, bool) (include/dictionary.hpp:42,76):
   a 2-bit packed k-mer, first base in the least-significant bits (include/kmer.hpp:80,194) -- one word for k <= 31,
   two for k <= 63 (the reference's uint64_t / __uint128_t kmer_t) */
struct uint_kmer_t {
    uint64_t bits[2] = {0, 0};
    uint_kmer_t() = default;
    uint_kmer_t(uint64_t lo) : bits{lo, 0} {}
    uint_kmer_t(unsigned __int128 v) : bits{uint64_t(v), uint64_t(v >> 64)} {}
};

/* reference include/util.hpp:76-80: the four forward and the four backward neighbours of a k-mer, indexed by the 2-bit
   code of the added character (alphabet order "ACTG", include/kmer.hpp:118) */
struct neighbourhood {
    std::array<lookup_result, 4> forward, backward;
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
    lookup_result lookup(uint_kmer_t uint_kmer, bool check_reverse_complement = true) const {
        return lookup_batch(uint_kmer.bits, 1, check_reverse_complement)[0];
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

    /* include/dictionary.hpp:48-62, src/dictionary.cpp:111-201: the reference's navigational queries with its own
       signatures -- forward neighbours = suffix + each character, backward = each character + prefix; the *_forward_ /
       *_backward_ variants leave the other half default-constructed (not found), as the reference does */
    neighbourhood kmer_neighbours(uint_kmer_t uint_kmer, bool check_reverse_complement = true) const {
        return neighbours_of(uint_kmer, check_reverse_complement, true, true);
    }
    neighbourhood kmer_neighbours(char const* string_kmer, bool check_reverse_complement = true) const {
        return neighbours_of(pack(string_kmer), check_reverse_complement, true, true);
    }
    neighbourhood kmer_forward_neighbours(uint_kmer_t uint_kmer, bool check_reverse_complement = true) const {
        return neighbours_of(uint_kmer, check_reverse_complement, true, false);
    }
    neighbourhood kmer_forward_neighbours(char const* string_kmer, bool check_reverse_complement = true) const {
        return neighbours_of(pack(string_kmer), check_reverse_complement, true, false);
    }
    neighbourhood kmer_backward_neighbours(uint_kmer_t uint_kmer, bool check_reverse_complement = true) const {
        return neighbours_of(uint_kmer, check_reverse_complement, false, true);
    }
    neighbourhood kmer_backward_neighbours(char const* string_kmer, bool check_reverse_complement = true) const {
        return neighbours_of(pack(string_kmer), check_reverse_complement, false, true);
    }
    neighbourhood string_neighbours(uint64_t string_id, bool check_reverse_complement = true) const {
        lookup_results r;
        sshash_results out = bind(r, 8);
        check(sshash_string_neighbours(m_h, &string_id, 1, check_reverse_complement, &out));
        neighbourhood n;
        for (int c = 0; c < 4; ++c) {
            n.forward[c] = r[c];
            n.backward[c] = r[4 + c];
        }
        return n;
    }

    /* Return the number of kmers in string -- include/dictionary.hpp:44-46 */
    uint64_t string_size(uint64_t string_id) const {
        uint64_t size = 0;
        check(sshash_string_size(m_h, &string_id, 1, &size));
        return size;
    }

    /* [begin, end) of a string in bases -- include/dictionary.hpp:105-108 */
    std::pair<uint64_t, uint64_t> string_offsets(uint64_t string_id) const {
        uint64_t begin = 0, end = 0;
        check(sshash_string_offsets(m_h, &string_id, 1, &begin, &end));
        return {begin, end};
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
    bool is_member(uint_kmer_t uint_kmer, bool check_reverse_complement = true) const {
        uint8_t out = 0;
        check(sshash_is_member_packed(m_h, uint_kmer.bits, 1, check_reverse_complement, &out));
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
        return to_report(s);
    }

    /* streaming_query::lookup for every k-mer of every read, batched (sshash_streaming_lookup): `r` gets one entry per
       base of `bases` -- entry read_offsets[i] + j = the k-mer starting at base j of read i; places where no k-mer starts
       keep a default (not found) result. */
    streaming_query_report streaming_lookup(char const* bases, uint64_t const* read_offsets, uint64_t num_reads, lookup_results& r) const {
        const uint64_t total = num_reads ? read_offsets[num_reads] : 0;
        sshash_results out = bind(r, total);
        out.minimizer_found = nullptr;  // not part of what streaming results are compared on (include/util.hpp:107-141)
        std::fill(r.kmer_id.begin(), r.kmer_id.end(), constants::invalid_uint64);
        std::fill(r.kmer_id_in_string.begin(), r.kmer_id_in_string.end(), constants::invalid_uint64);
        std::fill(r.kmer_offset.begin(), r.kmer_offset.end(), constants::invalid_uint64);
        std::fill(r.string_id.begin(), r.string_id.end(), constants::invalid_uint64);
        std::fill(r.string_begin.begin(), r.string_begin.end(), constants::invalid_uint64);
        std::fill(r.string_end.begin(), r.string_end.end(), constants::invalid_uint64);
        std::fill(r.kmer_orientation.begin(), r.kmer_orientation.end(), int8_t(constants::forward_orientation));
        std::fill(r.minimizer_found.begin(), r.minimizer_found.end(), uint8_t(1));
        sshash_streaming_report s;
        check(sshash_streaming_lookup(m_h, bases, read_offsets, num_reads, &out, &s));
        return to_report(s);
    }

    sshash_dict* handle() const { return m_h; }
    uint32_t words_per_kmer() const { return m_info.words_per_kmer; }

private:
    static streaming_query_report to_report(sshash_streaming_report const& s) {
        streaming_query_report r;
        r.num_kmers = s.num_kmers;
        r.num_positive_kmers = s.num_positive_kmers;
        r.num_negative_kmers = s.num_negative_kmers;
        r.num_invalid_kmers = s.num_invalid_kmers;
        r.num_searches = s.num_searches;
        r.num_extensions = s.num_extensions;
        return r;
    }
    static void check(sshash_status s) {
        if (s != SSHASH_OK) throw std::runtime_error(sshash_last_error());
    }
    void reset() {
        sshash_free(m_h);
        m_h = nullptr;
    }
    /* util::string_to_uint_kmer (include/util.hpp:207-213): k characters, (c >> 1) & 3 each, no validation */
    uint_kmer_t pack(char const* string_kmer) const {
        uint_kmer_t x;
        for (uint64_t i = 0; i < m_info.k; ++i) x.bits[i >> 5] |= uint64_t((string_kmer[i] >> 1) & 3) << (2 * (i & 31));
        return x;
    }
    neighbourhood neighbours_of(uint_kmer_t x, bool check_reverse_complement, bool forward, bool backward) const {
        lookup_results r;
        sshash_results out = bind(r, 8);
        check(sshash_neighbours_packed(m_h, x.bits, 1, check_reverse_complement, &out));
        neighbourhood n;
        for (int c = 0; c < 4; ++c) {
            if (forward) n.forward[c] = r[c];
            if (backward) n.backward[c] = r[4 + c];
        }
        return n;
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

/* streaming_query<Dict, canonical> -- reference include/streaming_query.hpp:9-198: a stateful cursor over the k-mers of
   a read. lookup(kmer) takes a pointer to k characters, the k-mer following the previous call's; reset() starts a new
   read. Same counters, same error (canonical mismatch -> std::runtime_error, :40-45), non-owning pointer to the
   dictionary (:118). The reference asserts that what lookup returns equals the point lookup (:107): here it IS the
   point lookup, with the reference's bookkeeping around it -- a k-mer with an invalid character resets the state
   (:59-65); a positive k-mer is an extension when the previous one was positive in the same string and the id moved by
   the orientation carried along (:86-100), else a search (:144-197). One call = one device round trip: this class is
   the reference's shape; lookup_read() is the form meant for the GPU (one batched call per read). */
template <bool canonical>
struct streaming_query {
    explicit streaming_query(dictionary const* dict) : m_dict(dict), m_k(dict->k()) {
        if (canonical != dict->canonical())
            throw std::runtime_error(std::string("dict.canonical() = ") + (dict->canonical() ? "true" : "false") + " but required " +
                                     (canonical ? "true" : "false"));
        reset();
    }

    void reset() {
        m_res = lookup_result();
        m_positive = false;
    }

    lookup_result lookup(char const* kmer) {
        for (uint64_t i = 0; i != m_k; ++i) {
            if (!is_valid(kmer[i])) {
                m_num_invalid += 1;
                reset();
                return m_res;
            }
        }
        account(m_dict->lookup(kmer));
        return m_res;
    }

    /* every k-mer of `read` (reset() first, as src/query.cpp:78-108 does per read): one batched call */
    std::vector<lookup_result> lookup_read(char const* read, uint64_t length) {
        reset();
        std::vector<lookup_result> out;
        if (length < m_k) return out;
        const uint64_t offsets[2] = {0, length};
        lookup_results r;
        const streaming_query_report rep = m_dict->streaming_lookup(read, offsets, 1, r);
        m_num_searches += rep.num_searches;
        m_num_extensions += rep.num_extensions;
        m_num_negative += rep.num_negative_kmers;
        m_num_invalid += rep.num_invalid_kmers;
        out.reserve(length - m_k + 1);
        for (uint64_t i = 0; i + m_k <= length; ++i) out.push_back(r[i]);
        return out;
    }

    uint64_t num_searches() const { return m_num_searches; }
    uint64_t num_extensions() const { return m_num_extensions; }
    uint64_t num_positive_lookups() const { return num_searches() + num_extensions(); }
    uint64_t num_negative_lookups() const { return m_num_negative; }
    uint64_t num_invalid_lookups() const { return m_num_invalid; }

private:
    static bool is_valid(char c) {  // include/kmer.hpp:209-219,253-255
        switch (c) {
            case 'A': case 'C': case 'G': case 'T': case 'a': case 'c': case 'g': case 't': return true;
            default: return false;
        }
    }
    void account(lookup_result const& now) {
        if (now.kmer_id == constants::invalid_uint64) {
            m_num_negative += 1;
            m_res = now;
            m_positive = false;
            return;
        }
        const bool extension = m_positive && now.string_id == m_res.string_id &&
                               now.kmer_id == m_res.kmer_id + uint64_t(m_res.kmer_orientation);
        const int64_t carried = m_res.kmer_orientation;
        m_res = now;
        if (extension) {
            m_num_extensions += 1;
            m_res.kmer_orientation = carried;  // :94-95
        } else {
            m_num_searches += 1;
        }
        m_positive = true;
    }

    dictionary const* m_dict;
    uint64_t m_k;
    lookup_result m_res;
    bool m_positive = false;
    uint64_t m_num_searches = 0, m_num_extensions = 0, m_num_invalid = 0, m_num_negative = 0;
};

}  // namespace sshash_amd
