// check_dictionary.cpp -- the reference's self-consistency checkers, re-expressed over the C++ facade
// (include/sshash_amd.hpp) with batched calls. What is checked, and where the reference checks it:
//   [A] streaming the build input: every other sequence lower-cased, every other k-mer
//       reverse-complemented -> kmer_id == running counter, orientation, kmer_id_in_string /
//       string_id sequencing, string size, access round trip, is_member
//       (reference test/check_from_file.hpp:9-171)
//   [B] for every id: access(id) -> lookup -> same id, is_member      (reference test/check.hpp:7-76)
//   [C] random negative lookups                                        (reference test/check.hpp:78-96)
// Usage: check_dictionary <input.fa[.gz]> <k> <m> [--canonical]
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "sshash_amd.hpp"

using namespace sshash_amd;

static std::string reverse_complement(std::string const& s) {
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[s.size() - 1 - i], o = 0;
        switch (c) {
            case 'A': o = 'T'; break;  case 'C': o = 'G'; break;  case 'G': o = 'C'; break;  case 'T': o = 'A'; break;
            case 'a': o = 't'; break;  case 'c': o = 'g'; break;  case 'g': o = 'c'; break;  case 't': o = 'a'; break;
        }
        r[i] = o;
    }
    return r;
}

static std::vector<std::string> read_sequences(std::string const& filename, uint64_t k) {
    gzFile f = gzopen(filename.c_str(), "rb");
    if (!f) throw std::runtime_error("error in opening the file '" + filename + "'");
    std::vector<std::string> seqs;
    std::vector<char> buf(1 << 20);
    std::string line;
    bool header = true;
    auto flush = [&](bool complete) {
        if (!header && complete && line.size() >= k) seqs.push_back(line);
        if (complete) header = !header;
        line.clear();
    };
    while (gzgets(f, buf.data(), int(buf.size()))) {
        size_t n = strlen(buf.data());
        bool complete = n && buf[n - 1] == '\n';
        line.append(buf.data(), complete ? n - 1 : n);
        if (complete) flush(true);
    }
    gzclose(f);
    return seqs;  // an unterminated last line is not a record (as the builder)
}

#define FAIL(msg)                                   \
    do {                                            \
        std::cout << "ERROR: " << msg << std::endl; \
        return false;                               \
    } while (0)

static bool check_lookup_access(dictionary const& dict, std::vector<std::string> const& seqs) {
    const uint64_t k = dict.k();
    std::cout << "checking correctness of access, positive lookup, and membership..." << std::endl;
    uint64_t num_kmers = 0, num_sequences = 0;
    lookup_result prev;
    prev.string_id = 0;
    std::string batch;
    std::vector<int> expected_orientation;
    auto run = [&]() -> bool {
        const uint64_t n = expected_orientation.size();
        if (!n) return true;
        auto res = dict.lookup_batch(batch.data(), n);
        auto member = dict.is_member_batch(batch.data(), n);
        std::string got(k, 0);
        for (uint64_t i = 0; i < n; ++i, ++num_kmers) {
            const lookup_result curr = res[i];
            if (curr.kmer_id != num_kmers) FAIL("wrong id assigned: got " << curr.kmer_id << " expected " << num_kmers);
            if (curr.kmer_orientation != expected_orientation[i]) FAIL("got orientation " << curr.kmer_orientation);
            const uint64_t size = curr.string_end - curr.string_begin - k + 1;
            if (curr.kmer_id_in_string >= size) FAIL("kmer_id_in_string out of range");
            if (num_kmers == 0) {
                if (curr.string_id != 0) FAIL("first string_id must be 0");
            } else if (curr.string_id == prev.string_id) {
                if (curr.kmer_id_in_string != prev.kmer_id_in_string + 1) FAIL("kmer_id_in_string not sequential");
                if (curr.string_end != prev.string_end || curr.string_begin != prev.string_begin) FAIL("string bounds changed");
            } else {
                if (curr.string_id != prev.string_id + 1) FAIL("string_id not sequential");
                if (curr.kmer_id_in_string != 0) FAIL("kmer_id_in_string must restart at 0");
            }
            prev = curr;
            if (!member[i]) FAIL("is_member false for an indexed k-mer");
            if (num_kmers % 997 == 0) {  // access round trip on a sample (one host call each)
                dict.access(curr.kmer_id, got.data());
                std::string q(batch.data() + i * k, k);
                std::transform(q.begin(), q.end(), q.begin(), [](unsigned char c) { return char(std::toupper(c)); });
                if (got != q && got != reverse_complement(q)) FAIL("access(" << curr.kmer_id << ") = " << got << " but looked up " << q);
            }
        }
        batch.clear();
        expected_orientation.clear();
        return true;
    };
    uint64_t counter = 0;
    for (std::string sequence : seqs) {
        if ((num_sequences & 1) == 0) std::transform(sequence.begin(), sequence.end(), sequence.begin(), [](unsigned char c) { return char(std::tolower(c)); });
        ++num_sequences;
        for (uint64_t i = 0; i + k <= sequence.size(); ++i, ++counter) {
            std::string kmer = sequence.substr(i, k);
            int orientation = constants::forward_orientation;
            if ((counter & 1) == 0) {
                kmer = reverse_complement(kmer);
                orientation = constants::backward_orientation;
            }
            batch += kmer;
            expected_orientation.push_back(orientation);
        }
        if (expected_orientation.size() >= (1u << 20) && !run()) return false;
    }
    if (!run()) return false;
    if (num_kmers != dict.num_kmers()) FAIL("checked " << num_kmers << " k-mers but the dictionary holds " << dict.num_kmers());
    std::cout << "checked " << num_kmers << " kmers" << std::endl;
    return true;
}

static bool check_every_id(dictionary const& dict) {
    std::cout << "checking correctness of access and positive lookup for every id..." << std::endl;
    const uint64_t k = dict.k(), n = dict.num_kmers(), chunk = 1 << 20;
    std::string batch;
    for (uint64_t begin = 0; begin < n; begin += chunk) {
        const uint64_t m = std::min(chunk, n - begin);
        batch.assign(m * k, 0);
        for (uint64_t i = 0; i < m; ++i) dict.access(begin + i, &batch[i * k]);
        auto res = dict.lookup_batch(batch.data(), m);
        auto member = dict.is_member_batch(batch.data(), m);
        for (uint64_t i = 0; i < m; ++i) {
            if (res.kmer_id[i] == constants::invalid_uint64) FAIL("kmer with id " << begin + i << " not found");
            if (res.kmer_id[i] != begin + i) FAIL("expected id " << begin + i << " but got id " << res.kmer_id[i]);
            if (!member[i]) FAIL("id " << begin + i << " not found by is_member");
        }
    }
    return true;
}

static bool check_negative(dictionary const& dict) {
    std::cout << "checking correctness of negative lookup with random kmers..." << std::endl;
    const uint64_t k = dict.k(), n = std::min<uint64_t>(1000000, dict.num_kmers());
    std::string batch(n * k, 'A');
    srand(42);
    for (auto& c : batch) c = "ACGT"[rand() % 4];
    auto res = dict.lookup_batch(batch.data(), n);
    uint64_t found = 0;
    for (uint64_t i = 0; i < n; ++i) found += res.kmer_id[i] != constants::invalid_uint64;
    /* the reference only prints when a random k-mer is found (it has no ground truth at hand);
       for k >= 21 a hit is astronomically unlikely, so treat it as an error */
    if (found && k >= 21) FAIL(found << " random kmers found");
    return true;
}

/* test/check_from_file.hpp:174-221 (kmer_neighbours along the input) and test/check.hpp:99-141 (string_neighbours
   = backward neighbours of a string's first k-mer + forward neighbours of its last one), batched */
static bool check_navigational(dictionary const& dict, std::vector<std::string> const& seqs) {
    std::cout << "checking correctness of navigational queries for kmers and strings..." << std::endl;
    const uint64_t k = dict.k(), W = dict.words_per_kmer();
    auto code = [](char c) { return (uint64_t(uint8_t(c)) >> 1) & 3; };
    uint64_t kmer_id = 0;
    for (uint64_t s = 0; s < seqs.size(); ++s) {
        std::string const& seq = seqs[s];
        const uint64_t n = seq.size() - k + 1;
        if (dict.string_size(s) != n) FAIL("string_size of string " << s);
        std::vector<uint64_t> packed(n * W, 0);
        for (uint64_t i = 0; i < n; ++i)
            for (uint64_t j = 0; j < k; ++j) packed[i * W + (2 * j) / 64] |= code(seq[i + j]) << ((2 * j) % 64);
        const std::vector<uint64_t> nb = dict.neighbours_ids(packed.data(), n);
        for (uint64_t i = 0; i < n; ++i) {
            if (i + 1 < n && nb[8 * i + code(seq[i + k])] != kmer_id + i + 1) FAIL("expected forward[" << seq[i + k] << "]");
            if (i > 0 && nb[8 * i + 4 + code(seq[i - 1])] != kmer_id + i - 1) FAIL("expected backward[" << seq[i - 1] << "]");
        }
        const std::vector<uint64_t> sn = dict.string_neighbours_ids(&s, 1);
        for (uint64_t c = 0; c < 4; ++c) {
            if (sn[c] != nb[8 * (n - 1) + c]) FAIL("string_neighbours forward of string " << s);
            if (sn[4 + c] != nb[4 + c]) FAIL("string_neighbours backward of string " << s);
        }
        if (s < 20) {
            /* the reference's own shapes (include/dictionary.hpp:48-62): neighbourhood structs, one k-mer per call, every
               field a lookup of the neighbouring k-mer string (src/dictionary.cpp:111-126) */
            const uint64_t mid = n / 2;
            const neighbourhood both = dict.kmer_neighbours(seq.c_str() + mid);
            const neighbourhood fwd = dict.kmer_forward_neighbours(seq.c_str() + mid), bwd = dict.kmer_backward_neighbours(seq.c_str() + mid);
            uint_kmer_t x;
            for (uint64_t j = 0; j < k; ++j) x.bits[(2 * j) / 64] |= code(seq[mid + j]) << ((2 * j) % 64);
            const neighbourhood typed = dict.kmer_neighbours(x);
            const neighbourhood of_string = dict.string_neighbours(s);
            for (uint64_t c = 0; c < 4; ++c) {
                const std::string next = seq.substr(mid + 1, k - 1) + "ACTG"[c], prev = std::string(1, "ACTG"[c]) + seq.substr(mid, k - 1);
                const lookup_result f = dict.lookup(next.c_str()), b = dict.lookup(prev.c_str());
                auto same = [](lookup_result const& p, lookup_result const& q) {
                    return p.kmer_id == q.kmer_id && p.kmer_id_in_string == q.kmer_id_in_string && p.kmer_offset == q.kmer_offset &&
                           p.kmer_orientation == q.kmer_orientation && p.string_id == q.string_id && p.string_begin == q.string_begin &&
                           p.string_end == q.string_end;
                };
                if (!same(both.forward[c], f) || !same(both.backward[c], b)) FAIL("kmer_neighbours(char const*) of string " << s);
                if (!same(typed.forward[c], f) || !same(typed.backward[c], b)) FAIL("kmer_neighbours(Kmer) of string " << s);
                if (!same(fwd.forward[c], f) || fwd.backward[c].kmer_id != constants::invalid_uint64) FAIL("kmer_forward_neighbours of string " << s);
                if (!same(bwd.backward[c], b) || bwd.forward[c].kmer_id != constants::invalid_uint64) FAIL("kmer_backward_neighbours of string " << s);
                if (of_string.forward[c].kmer_id != sn[c] || of_string.backward[c].kmer_id != sn[4 + c]) FAIL("string_neighbours of string " << s);
            }
            if (mid + 1 < n && both.forward[code(seq[mid + k])].kmer_id != kmer_id + mid + 1) FAIL("forward neighbour along the string");
        }
        kmer_id += n;
        if (s >= 400) break;  // the batched form makes one round trip per string: a prefix of the file is enough
    }
    return true;
}

/* include/streaming_query.hpp: the class, one k-mer per call (the reference's shape), must agree with the point lookup
   on what equal_lookup_result compares (include/util.hpp:107-141; the reference asserts it, :107), with the batched
   lookup_read() of the same read, and the two must count alike; the typed overloads lookup(Kmer) / is_member(Kmer)
   (include/dictionary.hpp:42,76) must agree with the string ones. */
template <bool canonical>
static bool check_streaming_query(dictionary const& dict, std::vector<std::string> const& seqs) {
    std::cout << "checking correctness of streaming_query and of the typed lookups..." << std::endl;
    const uint64_t k = dict.k();
    auto same = [](lookup_result const& a, lookup_result const& b) {
        return a.kmer_id == b.kmer_id && a.kmer_id_in_string == b.kmer_id_in_string && a.string_id == b.string_id &&
               a.string_begin == b.string_begin && a.string_end == b.string_end &&
               (b.kmer_id == constants::invalid_uint64 || a.kmer_orientation == b.kmer_orientation);
    };
    /* a read: a stretch of one string, a mutation, an N, a stretch of another string reverse-complemented, junk */
    std::string read = seqs[0].substr(0, std::min<size_t>(seqs[0].size(), k + 20));
    read[read.size() / 2] = read[read.size() / 2] == 'A' ? 'C' : 'A';
    read += "N";
    read += reverse_complement(seqs[1 % seqs.size()].substr(0, std::min<size_t>(seqs[1 % seqs.size()].size(), k + 12)));
    read += "ACGTTGCAACGTACGTAGCTAGCTAGCATCGATCGATCAGCTAGCTAGCATCGATCAGCTACGAT";
    streaming_query<canonical> one_by_one(&dict), batched(&dict);
    one_by_one.reset();
    std::vector<lookup_result> all = batched.lookup_read(read.data(), read.size());
    if (all.size() != read.size() - k + 1) FAIL("lookup_read returned " << all.size() << " results");
    for (uint64_t i = 0; i + k <= read.size(); ++i) {
        const lookup_result got = one_by_one.lookup(read.data() + i);
        bool valid = true;
        for (uint64_t j = 0; j < k; ++j) valid = valid && std::string("ACGTacgt").find(read[i + j]) != std::string::npos;
        lookup_result point = valid ? dict.lookup(read.data() + i) : lookup_result();
        if (!same(got, point)) FAIL("streaming_query::lookup differs from the point lookup at k-mer " << i);
        if (!same(all[i], point)) FAIL("lookup_read differs from the point lookup at k-mer " << i);
        if (valid) {
            /* typed overloads: the same k-mer, packed */
            uint_kmer_t x;
            for (uint64_t j = 0; j < k; ++j) x.bits[(2 * j) / 64] |= ((uint64_t(uint8_t(read[i + j])) >> 1) & 3) << ((2 * j) % 64);
            lookup_result typed = dict.lookup(x);
            if (typed.kmer_id != point.kmer_id || (typed.kmer_id != constants::invalid_uint64 && typed.kmer_orientation != point.kmer_orientation))
                FAIL("lookup(Kmer) differs from lookup(char const*) at k-mer " << i);
            if (dict.is_member(x) != (point.kmer_id != constants::invalid_uint64)) FAIL("is_member(Kmer) at k-mer " << i);
            if (dict.is_member(read.data() + i) != dict.is_member(x)) FAIL("is_member(char const*) at k-mer " << i);
        }
    }
    if (one_by_one.num_searches() != batched.num_searches() || one_by_one.num_extensions() != batched.num_extensions() ||
        one_by_one.num_negative_lookups() != batched.num_negative_lookups() || one_by_one.num_invalid_lookups() != batched.num_invalid_lookups())
        FAIL("counters of the two forms differ: searches " << one_by_one.num_searches() << "/" << batched.num_searches() << " extensions "
                                                          << one_by_one.num_extensions() << "/" << batched.num_extensions());
    if (one_by_one.num_extensions() == 0 || one_by_one.num_invalid_lookups() == 0 || one_by_one.num_negative_lookups() == 0)
        FAIL("the read was meant to exercise extensions, invalid and negative k-mers");
    if (one_by_one.num_positive_lookups() + one_by_one.num_negative_lookups() + one_by_one.num_invalid_lookups() != read.size() - k + 1)
        FAIL("counters do not add up (src/query.cpp:44-45)");
    /* canonical mismatch -> std::runtime_error (include/streaming_query.hpp:40-45) */
    try {
        streaming_query<!canonical> wrong(&dict);
        FAIL("a streaming_query of the other flavour was accepted");
    } catch (std::runtime_error const& e) {
        if (std::string(e.what()).find("but required") == std::string::npos) FAIL("unexpected message: " << e.what());
    }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::cerr << "Usage: " << argv[0] << " <input.fa[.gz]> <k> <m> [--canonical]" << std::endl;
        return 2;
    }
    try {
        build_configuration cfg;
        cfg.k = std::stoull(argv[2]);
        cfg.m = std::stoull(argv[3]);
        cfg.canonical = argc > 4 && std::string(argv[4]) == "--canonical";
        cfg.num_threads = 8;
        dictionary dict;
        dict.build(argv[1], cfg);
        dict.to_device(0);
        auto seqs = read_sequences(argv[1], cfg.k);
        if (!check_lookup_access(dict, seqs)) return 1;
        if (!check_every_id(dict)) return 1;
        if (!check_negative(dict)) return 1;
        if (!check_navigational(dict, seqs)) return 1;
        if (!(cfg.canonical ? check_streaming_query<true>(dict, seqs) : check_streaming_query<false>(dict, seqs))) return 1;
        /* error channel: exceptions with the reference's wording */
        try {
            dictionary other;
            other.load("/nonexistent/index.sshash");
            std::cout << "ERROR: loading a missing file did not throw" << std::endl;
            return 1;
        } catch (std::runtime_error const& e) {
            if (std::string(e.what()).find("error in opening the file") == std::string::npos) return 1;
        }
        std::cout << "EVERYTHING OK!" << std::endl;
    } catch (std::exception const& e) {
        std::cout << "EXCEPTION: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
