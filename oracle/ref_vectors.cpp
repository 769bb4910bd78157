// ref_vectors.cpp -- golden-vector generator.  TEST INFRASTRUCTURE ONLY.
//
// Compiles against the two pieces of the reference that build standalone -- include/kmer.hpp
// and external/cityhash/cityhash.{hpp,cpp} -- where they lie under the reference checkout
// (nothing is copied into this repo) and prints JSON with inputs and the reference's outputs:
//   * 2-bit encoding of ASCII k-mers           (uint_kmer_t::set + char_to_uint, kmer.hpp:80,194)
//   * reverse complements, 64- and 128-bit     (dna_uint_kmer_t::reverse_complement_inplace, kmer.hpp:159-165)
//   * string reverse complement / validity     (kmer.hpp:245-255)
//   * CityHash128WithSeed of 8- and 16-byte keys with seed pair {s, ~s}  (hash_util.hpp:12-16,62-66)
// The binary is built into oracle/_ref/ by oracle/Makefile when the reference checkout exists;
// its output is committed as tests/golden/ref_vectors.json by tests/golden/make_golden.py.
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "include/kmer.hpp"
#include "external/cityhash/cityhash.hpp"

using namespace sshash;

static uint64_t rng_state = 0x5555AAAA12345678ULL;
static uint64_t next_u64() {  // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

template <typename K>
static K encode(std::string const& s) {
    K x = 0;
    for (uint64_t i = 0; i != s.size(); ++i) x.set(i, K::char_to_uint(s[i]));
    return x;
}

int main() {
    using k64 = dna_uint_kmer_t<uint64_t>;
    using k128 = dna_uint_kmer_t<__uint128_t>;
    const char letters[] = "ACGTacgt";
    printf("{\n\"kmers\": [\n");
    bool first = true;
    for (uint32_t k : {1u, 2u, 5u, 13u, 21u, 30u, 31u, 32u, 33u, 47u, 62u, 63u}) {
        for (int rep = 0; rep < 12; ++rep) {
            std::string s(k, 'A');
            for (auto& c : s) c = letters[next_u64() % 8];
            std::string rc(k, 0);
            k64::compute_reverse_complement(s.data(), rc.data(), k);
            k128 x = encode<k128>(s);
            k128 r = x;
            r.reverse_complement_inplace(k);
            printf("%s{\"k\": %u, \"s\": \"%s\", \"rc_s\": \"%s\", \"lo\": \"%016llx\", \"hi\": \"%016llx\", "
                   "\"rc_lo\": \"%016llx\", \"rc_hi\": \"%016llx\"",
                   first ? "" : ",\n", k, s.c_str(), rc.c_str(), (unsigned long long)uint64_t(x.bits),
                   (unsigned long long)uint64_t(x.bits >> 64), (unsigned long long)uint64_t(r.bits),
                   (unsigned long long)uint64_t(r.bits >> 64));
            if (k <= 31) {
                k64 y = encode<k64>(s);
                k64 ry = y;
                ry.reverse_complement_inplace(k);
                printf(", \"w1\": \"%016llx\", \"rc_w1\": \"%016llx\"", (unsigned long long)y.bits, (unsigned long long)ry.bits);
            }
            printf("}");
            first = false;
        }
    }
    printf("\n],\n\"valid_chars\": \"");
    for (int c = 1; c < 128; ++c)
        if (k64::is_valid(char(c))) printf("%c", c);
    printf("\",\n\"city128\": [\n");
    first = true;
    const uint64_t seeds[] = {1234567890ULL, ~1234567890ULL, 0ULL, 1ULL, 0xDEADBEEFCAFEF00DULL};
    for (uint64_t seed : seeds) {
        for (int rep = 0; rep < 10; ++rep) {
            uint64_t key[2] = {rep == 0 ? 0x0123456789abcdefULL : next_u64(), next_u64()};
            if (rep == 1) key[0] = 0;
            auto h8 = cityhash::CityHash128WithSeed(reinterpret_cast<char const*>(key), 8, {seed, ~seed});
            auto h16 = cityhash::CityHash128WithSeed(reinterpret_cast<char const*>(key), 16, {seed, ~seed});
            printf("%s{\"seed\": \"%016llx\", \"k0\": \"%016llx\", \"k1\": \"%016llx\", \"h8\": [\"%016llx\", \"%016llx\"], "
                   "\"h16\": [\"%016llx\", \"%016llx\"]}",
                   first ? "" : ",\n", (unsigned long long)seed, (unsigned long long)key[0], (unsigned long long)key[1],
                   (unsigned long long)h8.first, (unsigned long long)h8.second, (unsigned long long)h16.first,
                   (unsigned long long)h16.second);
            first = false;
        }
    }
    printf("\n]\n}\n");
    return 0;
}
