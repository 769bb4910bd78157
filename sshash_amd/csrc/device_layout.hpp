// device_layout.hpp -- how the dictionary lives in MI355X HBM.
//
// The host index keeps the reference's components (index.hpp). On upload they are re-laid
// for what bounds this workload on the GPU: the number of *dependent random 64-byte sector
// fetches* per query, not bytes. Two things differ from the host layout:
//
//  (1) `strings` and the string endpoints are fused into 16-byte granules
//          { u32 rank, u32 marks, u64 bases }      one granule = 32 consecutive bases
//      bases : the same 2-bit codes, base 32g+i in bits [2i,2i+1]
//      marks : bit i set iff a string begins at base 32g+i (the final end offset is marked too)
//      rank  : number of marks at bases < 32g
//      so that ONE 32-byte read (two granules; three for k > 31) yields the k-mer to compare,
//      the id of the string it lies in (rank + popcount) and whether it runs across a string
//      boundary -- the information the reference obtains from `read_kmer_at` plus an Elias-Fano
//      `locate` (include/util.hpp:248-257, include/offsets.hpp:138-154: 1 + ~3 dependent misses).
//      Costs 4 bits/base instead of ~2.1; HBM capacity (288 GB) is not the constraint here.
//  (2) the endpoints themselves are kept as a plain u64 array, touched only when the caller
//      asks for the full lookup_result (string_begin/string_end).
//
//  (3) control codewords are widened to one aligned u64 per minimizer id:
//          bits [0, cw_width)   the reference's control codeword, unchanged
//          bits [cw_width, 64)  a fingerprint of the bucket's minimizer (top bits of a multiplicative
//                               hash; canonical indexes fingerprint min(minimizer, its reverse complement))
//      A query whose minimizer is not the bucket's (every negative query, and the first probe of a
//      reverse-complemented positive) is rejected right after the codeword read, without touching
//      the strings: the reference's own minimizer check (spectrum_preserving_string_set.hpp:46-65)
//      answered from the codeword's sector. Equal fingerprints still go through the exact check, so
//      results are unchanged; only the number of sector fetches per miss drops (3.3 -> 2.1).
//      The fingerprints are derived on the GPU at upload time from the strings themselves.
//
// The remaining packed vectors (bucket offset lists, pilots, skew positions) keep their bit-packed
// form: one 8-byte read, sometimes two adjacent ones.
#pragma once

#include "mphf.hpp"

namespace sshash_amd {

struct alignas(16) granule {
    uint32_t rank;
    uint32_t marks;
    uint64_t bases;
};
static_assert(sizeof(granule) == 16, "granule layout");

constexpr uint32_t GRANULE_BASES = 32;
constexpr uint32_t GRANULE_PAD = 4;  // zero granules after the last real one

constexpr uint64_t FINGERPRINT_MUL = 0x9E3779B97F4A7C15ULL;

/* fingerprint of a minimizer, comparable with (entry >> cw_width) */
SSH_HD uint64_t minimizer_fingerprint(uint64_t minimizer, uint32_t m, bool canonical, uint32_t cw_width) {
    uint64_t key = minimizer;
    if (canonical) {
        const uint64_t rc = mmer_revcomp(minimizer, m);
        key = rc < key ? rc : key;
    }
    return cw_width >= 64 ? 0 : (key * FINGERPRINT_MUL) >> cw_width;
}

struct dict_view {
    uint32_t k, m;
    uint32_t canonical;
    uint32_t skew_parts;
    uint64_t hash_magic;
    uint64_t num_kmers, num_strings, num_bases;

    granule const* granules;
    uint64_t const* endpoints;  // num_strings + 1

    mphf_view minimizers;
    uint64_t const* codewords;  // one u64 per minimizer id: code | fingerprint << cw_width
    uint32_t cw_width;
    uint32_t off_width;  // width of the entries of mid_load / heavy_load
    uint32_t const* begin_buckets_of_size;  // 65 entries
    uint64_t const* mid_load;
    uint64_t const* heavy_load;
    uint64_t heavy_size;

    mphf_view skew_f[8];
    uint64_t const* skew_pos[8];
    uint32_t skew_pos_width[8];
};

/* Struct-of-arrays lookup output; any pointer except kmer_id may be null.
   Field meaning: lookup_result, include/util.hpp:38-62. */
struct result_view {
    uint64_t* kmer_id;
    uint64_t* kmer_id_in_string;
    uint64_t* kmer_offset;
    uint64_t* string_id;
    uint64_t* string_begin;
    uint64_t* string_end;
    int8_t* kmer_orientation;
    uint8_t* minimizer_found;
};

}  // namespace sshash_amd
