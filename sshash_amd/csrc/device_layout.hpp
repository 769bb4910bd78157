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
// Packed vectors (control codewords, bucket offset lists, pilots, skew positions) keep their
// bit-packed form: one 8-byte read, sometimes two adjacent ones.
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

struct dict_view {
    uint32_t k, m;
    uint32_t canonical;
    uint32_t skew_parts;
    uint64_t hash_magic;
    uint64_t num_kmers, num_strings, num_bases;

    granule const* granules;
    uint64_t const* endpoints;  // num_strings + 1

    mphf_view minimizers;
    uint64_t const* codewords;
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
