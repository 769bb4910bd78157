// mphf.hpp -- minimal perfect hash function: evaluation (host + device) and layout.
//
// The reference evaluates `pthash::partitioned_phf<city_hasher_128, opt_bucketer, compact, true>`
// (include/hash_util.hpp:39-45, call sites include/minimizers_control_map.hpp:37 and
// include/sparse_and_skew_index.hpp:41). The PTHash sources are NOT part of the reference
// checkout (external/pthash is an empty submodule), so this file restates the *published*
// PTHash scheme (Pibiri & Trani, "PTHash: Revisiting FCH Minimal Perfect Hashing", SIGIR'21;
// partitioned variant: "Parallel and External-Memory Construction of Minimal Perfect Hash
// Functions with PTHash", TKDE'23):
//
//     (h1,h2) = CityHash128WithSeed(key, seed)              base hasher, as the reference
//     part    = range(h1 ^ h2, P)                           uniform partitioner
//     bucket  = skew(h1): 60% of keys -> 30% of buckets     the paper's bucketer (integer only;
//                                                           PTHash's newer `opt_bucketer` goes
//                                                           through double-precision log, which
//                                                           is a bit-exactness hazard on a GPU)
//     pilot   = pilots[part.pilot_base + bucket]            fixed-width ("compact") encoding
//     p       = position(h2, pilot, table_size)             XOR displacement
//     p       = p < n ? p : free_slots[p - n]               minimal output
//     id      = part.key_offset + p
//
// MPHF *values* never leak into lookup results (k-mer ids are defined by input order,
// include/spectrum_preserving_string_set.hpp:227), so any correct MPHF gives identical
// results; what must match the reference exactly is the base hasher, restated below from
// the vendored CityHash (external/cityhash/cityhash.cpp:116-135,238-269) for the two key
// sizes the reference ever hashes: 8 bytes (minimizers, k-mers with k<=31) and 16 bytes
// (k-mers with k<=63) -- include/hash_util.hpp:9-17,58-67.
#pragma once

#include "kmer_core.hpp"

namespace sshash_amd {

struct hash128 {
    uint64_t first, second;
};

namespace city {
constexpr uint64_t K1 = 0xb492b66fbe98f273ULL;
constexpr uint64_t KMUL = 0x9ddfea08eb382d69ULL;

SSH_HD uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }

/* Hash128to64 (external/cityhash/cityhash.hpp:90-99) */
SSH_HD uint64_t len16(uint64_t u, uint64_t v) {
    uint64_t a = (u ^ v) * KMUL;
    a ^= (a >> 47);
    uint64_t b = (v ^ a) * KMUL;
    b ^= (b >> 47);
    return b * KMUL;
}

/* CityMurmur tail shared by both key sizes (cityhash.cpp:263-265) */
SSH_HD hash128 finish(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    a = len16(a, c);
    b = len16(d, b);
    hash128 r;
    r.first = a ^ b;
    r.second = len16(b, a);
    return r;
}
}  // namespace city

/* CityHash128WithSeed(&x, 8, {seed, ~seed}) : len < 128 -> CityMurmur, len <= 16 branch,
   HashLen0to16 takes its 4..8-byte branch (cityhash.cpp:123-126). */
SSH_HD hash128 city128_u64(uint64_t x, uint64_t seed) {
    uint64_t a = seed, b = ~seed;
    a = city::shift_mix(a * city::K1) * city::K1;
    const uint64_t lo32 = x & 0xFFFFFFFFULL, hi32 = x >> 32;
    const uint64_t c = b * city::K1 + city::len16(8 + (lo32 << 3), hi32);
    const uint64_t d = city::shift_mix(a + x);
    return city::finish(a, b, c, d);
}

/* CityHash128WithSeed(&x, 16, {seed, ~seed}), x = {lo, hi} little-endian words:
   HashLen0to16 takes its 9..16-byte branch (cityhash.cpp:118-121). */
SSH_HD hash128 city128_u128(uint64_t lo, uint64_t hi, uint64_t seed) {
    uint64_t a = seed, b = ~seed;
    a = city::shift_mix(a * city::K1) * city::K1;
    const uint64_t t = hi + 16;
    const uint64_t rot = (t >> 16) | (t << 48);
    const uint64_t c = b * city::K1 + (city::len16(lo, rot) ^ hi);
    const uint64_t d = city::shift_mix(a + lo);
    return city::finish(a, b, c, d);
}

template <int W>
SSH_HD hash128 city128_kmer(kmer_w<W> const& x, uint64_t seed) {
    if constexpr (W == 1) return city128_u64(x.w[0], seed);
    else return city128_u128(x.w[0], x.w[1], seed);
}

/* ---- fixed-width packed vector ("compact vector") access ------------------------------ */

/* Field i of width w (1..64) from a little-endian bit stream of 64-bit words. The array
   carries one padding word so that word+1 is always addressable. */
SSH_HD uint64_t packed_get(uint64_t const* __restrict__ data, uint64_t i, uint32_t w) {
    const uint64_t bit = i * w;
    const uint64_t word = bit >> 6;
    const uint32_t sh = uint32_t(bit & 63);
    uint64_t v = data[word] >> sh;
    if (sh + w > 64) v |= data[word + 1] << (64 - sh);
    return v & low_mask(w);
}

/* ---- MPHF layout ----------------------------------------------------------------------- */

struct mphf_partition {   // 48 bytes, 16-byte aligned: three dwordx4 loads on device
    uint64_t key_offset;  // ids of this partition start here
    uint64_t pilot_base;  // index of the partition's first pilot in the packed pilot vector
    uint64_t free_base;   // index of the partition's first entry in free_slots
    uint32_t num_keys;
    uint32_t table_size;  // >= num_keys; positions are drawn in [0, table_size)
    uint32_t dense_buckets;   // buckets receiving the 60% "dense" keys
    uint32_t sparse_buckets;  // the rest
    uint64_t reserved;
};
static_assert(sizeof(mphf_partition) == 48, "mphf_partition layout");

struct mphf_view {  // plain pointers: valid for host arrays or device arrays alike
    mphf_partition const* parts;
    uint64_t const* pilots;      // packed, pilot_width bits each (+1 padding word)
    uint32_t const* free_slots;
    uint64_t seed;
    uint64_t num_keys;
    uint32_t num_parts;
    uint32_t pilot_width;
};

constexpr uint32_t MPHF_DENSE_THRESHOLD = 0x9999999Au;  // ~0.6 * 2^32
constexpr uint64_t MPHF_PILOT_MUL = 0x9E3779B97F4A7C15ULL;
constexpr uint64_t MPHF_POS_MUL = 0xD6E8FEB86659FD93ULL;

SSH_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return uint32_t((uint64_t(a) * b) >> 32); }

SSH_HD uint32_t mphf_partition_of(hash128 h, uint32_t num_parts) {
    return mulhi32(uint32_t((h.first ^ h.second) >> 32), num_parts);
}

SSH_HD uint32_t mphf_bucket_of(hash128 h, uint32_t dense, uint32_t sparse) {
    const uint32_t sel = uint32_t(h.first >> 32);
    const uint32_t v = uint32_t(h.first);
    return sel < MPHF_DENSE_THRESHOLD ? mulhi32(v, dense) : dense + mulhi32(v, sparse);
}

/* XOR displacement: the pilot perturbs h2, one xorshift-multiply round spreads the change
   to the top bits, the top 32 bits are range-reduced to the table. */
SSH_HD uint32_t mphf_position(uint64_t h2, uint64_t pilot, uint32_t table_size) {
    uint64_t x = h2 ^ (pilot * MPHF_PILOT_MUL);
    x = (x ^ (x >> 32)) * MPHF_POS_MUL;
    return mulhi32(uint32_t(x >> 32), table_size);
}

SSH_HD uint64_t mphf_eval(mphf_view const& f, hash128 h) {
    const uint32_t pi = mphf_partition_of(h, f.num_parts);
    const mphf_partition part = f.parts[pi];
    const uint32_t bucket = mphf_bucket_of(h, part.dense_buckets, part.sparse_buckets);
    const uint64_t pilot = packed_get(f.pilots, part.pilot_base + bucket, f.pilot_width);
    uint32_t p = mphf_position(h.second, pilot, part.table_size);
    if (p >= part.num_keys) p = f.free_slots[part.free_base + (p - part.num_keys)];
    return part.key_offset + p;
}

}  // namespace sshash_amd
