// kmer_core.hpp -- 2-bit k-mer primitives shared by host (builder) and device (HIP kernels).
//
// Everything here is integer/bit arithmetic. The functions restate *behaviour* of the
// reference (jermp/sshash) so that results are bit-identical, written for 64-wide
// wavefronts: no tables, no byte loops in the hot functions, W = 1 (k <= 31) or
// W = 2 (k <= 63) 64-bit words per k-mer.
//
// Reference behaviour followed (paths relative to the reference checkout):
//   * base code  (c >> 1) & 3  -> A0 C1 T2 G3, case-insensitive   include/kmer.hpp:194
//   * first base of the k-mer sits in the least-significant 2 bits include/kmer.hpp:80, include/util.hpp:207-213
//   * reverse complement                                          include/kmer.hpp:141-165
//   * m-mer hash (x * 0x517cc1b727220a95) ^ magic                 include/hash_util.hpp:91
//   * minimizer = leftmost m-mer with the smallest hash           include/util.hpp:262-283
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SSH_HD __host__ __device__ __forceinline__
#else
#define SSH_HD inline
#endif

namespace sshash_amd {

constexpr uint64_t INVALID_U64 = ~uint64_t(0);       // include/constants.hpp:5
constexpr uint32_t MIN_L = 6;                        // include/constants.hpp:13
constexpr uint32_t MAX_L = 13;                       // include/constants.hpp:14
constexpr uint32_t MAX_BUCKET_SMALL = 1u << MIN_L;   // buckets of <= 64 positions are MIDLOAD
constexpr uint64_t MMER_HASH_MUL = 0x517cc1b727220a95ULL;

template <int W>
struct kmer_w {
    uint64_t w[W];
};

SSH_HD uint32_t base_code(char c) { return (uint32_t(uint8_t(c)) >> 1) & 3u; }

/* A C G T a c g t only (include/kmer.hpp:209-219,253-255). */
SSH_HD bool base_is_valid(char c) {
    const uint32_t u = uint32_t(uint8_t(c)) & 0xDFu;  // fold case
    return u == 'A' || u == 'C' || u == 'G' || u == 'T';
}

/* reverse the order of the 32 2-bit groups of x and complement each base
   (complement of a base = code ^ 2 with the A0 C1 T2 G3 map). */
SSH_HD uint64_t revcomp_word(uint64_t x) {
    x ^= 0xAAAAAAAAAAAAAAAAULL;
#if defined(__HIP_DEVICE_COMPILE__)
    /* v_bfrev reverses all 64 bits, which also flips the two bits inside every base:
       swap them back. */
    const uint64_t r = __brevll(x);
    return ((r & 0x5555555555555555ULL) << 1) | ((r >> 1) & 0x5555555555555555ULL);
#else
    uint64_t r = __builtin_bswap64(x);
    r = ((r & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((r >> 4) & 0x0F0F0F0F0F0F0F0FULL);
    r = ((r & 0x3333333333333333ULL) << 2) | ((r >> 2) & 0x3333333333333333ULL);
    return r;
#endif
}

SSH_HD uint64_t low_mask(uint32_t bits) {  // bits in [0,64]
    return bits >= 64 ? ~uint64_t(0) : ((uint64_t(1) << bits) - 1);
}

template <int W>
SSH_HD kmer_w<W> kmer_zero() {
    kmer_w<W> x;
    for (int i = 0; i < W; ++i) x.w[i] = 0;
    return x;
}

template <int W>
SSH_HD bool kmer_eq(kmer_w<W> const& a, kmer_w<W> const& b) {
    bool e = true;
    for (int i = 0; i < W; ++i) e = e && (a.w[i] == b.w[i]);
    return e;
}

/* numeric order of the packed value (include/kmer.hpp:33), most significant word last */
template <int W>
SSH_HD bool kmer_less(kmer_w<W> const& a, kmer_w<W> const& b) {
    for (int i = W - 1; i >= 0; --i) {
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
    }
    return false;
}

/* x >> (2*chars), chars < 32*W */
template <int W>
SSH_HD kmer_w<W> kmer_shr_chars(kmer_w<W> x, uint32_t chars) {
    const uint32_t s = 2 * chars;
    if constexpr (W == 1) {
        x.w[0] = s >= 64 ? 0 : (x.w[0] >> s);
    } else {
        if (s == 0) return x;
        if (s >= 64) {
            x.w[0] = s >= 128 ? 0 : (x.w[1] >> (s - 64));
            x.w[1] = 0;
        } else {
            x.w[0] = (x.w[0] >> s) | (x.w[1] << (64 - s));
            x.w[1] >>= s;
        }
    }
    return x;
}

/* keep the low 2*chars bits */
template <int W>
SSH_HD kmer_w<W> kmer_take_chars(kmer_w<W> x, uint32_t chars) {
    const uint32_t b = 2 * chars;
    if constexpr (W == 1) {
        x.w[0] &= low_mask(b);
    } else {
        if (b <= 64) {
            x.w[0] &= low_mask(b);
            x.w[1] = 0;
        } else {
            x.w[1] &= low_mask(b - 64);
        }
    }
    return x;
}

template <int W>
SSH_HD kmer_w<W> kmer_revcomp(kmer_w<W> x, uint32_t k) {
    kmer_w<W> r;
    if constexpr (W == 1) {
        r.w[0] = revcomp_word(x.w[0]) >> (64 - 2 * k);
    } else {
        /* word order swaps too (include/kmer.hpp:162), then drop the unused top */
        const uint64_t hi = revcomp_word(x.w[0]);
        const uint64_t lo = revcomp_word(x.w[1]);
        const uint32_t s = 128 - 2 * k;  // k >= 1 so s <= 126; k <= 63 so s >= 2
        if (s >= 64) {
            r.w[0] = hi >> (s - 64);
            r.w[1] = 0;
        } else {
            r.w[0] = (lo >> s) | (hi << (64 - s));
            r.w[1] = hi >> s;
        }
    }
    return r;
}

/* reverse complement of an m-mer (m <= 31) held in one word */
SSH_HD uint64_t mmer_revcomp(uint64_t x, uint32_t m) { return revcomp_word(x) >> (64 - 2 * m); }

/* ASCII -> packed, reading exactly k chars, no validation (src/dictionary.cpp:58-63). */
template <int W>
SSH_HD kmer_w<W> kmer_from_ascii(char const* s, uint32_t k) {
    kmer_w<W> x = kmer_zero<W>();
    for (uint32_t i = 0; i < k; ++i) {
        const uint64_t c = base_code(s[i]);
        if constexpr (W == 1) {
            x.w[0] |= c << (2 * i);
        } else {
            if (i < 32) x.w[0] |= c << (2 * i);
            else x.w[1] |= c << (2 * (i - 32));
        }
    }
    return x;
}

/* Slide a k-mer window one base to the right: drop the first base, append `code` as the
   last (streaming_query.hpp:68-70); and the matching update of its reverse complement:
   prepend the complement, drop the last (streaming_query.hpp:72-75). */
template <int W>
SSH_HD kmer_w<W> kmer_roll(kmer_w<W> x, uint64_t code, uint32_t k) {
    x = kmer_shr_chars<W>(x, 1);
    const uint32_t b = 2 * (k - 1);
    if constexpr (W == 1) {
        x.w[0] |= code << b;
    } else {
        if (b < 64) x.w[0] |= code << b;
        else x.w[1] |= code << (b - 64);
    }
    return kmer_take_chars<W>(x, k);
}

template <int W>
SSH_HD kmer_w<W> kmer_roll_rc(kmer_w<W> x, uint64_t code, uint32_t k) {
    if constexpr (W == 1) {
        x.w[0] = (x.w[0] << 2) | (code ^ 2);
    } else {
        x.w[1] = (x.w[1] << 2) | (x.w[0] >> 62);
        x.w[0] = (x.w[0] << 2) | (code ^ 2);
    }
    return kmer_take_chars<W>(x, k);
}

/* Neighbours of a k-mer in the de Bruijn graph (src/dictionary.cpp:111-126): which = 0..3 the forward
   neighbour suffix(x) + "ACTG"[which], which = 4..7 the backward neighbour "ACTG"[which - 4] + prefix(x):
   neighbourhood::forward/backward are indexed by the character's code (alphabet "ACTG", include/kmer.hpp:118). */
template <int W>
SSH_HD kmer_w<W> kmer_neighbour(kmer_w<W> const& x, uint32_t which, uint32_t k) {
    const uint64_t code = which & 3;
    return which < 4 ? kmer_roll<W>(x, code, k) : kmer_roll_rc<W>(x, code ^ 2, k);
}

struct minimizer_t {
    uint64_t value;  // the m-mer itself (not its hash)
    uint32_t pos;    // position of its first base inside the k-mer
};

SSH_HD uint64_t mmer_hash(uint64_t mmer, uint64_t magic) { return (mmer * MMER_HASH_MUL) ^ magic; }

/* Stateless minimizer: leftmost position among equal hashes (strict <). */
template <int W>
SSH_HD minimizer_t compute_minimizer(kmer_w<W> x, uint32_t k, uint32_t m, uint64_t magic) {
    const uint64_t mask = low_mask(2 * m);
    uint64_t best_hash = INVALID_U64;
    minimizer_t r;
    r.value = INVALID_U64;
    r.pos = 0;
    const uint32_t n = k - m + 1;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t mmer = x.w[0] & mask;
        const uint64_t h = mmer_hash(mmer, magic);
        if (h < best_hash) {
            best_hash = h;
            r.value = mmer;
            r.pos = i;
        }
        x = kmer_shr_chars<W>(x, 1);
    }
    return r;
}

/* Owner shard of a minimizer when the sparse-and-skew index is partitioned by minimizer over several
   GPUs. Independent of the MPHF / directory hashes. */
SSH_HD uint32_t shard_of_minimizer(uint64_t minimizer, uint32_t num_shards) {
    const uint64_t h = minimizer * 0xA24BAED4963EE407ULL;
    return uint32_t((uint64_t(uint32_t(h >> 32) ^ uint32_t(h >> 11)) * num_shards) >> 32);
}

/* XXH64 of one little-endian 64-bit word (published xxHash algorithm). The reference
   derives the m-mer hash magic as xxhash_64(seed, 0) (include/hash_util.hpp:88). */
inline uint64_t xxh64_of_u64(uint64_t value, uint64_t seed) {
    constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL,
                       P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL,
                       P5 = 0x27D4EB2F165667C5ULL;
    auto rotl = [](uint64_t v, int r) { return (v << r) | (v >> (64 - r)); };
    uint64_t h = seed + P5 + 8;
    uint64_t k1 = rotl(value * P2, 31) * P1;
    h ^= k1;
    h = rotl(h, 27) * P1 + P4;
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return h;
}

}  // namespace sshash_amd
