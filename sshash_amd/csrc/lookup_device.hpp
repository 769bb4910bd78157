// lookup_device.hpp -- device-side point lookup (one query per lane), gfx950.
//
// Restates, for the device layout of device_layout.hpp, the reference call tree
//   dictionary::lookup                src/dictionary.cpp:64-78 (regular: forward probe, then
//                                     reverse-complement probe), :24-56 (canonical)
//   sparse_and_skew_index::lookup     include/sparse_and_skew_index.hpp:112-137, skew :34-44
//   spss::lookup_regular/_canonical   include/spectrum_preserving_string_set.hpp:29-112,213-275
// Integer/bit work only; every load is an 8- or 16-byte aligned global load.
#pragma once

#include <hip/hip_runtime.h>

#include "device_layout.hpp"

namespace sshash_amd {

struct skew_part_dev {  // lives in device memory: indexed per lane by the codeword's partition id
    mphf_view f;
    uint64_t const* positions;
    uint32_t pos_width;
    uint32_t pad;
};

struct hit_t {
    uint64_t kmer_offset;
    uint32_t string_id;
    int8_t orientation;
    bool found;
    bool minimizer_found;
};

template <int W>
struct window_t {
    kmer_w<W> kmer;      // k bases starting at the requested offset
    uint32_t string_id;  // string containing the first base
    bool crosses;        // a string boundary lies in (off, off + k - 1]
};

/* 16 bytes that will not be read again soon (a bucket line, an atom, a tile of queries): nontemporal */
typedef uint32_t sk_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 sk_load_piece(char const* p) {
    const sk_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const sk_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint64_t funnel_shr(uint64_t lo, uint64_t hi, uint32_t s /* 0..63 */) {
    return (lo >> s) | ((hi << 1) << (63 - s));
}

/* k <= 31: one aligned 32-byte atom holds the whole window (device_layout.hpp (1)) */
template <int W>
__device__ __forceinline__ window_t<W> read_window(void const* __restrict__ blocks, uint64_t off, uint32_t k);

template <>
__device__ __forceinline__ window_t<1> read_window<1>(void const* __restrict__ blocks, uint64_t off, uint32_t k) {
    const uint4* A = reinterpret_cast<const uint4*>(blocks) + 2 * (off >> 5);
    const uint32_t r = uint32_t(off) & 31u;
    const uint4 q0 = A[0];  // bases[0], bases[1]
    const uint4 q1 = A[1];  // marks, rank, spare
    const uint64_t b0 = uint64_t(q0.x) | (uint64_t(q0.y) << 32);
    const uint64_t b1 = uint64_t(q0.z) | (uint64_t(q0.w) << 32);
    window_t<1> w;
    w.kmer.w[0] = funnel_shr(b0, b1, 2 * r) & low_mask(2 * k);
    const uint64_t marks = uint64_t(q1.x) | (uint64_t(q1.y) << 32);
    w.crosses = ((marks >> (r + 1)) & low_mask(k - 1)) != 0;
    w.string_id = q1.z + __popc(q1.x & uint32_t((uint64_t(2) << r) - 1)) - 1;
    return w;
}

/* k <= 63: three consecutive 16-byte granules */
template <>
__device__ __forceinline__ window_t<2> read_window<2>(void const* __restrict__ blocks, uint64_t off, uint32_t k) {
    const uint4* G = reinterpret_cast<const uint4*>(blocks) + (off >> 5);
    const uint32_t r = uint32_t(off) & 31u;
    const uint32_t s = 2 * r;
    const uint4 g0 = G[0];
    const uint4 g1 = G[1];
    const uint4 g2 = G[2];
    const uint64_t b0 = uint64_t(g0.z) | (uint64_t(g0.w) << 32);
    const uint64_t b1 = uint64_t(g1.z) | (uint64_t(g1.w) << 32);
    const uint64_t b2 = uint64_t(g2.z) | (uint64_t(g2.w) << 32);
    window_t<2> w;
    w.kmer.w[0] = funnel_shr(b0, b1, s);
    w.kmer.w[1] = funnel_shr(b1, b2, s);
    w.kmer = kmer_take_chars<2>(w.kmer, k);
    const uint64_t m_lo = uint64_t(g0.y) | (uint64_t(g1.y) << 32);
    const uint64_t m_hi = uint64_t(g2.y);
    const uint64_t following = (m_lo >> (r + 1)) | (m_hi << (63 - r));  // mark bits of off+1, off+2, ...
    w.crosses = (following & low_mask(k - 1)) != 0;
    w.string_id = g0.x + __popc(g0.y & uint32_t((uint64_t(2) << r) - 1)) - 1;
    return w;
}

/* the m-mer (m <= 31) starting at base `off`, whatever the block layout */
__device__ __forceinline__ uint64_t read_mmer(dict_view const& d, uint64_t off) {
    return d.k <= 31 ? read_window<1>(d.granules, off, d.m).kmer.w[0] : read_window<2>(d.granules, off, d.m).kmer.w[0];
}

template <int W>
__device__ __forceinline__ uint64_t mmer_at(kmer_w<W> const& x, uint32_t pos, uint32_t m) {
    return kmer_shr_chars<W>(x, pos).w[0] & low_mask(2 * m);
}

struct bucket_t {
    uint64_t first_offset;  // offset of the first (or only) minimizer position
    uint64_t begin;         // index of the bucket in mid_load (MIDLOAD only)
    uint32_t size;
    bool heavy;
    bool valid;      // false: skew index pointed outside heavy_load (absent k-mer)
    bool other_key;  // the map proves that this minimizer has no bucket here
    bool retry;      // directory sector overflowed: the MPHF path must have the last word
};

/* control codeword -> bucket (include/sparse_and_skew_index.hpp:112-137, skew :34-44) */
template <int W>
__device__ __forceinline__ void decode_codeword(dict_view const& d, skew_part_dev const* __restrict__ skew, uint64_t code,
                                                kmer_w<W> const& skew_key, bucket_t& b) {
    if ((code & 1) == 0) {  // SINGLETON
        b.first_offset = code >> 1;
    } else if ((code & 3) == 1) {  // MIDLOAD
        b.size = uint32_t((code >> 2) & (MAX_BUCKET_SMALL - 1)) + 2;
        b.begin = uint64_t(d.begin_buckets_of_size[b.size]) + (code >> (2 + MIN_L)) * b.size;
        b.first_offset = packed_get(d.mid_load, b.begin, d.off_width);
    } else {  // HEAVYLOAD: second MPHF keyed by the k-mer
        b.heavy = true;
        const skew_part_dev sp = skew[(code >> 2) & 7];
        const uint64_t kid = mphf_eval(sp.f, city128_kmer<W>(skew_key, sp.f.seed));
        const uint64_t at = (code >> 5) + packed_get(sp.positions, kid, sp.pos_width);
        /* for a k-mer that is not a key the position is arbitrary and may fall outside the
           array (spectrum_preserving_string_set.hpp:51-64): treat as a miss */
        b.valid = at < d.heavy_size;
        b.first_offset = b.valid ? packed_get(d.heavy_load, at, d.off_width) : 0;
    }
}

__device__ __forceinline__ bucket_t empty_bucket() {
    bucket_t b;
    b.first_offset = 0;
    b.begin = 0;
    b.size = 1;
    b.heavy = false;
    b.valid = true;
    b.other_key = false;
    b.retry = false;
    return b;
}

/* minimizer -> MPHF -> control codeword (+ fingerprint) -> bucket:
   minimizers_control_map::lookup, include/minimizers_control_map.hpp:36-39 */
template <int W>
__device__ __forceinline__ bucket_t resolve_bucket_mphf(dict_view const& d, skew_part_dev const* __restrict__ skew,
                                                        uint64_t minimizer, kmer_w<W> const& skew_key) {
    bucket_t b = empty_bucket();
    const uint64_t id = mphf_eval(d.minimizers, city128_u64(minimizer, d.minimizers.seed));
    if (d.cw_packed) {  // (uniform) no fingerprint: the m-mer comparison at the bucket's first offset decides, as in the reference
        decode_codeword<W>(d, skew, packed_get(d.codewords, id, d.cw_width), skew_key, b);
        return b;
    }
    const uint64_t entry = d.codewords[id];
    const uint64_t code = entry & low_mask(d.cw_width);
    if ((entry >> d.cw_width) != minimizer_fingerprint(minimizer, d.m, d.canonical != 0, d.cw_width)) {
        /* same outcome as the m-mer comparison at the bucket's first offset failing
           (spectrum_preserving_string_set.hpp:46-65): a miss whose minimizer_found is true only
           for HEAVYLOAD buckets */
        b.other_key = true;
        b.heavy = (code & 3) == 3;
        return b;
    }
    decode_codeword<W>(d, skew, code, skew_key, b);
    return b;
}

struct dir_answer {
    uint64_t code;   // control codeword of the matching entry
    bool present;    // a fingerprint matched
    bool overflow;   // bucket flagged: a negative answer is not final
};

/* one 32-byte atom: four entries, fingerprints compared in 32-bit arithmetic */
__device__ __forceinline__ dir_answer directory_probe(dict_view const& d, uint64_t minimizer) {
    const uint64_t h = directory_hash(minimizer);
    const uint4* B = reinterpret_cast<const uint4*>(d.directory.buckets + 4 * uint64_t(directory_bucket(h, d.directory.num_buckets)));
    const uint4 q0 = B[0], q1 = B[1];
    const uint32_t want = (directory_fingerprint(h) << 8) | (1u << 24);  // fingerprint + valid bit, as laid out in the high dword
    const uint32_t mask = 0x01FFFF00u;
    const bool m0 = (q0.y & mask) == want, m1 = (q0.w & mask) == want, m2 = (q1.y & mask) == want, m3 = (q1.w & mask) == want;
    uint32_t lo = 0, hi = 0;  // fingerprints are unique inside a bucket: at most one match
    lo = m0 ? q0.x : lo;  hi = m0 ? q0.y : hi;
    lo = m1 ? q0.z : lo;  hi = m1 ? q0.w : hi;
    lo = m2 ? q1.x : lo;  hi = m2 ? q1.y : hi;
    lo = m3 ? q1.z : lo;  hi = m3 ? q1.w : hi;
    dir_answer a;
    a.present = m0 || m1 || m2 || m3;
    a.code = uint64_t(lo) | (uint64_t(hi & 0xFFu) << 32);
    uint32_t flag = q0.y >> 31;
    asm volatile("" : "+v"(flag));  // materialise now: the probe's registers are recycled by the bucket scan
    a.overflow = flag != 0;
    return a;
}

/* minimizer -> directory bucket (one 32-byte fetch) -> control codeword -> bucket */
template <int W>
__device__ __forceinline__ bucket_t resolve_bucket_directory(dict_view const& d, skew_part_dev const* __restrict__ skew,
                                                             uint64_t minimizer, kmer_w<W> const& skew_key) {
    bucket_t b = empty_bucket();
    const dir_answer a = directory_probe(d, minimizer);
    b.retry = a.overflow;
    if (!a.present) {
        b.other_key = true;  // (unless retry) the minimizer is not in the dictionary at all
        return b;
    }
    decode_codeword<W>(d, skew, a.code, skew_key, b);
    return b;
}

__device__ __forceinline__ hit_t miss(bool minimizer_found) {
    hit_t h;
    h.kmer_offset = INVALID_U64;
    h.string_id = 0;
    h.orientation = 1;
    h.found = false;
    h.minimizer_found = minimizer_found;
    return h;
}

/* spss::lookup_regular (include/spectrum_preserving_string_set.hpp:29-73,213-235) */
template <int W>
__device__ __forceinline__ hit_t scan_regular(dict_view const& d, bucket_t const& b, kmer_w<W> const& x, minimizer_t mini) {
    if (b.other_key) return miss(b.heavy);
    if (!b.valid) return miss(true);
    hit_t h = miss(true);
    uint64_t p = b.first_offset;
    if (p >= mini.pos) {
        /* one read serves the minimizer check (:46-65) and the first candidate (:68-70) */
        const window_t<W> w = read_window<W>(d.granules, p - mini.pos, d.k);
        if (mmer_at<W>(w.kmer, mini.pos, d.m) != mini.value) return miss(b.heavy);
        if (kmer_eq<W>(w.kmer, x) && !w.crosses) {
            h.found = true;
            h.kmer_offset = p - mini.pos;
            h.string_id = w.string_id;
            return h;
        }
    } else {
        const window_t<W> w = read_window<W>(d.granules, p, d.k);
        if ((w.kmer.w[0] & low_mask(2 * d.m)) != mini.value) return miss(b.heavy);
    }
    for (uint32_t i = 1; i < b.size; ++i) {
        p = packed_get(d.mid_load, b.begin + i, d.off_width);
        if (p < mini.pos) continue;
        const window_t<W> w = read_window<W>(d.granules, p - mini.pos, d.k);
        if (kmer_eq<W>(w.kmer, x) && !w.crosses) {
            h.found = true;
            h.kmer_offset = p - mini.pos;
            h.string_id = w.string_id;
            return h;
        }
    }
    return h;
}

template <int W, bool DIRECTORY>
__device__ __forceinline__ hit_t probe_regular(dict_view const& d, skew_part_dev const* __restrict__ skew,
                                               kmer_w<W> const& x, minimizer_t mini) {
    if constexpr (DIRECTORY) {
        if (d.directory.enabled) {
            const bucket_t b = resolve_bucket_directory<W>(d, skew, mini.value, x);
            hit_t h = scan_regular<W>(d, b, x, mini);
            if (h.found || !b.retry) {
                if (!h.found && b.other_key) h.minimizer_found = false;
                return h;
            }
        }
    }
    return scan_regular<W>(d, resolve_bucket_mphf<W>(d, skew, mini.value, x), x, mini);
}

/* spss::lookup_canonical (include/spectrum_preserving_string_set.hpp:75-112,237-275) */
template <int W>
__device__ __forceinline__ hit_t scan_canonical(dict_view const& d, bucket_t const& b, kmer_w<W> const& x,
                                                kmer_w<W> const& x_rc, minimizer_t mini) {
    if (b.other_key) return miss(b.heavy);
    if (!b.valid) return miss(true);
    hit_t h = miss(true);
    uint64_t p = b.first_offset;
    {
        const window_t<W> w = read_window<W>(d.granules, p, d.k);
        const uint64_t mm = w.kmer.w[0] & low_mask(2 * d.m);
        if (mm != mini.value && mm != mmer_revcomp(mini.value, d.m)) return miss(b.heavy);
    }
    for (uint32_t i = 0; i < b.size; ++i) {
        if (i) p = packed_get(d.mid_load, b.begin + i, d.off_width);
        uint32_t pos = mini.pos;
        for (int attempt = 0; attempt < 2; ++attempt, pos = d.k - d.m - mini.pos) {
            if (p < pos) continue;
            const window_t<W> w = read_window<W>(d.granules, p - pos, d.k);
            const bool fwd = kmer_eq<W>(w.kmer, x), bwd = kmer_eq<W>(w.kmer, x_rc);
            if ((fwd || bwd) && !w.crosses) {
                h.found = true;
                h.kmer_offset = p - pos;
                h.string_id = w.string_id;
                h.orientation = bwd ? -1 : 1;
                return h;
            }
        }
    }
    return h;
}

template <int W, bool DIRECTORY>
__device__ __forceinline__ hit_t probe_canonical(dict_view const& d, skew_part_dev const* __restrict__ skew,
                                                 kmer_w<W> const& x, kmer_w<W> const& x_rc, minimizer_t mini) {
    const kmer_w<W> key = kmer_less<W>(x_rc, x) ? x_rc : x;  // src/dictionary.cpp:53
    if constexpr (DIRECTORY) {
        if (d.directory.enabled) {
            const bucket_t b = resolve_bucket_directory<W>(d, skew, mini.value, key);
            hit_t h = scan_canonical<W>(d, b, x, x_rc, mini);
            if (h.found || !b.retry) {
                if (!h.found && b.other_key) h.minimizer_found = false;
                return h;
            }
        }
    }
    return scan_canonical<W>(d, resolve_bucket_mphf<W>(d, skew, mini.value, key), x, x_rc, mini);
}

/* dictionary::lookup(Kmer, bool) -- src/dictionary.cpp:64-78 and :24-42.
   DIRECTORY: resolve minimizers through the one-sector directory (device_layout.hpp (4)). */
template <int W, bool CANON, bool DIRECTORY>
__device__ __forceinline__ hit_t lookup_one(dict_view const& d, skew_part_dev const* __restrict__ skew,
                                            kmer_w<W> const& x, bool check_rc) {
    if constexpr (CANON) {
        const kmer_w<W> x_rc = kmer_revcomp<W>(x, d.k);
        const minimizer_t mf = compute_minimizer<W>(x, d.k, d.m, d.hash_magic);
        const minimizer_t mr = compute_minimizer<W>(x_rc, d.k, d.m, d.hash_magic);
        if (mf.value < mr.value) return probe_canonical<W, DIRECTORY>(d, skew, x, x_rc, mf);
        if (mr.value < mf.value) return probe_canonical<W, DIRECTORY>(d, skew, x, x_rc, mr);
        hit_t h = probe_canonical<W, DIRECTORY>(d, skew, x, x_rc, mf);
        if (!h.found) h = probe_canonical<W, DIRECTORY>(d, skew, x, x_rc, mr);
        return h;
    } else {
        hit_t h = probe_regular<W, DIRECTORY>(d, skew, x, compute_minimizer<W>(x, d.k, d.m, d.hash_magic));
        if (!h.found && check_rc) {
            const kmer_w<W> x_rc = kmer_revcomp<W>(x, d.k);
            h = probe_regular<W, DIRECTORY>(d, skew, x_rc, compute_minimizer<W>(x_rc, d.k, d.m, d.hash_magic));
            h.orientation = -1;
        }
        return h;
    }
}

/* `minimizer_found` of a k-mer already known to be absent (the table said so): what the reference's lookup returns for it is
   the result of its LAST probe -- a regular dictionary probes the k-mer, then, not having found it, its reverse complement
   and returns that result (src/dictionary.cpp:70-75); a canonical one probes once (or twice on a minimizer tie) -- so the
   forward probe of a regular dictionary need not be repeated here. MPHF path, no directory (device_layout.hpp (4)). */
template <int W, bool CANON>
__device__ __forceinline__ bool minimizer_found_of_a_miss(dict_view const& d, skew_part_dev const* __restrict__ skew, kmer_w<W> const& x,
                                                          bool check_rc) {
    if constexpr (CANON) {
        return lookup_one<W, true, false>(d, skew, x, check_rc).minimizer_found;
    } else {
        const kmer_w<W> y = check_rc ? kmer_revcomp<W>(x, d.k) : x;
        return probe_regular<W, false>(d, skew, y, compute_minimizer<W>(y, d.k, d.m, d.hash_magic)).minimizer_found;
    }
}

/* ---- lookup without the table (minimizer shards, SSHASH_AMD_SKTABLE=0): the common case in a lean kernel,
   everything else deferred ------------
   The generic `lookup_one` above carries the code of every rare case (MIDLOAD scans, the skew index,
   the MPHF fallback of overflowed directory sectors); a wave executes all of it as soon as ONE of its
   64 lanes needs it. `fast_probe_*` resolves: minimizer absent (the directory says so), SINGLETON and
   MIDLOAD buckets -- ~98 % of the probes of a random batch -- and reports DEFER for the rest (HEAVYLOAD
   buckets, probes left open by an overflowed directory bucket, canonical minimizer ties); deferred
   queries are compacted into a queue and re-run through `lookup_one` by a second, small launch. */

enum fast_outcome : int {
    FAST_MISS = 0,
    FAST_HIT = 1,
    FAST_DEFER = 2,
    FAST_CONTINUE = 3,  // table probe to be resumed; kmer_offset = queue-entry flags
    FAST_SCAN = 4       // MIDLOAD bucket whose first position did not hold the k-mer: the rest is scanned by the
                        // wave-cooperative pass; kmer_offset = scan_meta()
};

struct fast_t {
    uint64_t kmer_offset;
    uint32_t string_id;
    int outcome;
    int8_t orientation;
};

/* what the first pass hands to the bucket-scan pass: where the bucket's offsets lie in mid_load (the builder keeps that
   array below 2^32 entries), how many are left to try (size - 1 <= 63), where the minimizer starts in the probing k-mer,
   and which strand probes (regular dictionaries: 1 = the reverse complement of the query) */
__device__ __forceinline__ uint64_t scan_meta(bucket_t const& b, uint32_t pos, bool rc_strand) {
    return (b.begin & 0xFFFFFFFFull) | (uint64_t(b.size - 1) << 32) | (uint64_t(pos) << 38) | (uint64_t(rc_strand ? 1 : 0) << 44);
}

/* control codeword -> SINGLETON / MIDLOAD bucket; false when the probe has to be deferred (HEAVYLOAD) */
__device__ __forceinline__ bool fast_bucket(dict_view const& d, uint64_t code, bucket_t& b) {
    b = empty_bucket();
    if ((code & 1) == 0) {
        b.first_offset = code >> 1;
        return true;
    }
    if ((code & 3) == 3) return false;
    b.size = uint32_t((code >> 2) & (MAX_BUCKET_SMALL - 1)) + 2;
    b.begin = uint64_t(d.begin_buckets_of_size[b.size]) + (code >> (2 + MIN_L)) * b.size;
    b.first_offset = packed_get(d.mid_load, b.begin, d.off_width);
    return true;
}

__device__ __forceinline__ fast_t fast_unsettled(bool defer) {
    fast_t r;
    r.kmer_offset = 0;
    r.string_id = 0;
    r.orientation = 1;
    r.outcome = defer ? FAST_DEFER : FAST_MISS;
    return r;
}

/* minimizer -> control codeword through whatever the replica holds: the one-atom directory, or MPHF pilot + codeword (two
   dependent atoms; the codeword's fingerprint ends a probe with a foreign minimizer there). `settled`: the answer is
   final (false: an overflowed directory bucket -- the complete path must have the last word). */
struct resolve_t {
    uint64_t code;
    bool present;
    bool settled;
};

__device__ __forceinline__ resolve_t fast_resolve(dict_view const& d, uint64_t minimizer) {
    resolve_t r;
    if (d.directory.enabled) {  // uniform
        const dir_answer a = directory_probe(d, minimizer);
        r.code = a.code;
        r.present = a.present;
        r.settled = !a.overflow;
    } else {
        const uint64_t id = mphf_eval(d.minimizers, city128_u64(minimizer, d.minimizers.seed));
        /* the codeword is not read again before a few hundred million others have passed -- nontemporal, so that it does not push the
           pilots (137 MB on C3, read by every probe) out of the caches */
        const uint64_t entry = d.cw_packed ? packed_get(d.codewords, id, d.cw_width) : __builtin_nontemporal_load(d.codewords + id);
        r.code = entry & low_mask(d.cw_width);
        r.present = d.cw_packed || (entry >> d.cw_width) == minimizer_fingerprint(minimizer, d.m, d.canonical != 0, d.cw_width);
        r.settled = true;
    }
    return r;
}

/* the FIRST position of a bucket only: one read serves the minimizer check (spectrum_preserving_string_set.hpp:46-65) and
   the first candidate (:68-70). A MIDLOAD bucket whose first position does not hold the k-mer is left to the scan pass. */
template <int W>
__device__ __forceinline__ fast_t fast_probe_regular(dict_view const& d, kmer_w<W> const& x, minimizer_t mini, bool rc_strand,
                                                     resolve_t const& a) {
    if (!a.present) return fast_unsettled(!a.settled);
    bucket_t b;
    if (!fast_bucket(d, a.code, b)) return fast_unsettled(true);
    fast_t r = fast_unsettled(false);
    const uint64_t p = b.first_offset;
    const bool aligned = p >= mini.pos;
    const window_t<W> w = read_window<W>(d.granules, aligned ? p - mini.pos : p, d.k);
    const uint64_t mm = aligned ? mmer_at<W>(w.kmer, mini.pos, d.m) : (w.kmer.w[0] & low_mask(2 * d.m));
    if (mm != mini.value) return fast_unsettled(!a.settled);  // a fingerprint's false positive: final only if the directory bucket never overflowed
    if (aligned && kmer_eq<W>(w.kmer, x) && !w.crosses) {
        r.outcome = FAST_HIT;
        r.kmer_offset = p - mini.pos;
        r.string_id = w.string_id;
        return r;
    }
    if (b.size > 1) {
        r.outcome = FAST_SCAN;
        r.kmer_offset = scan_meta(b, mini.pos, rc_strand);
    }
    return r;
}

template <int W>
__device__ __forceinline__ fast_t fast_probe_canonical(dict_view const& d, kmer_w<W> const& x, kmer_w<W> const& x_rc,
                                                       minimizer_t mini) {
    const resolve_t a = fast_resolve(d, mini.value);
    if (!a.present) return fast_unsettled(!a.settled);
    bucket_t b;
    if (!fast_bucket(d, a.code, b)) return fast_unsettled(true);
    fast_t r = fast_unsettled(false);
    const uint64_t p = b.first_offset;
    /* both alignments of the first position (spectrum_preserving_string_set.hpp:237-247), issued together */
    const uint32_t pos2 = d.k - d.m - mini.pos;
    const window_t<W> w0 = read_window<W>(d.granules, p, d.k);
    const window_t<W> w1 = read_window<W>(d.granules, p >= mini.pos ? p - mini.pos : p, d.k);
    const window_t<W> w2 = read_window<W>(d.granules, p >= pos2 ? p - pos2 : p, d.k);
    const uint64_t mm = w0.kmer.w[0] & low_mask(2 * d.m);
    if (mm != mini.value && mm != mmer_revcomp(mini.value, d.m)) return fast_unsettled(!a.settled);
    auto attempt = [&](window_t<W> const& w, uint32_t pos) {
        if (p < pos || r.outcome == FAST_HIT) return;
        const bool fwd = kmer_eq<W>(w.kmer, x), bwd = kmer_eq<W>(w.kmer, x_rc);
        if ((fwd || bwd) && !w.crosses) {
            r.outcome = FAST_HIT;
            r.kmer_offset = p - pos;
            r.string_id = w.string_id;
            r.orientation = bwd ? -1 : 1;
        }
    };
    attempt(w1, mini.pos);
    attempt(w2, pos2);
    if (r.outcome != FAST_HIT && b.size > 1) {
        r.outcome = FAST_SCAN;
        r.kmer_offset = scan_meta(b, mini.pos, false);
    }
    return r;
}

/* ---- the same first pass with PAIR-COOPERATIVE 32-byte fetches (round 4; regular dictionaries, k <= 31) ---------------------
   A lane that reads a 32-byte unit -- a directory bucket, an atom of the strings -- with two 16-byte loads makes every load
   instruction of the wave touch 64 different pages: two translation requests per unit, and the chip serves 75 G of those a second
   (HISTORY.md): the table-less first pass ran at 39 G units/s, that bound, not the memory's. Here the two lanes of a pair
   read each other's units together: one load instruction fetches the even lane's unit (16 bytes a lane), the next the odd lane's,
   the halves change hands through DPP -- the same two load instructions per lane, but each touches 32 units instead of 64: one
   translation per unit. Called by all lanes of the wave (need = false: no unit wanted; the pair's load goes to unit 0). */
template <int SEL>  // 0: the pair's even lane's value; 1: the odd lane's; 2: the other lane's
__device__ __forceinline__ uint32_t pair_perm(uint32_t v) {
    constexpr int CTRL = SEL == 0 ? 0xA0 : SEL == 1 ? 0xF5 : 0xB1;  // quad_perm [0,0,2,2], [1,1,3,3], [1,0,3,2]
    return uint32_t(__builtin_amdgcn_mov_dpp(int(v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ void pair_load32(void const* __restrict__ base, uint32_t unit, bool need, uint4& lo, uint4& hi) {
    const uint32_t idx = need ? unit : 0u;
    const uint32_t ie = pair_perm<0>(idx), io = pair_perm<1>(idx);
    const uint32_t sub = threadIdx.x & 1u;
    char const* b = static_cast<char const*>(base) + 16 * sub;
    const uint4 pa = sk_load_piece(b + uint64_t(ie) * 32), pb = sk_load_piece(b + uint64_t(io) * 32);  // nontemporal: read once
    /* the even lane holds the first half of both units, the odd lane the second half of both: each gives the other what it lacks */
    const uint4 give = sub ? pa : pb;
    const uint4 got = make_uint4(pair_perm<2>(give.x), pair_perm<2>(give.y), pair_perm<2>(give.z), pair_perm<2>(give.w));
    lo = sub ? got : pa;
    hi = sub ? pb : got;
}

__device__ __forceinline__ resolve_t fast_resolve_pairs(dict_view const& d, uint64_t minimizer, bool need) {
    resolve_t r;
    r.code = 0;
    r.present = false;
    r.settled = true;
    if (d.directory.enabled) {  // uniform
        const uint64_t h = directory_hash(minimizer);
        uint4 q0, q1;
        pair_load32(d.directory.buckets, directory_bucket(h, d.directory.num_buckets), need, q0, q1);
        const uint32_t want = (directory_fingerprint(h) << 8) | (1u << 24);  // (as directory_probe)
        const uint32_t mask = 0x01FFFF00u;
        const bool m0 = (q0.y & mask) == want, m1 = (q0.w & mask) == want, m2 = (q1.y & mask) == want, m3 = (q1.w & mask) == want;
        uint32_t lo = 0, hi = 0;
        lo = m0 ? q0.x : lo;  hi = m0 ? q0.y : hi;
        lo = m1 ? q0.z : lo;  hi = m1 ? q0.w : hi;
        lo = m2 ? q1.x : lo;  hi = m2 ? q1.y : hi;
        lo = m3 ? q1.z : lo;  hi = m3 ? q1.w : hi;
        r.present = need && (m0 || m1 || m2 || m3);
        r.code = uint64_t(lo) | (uint64_t(hi & 0xFFu) << 32);
        r.settled = (q0.y >> 31) == 0;
    } else if (need) {  // pilot and codeword are 8-byte reads: one translation each as they are
        r = fast_resolve(d, minimizer);
    }
    return r;
}

__device__ __forceinline__ fast_t fast_probe_regular_pairs(dict_view const& d, kmer_w<1> const& x, minimizer_t mini, bool rc_strand,
                                                           resolve_t const& a, bool need) {
    fast_t r = fast_unsettled(need && !a.present && !a.settled);
    bucket_t b;
    b.first_offset = 0;
    b.size = 1;
    bool probe = need && a.present;
    if (probe && !fast_bucket(d, a.code, b)) {  // HEAVYLOAD: the complete path
        r = fast_unsettled(true);
        probe = false;
    }
    const uint64_t p = b.first_offset;
    const bool aligned = p >= mini.pos;
    const uint64_t off = aligned ? p - mini.pos : p;
    uint4 q0, q1;
    pair_load32(d.granules, uint32_t(off >> 5), probe, q0, q1);
    if (probe) {
        const uint32_t rr = uint32_t(off) & 31u;  // (read_window<1>)
        const uint64_t b0 = uint64_t(q0.x) | (uint64_t(q0.y) << 32), b1 = uint64_t(q0.z) | (uint64_t(q0.w) << 32);
        const uint64_t kmer = funnel_shr(b0, b1, 2 * rr) & low_mask(2 * d.k);
        const uint64_t marks = uint64_t(q1.x) | (uint64_t(q1.y) << 32);
        const bool crosses = ((marks >> (rr + 1)) & low_mask(d.k - 1)) != 0;
        const uint64_t mm = aligned ? ((kmer >> (2 * mini.pos)) & low_mask(2 * d.m)) : (kmer & low_mask(2 * d.m));
        if (mm != mini.value) {
            r = fast_unsettled(!a.settled);  // a fingerprint's false positive: final only if the directory bucket never overflowed
        } else if (aligned && kmer == x.w[0] && !crosses) {
            r.outcome = FAST_HIT;
            r.kmer_offset = p - mini.pos;
            r.string_id = q1.z + __popc(q1.x & uint32_t((uint64_t(2) << rr) - 1)) - 1;
        } else if (b.size > 1) {
            r.outcome = FAST_SCAN;
            r.kmer_offset = scan_meta(b, mini.pos, rc_strand);
        }
    }
    return r;
}

/* all lanes of the wave; active = false: no query */
__device__ __forceinline__ fast_t fast_lookup_pairs(dict_view const& d, kmer_w<1> const& x, bool active, bool check_rc) {
    const minimizer_t mf = compute_minimizer<1>(x, d.k, d.m, d.hash_magic);
    fast_t r = fast_probe_regular_pairs(d, x, mf, false, fast_resolve_pairs(d, mf.value, active), active);
    const bool second = active && r.outcome == FAST_MISS && check_rc;
    if (__ballot(second) != 0) {  // wave-uniform
        const kmer_w<1> x_rc = kmer_revcomp<1>(x, d.k);
        const minimizer_t mr = compute_minimizer<1>(x_rc, d.k, d.m, d.hash_magic);
        const fast_t r2 = fast_probe_regular_pairs(d, x_rc, mr, true, fast_resolve_pairs(d, mr.value, second), second);
        if (second) {
            r = r2;
            r.orientation = -1;
        }
    }
    return r;
}

template <int W, bool CANON>
__device__ __forceinline__ fast_t fast_lookup_one(dict_view const& d, kmer_w<W> const& x, bool check_rc) {
    if constexpr (CANON) {
        const kmer_w<W> x_rc = kmer_revcomp<W>(x, d.k);
        const minimizer_t mf = compute_minimizer<W>(x, d.k, d.m, d.hash_magic);
        const minimizer_t mr = compute_minimizer<W>(x_rc, d.k, d.m, d.hash_magic);
        if (mf.value == mr.value) return fast_unsettled(true);  // tie: both alignments (src/dictionary.cpp:35-40)
        return fast_probe_canonical<W>(d, x, x_rc, mf.value < mr.value ? mf : mr);
    } else {
        const minimizer_t mf = compute_minimizer<W>(x, d.k, d.m, d.hash_magic);
        /* (Asking for both strands' directory buckets at once -- three of four queries of the benchmark mix need the second
           answer anyway -- shortens the chain by one read and wastes a line on every forward hit: measured 12.1 against 13.0 G
           lookups/s on the C3 stand-in, profiles/r03/; the strands stay one after the other. Round 6 measured the same trade on the
           MPHF path, whose chains are three reads long -- both pilots together, then both codewords, fast_lookup_pairs --: 10.8 against
           11.96 G lookups/s, same box, two alternating rounds, profiles/r06/mphf_both_strands_together_ab.txt. These paths are bound by
           the REQUESTS they make of the memory system, not by the length of their chains.) */
        fast_t r = fast_probe_regular<W>(d, x, mf, false, fast_resolve(d, mf.value));
        if (r.outcome == FAST_MISS && check_rc) {
            const kmer_w<W> x_rc = kmer_revcomp<W>(x, d.k);
            const minimizer_t mr = compute_minimizer<W>(x_rc, d.k, d.m, d.hash_magic);
            r = fast_probe_regular<W>(d, x_rc, mr, true, fast_resolve(d, mr.value));
            r.orientation = -1;
        }
        return r;
    }
}

/* ---- lookup through the super-k-mer table (device_layout.hpp (5)) -------------------------------
   A probe follows the key's bucket sequence: HIT / final MISS / DEFER to the complete path. `key_seen`:
   some slot on the way carried the key's fingerprint; a MISS without it proves that no k-mer with this
   key is in the dictionary (the streaming query's negative short-cut uses that). */
/* may this replica's table settle a k-mer with key `kk`? (not on a tie; not a key owned by another table shard) */
__device__ __forceinline__ bool sk_usable(dict_view const& d, sk_key_t const& kk) {
    return !kk.tie && (d.sk.num_shards <= 1 || sk_owner(kk.key, d.sk.num_shards) == d.sk.shard_id);
}

/* what a probe compares the slots of a bucket with */
template <int W>
struct sk_query_t {
    kmer_w<W> y, y_rc;     // the k-mer as read on the key's strand, and its reverse complement
    uint32_t j;            // where the key starts in y
    uint32_t fingerprint;  // of the key
    bool s;                // the key was read on the reverse complement of x
};

template <int W>
__device__ __forceinline__ sk_query_t<W> sk_make_query(kmer_w<W> const& x, kmer_w<W> const& x_rc, sk_key_t const& kk,
                                                        uint32_t fingerprint) {
    sk_query_t<W> q;
    q.s = kk.rc;
    q.j = kk.pos;
    q.y = kk.rc ? x_rc : x;
    q.y_rc = kk.rc ? x : x_rc;
    q.fingerprint = fingerprint;
    return q;
}

template <int W>
__device__ __forceinline__ kmer_w<W> kmer_pick(bool first, kmer_w<W> const& a, kmer_w<W> const& b) {
    kmer_w<W> out;
    for (int t = 0; t < W; ++t) out.w[t] = first ? a.w[t] : b.w[t];
    return out;
}

/* c-th bucket of a key */
__device__ __forceinline__ uint32_t sk_choice(sk_hash_t const& h, uint32_t c) {
    return c == 0 ? h.bucket[0] : c == 1 ? h.bucket[1] : c == 2 ? h.bucket[2] : c == 3 ? h.bucket[3] : h.bucket[4];
}

/* What slot 0 of a bucket says about the bucket as a whole. */
struct sk_bucket_flags {
    uint32_t go_on;      // the bucket's flag for the choice being examined
    bool second_used;    // slot 1 holds something (k <= 63: it lives in the bucket's second line)
};

/* One SLOT against one query. `piece(i)` yields the i-th 16-byte piece of the slot -- out of LDS, where the quad
   staged the line, or out of global memory. Straight-line code (selects, no branch). FIRST: the slot is slot 0 of
   its bucket and carries the bucket's flags. Out: r (a hit), `marker` (the slot says that the key is heavy: its
   k-mers are entered under keys of their own), `key_seen`. */
/* TRACK (the streaming query, streaming.hip): `lasts` is lowered to what THIS slot allows -- for how many of the k-mers that follow
   the query along its read and elect the same key occurrence (sk_key_persists) the slot is sure to miss as it misses now. A slot that
   does not carry the key's fingerprint has no say. One that does holds the strings around an occurrence of the key: the read lies
   against it at a fixed offset for as long as the key occurrence is the same, so a base at which the two differ keeps them apart
   while the k-mer still holds that base -- the LAST differing base of the k-mer when the read runs along the strings here, the first
   when it runs against them. No difference (the k-mer ends outside the super-k-mer's extent) or a marker: no promise.
   `inline_seen`: the slot carries the fingerprint and is not a marker (what a walk needs to know before it lets the k-mers behind a
   heavy key's marker start on their own sequences: streaming.hip). */
template <int W, bool FIRST, bool TRACK, class Piece>
__device__ __forceinline__ void sk_examine_slot_tracking(dict_view const& d, sk_query_t<W> const Q, uint32_t c, Piece piece, fast_t& r,
                                                         bool& key_seen, bool& marker, sk_bucket_flags& flags, uint32_t& lasts, bool& inline_seen) {
    const uint32_t km = d.k - d.sk.m;
    const uint32_t j = Q.j;
    /* values, not references into Q: a select between two members of a by-reference struct is compiled into an
       indexed load, which pins the struct in scratch memory */
    kmer_w<W> y, y_rc;
    for (int t = 0; t < W; ++t) {
        y.w[t] = Q.y.w[t];
        y_rc.w[t] = Q.y_rc.w[t];
    }
    const uint4 q0 = piece(0), q1 = piece(1);
    uint4 q2 = q1;
    if constexpr (W == 2) q2 = piece(2);
    const uint32_t meta = q0.x;
    if constexpr (FIRST) {
        /* decided now, in its own register: hipcc 7.2 has been seen recycling a slot word that is only
           consumed much later (HISTORY.md) */
        uint32_t go_on = meta & (SK_GO_ON << c);
        /* first choice: only if a key with this query's filter index went on from here (device_layout.hpp) */
        if (c == 0) go_on &= 0u - ((meta >> (SK_FILTER_SHIFT + sk_filter_index(Q.fingerprint))) & 1u);
        asm volatile("" : "+v"(go_on));
        flags.go_on = go_on;
        flags.second_used = (meta & SK_SECOND_USED) != 0;
        if constexpr (W == 2) {
            /* slot 1 is in the bucket's second line: worth fetching only if it holds this query's key (an item holding
               the query's k-mer has the query's key, so equal fingerprints are necessary for anything slot 1 could add) */
            const uint4 spare = piece(SK_SECOND_FINGERPRINT_WORD / 4);
            flags.second_used = flags.second_used && spare.x == Q.fingerprint;
        }
    }
    const bool valid = (meta & SK_VALID) != 0, is_marker = (meta & SK_MARKER) != 0;
    const bool same_fingerprint = valid && (q0.w >> 8) == Q.fingerprint;
    key_seen = key_seen || same_fingerprint;
    marker = marker || (same_fingerprint && is_marker);
    /* inline super-k-mer: the strings read the key forward (strand 0: y aligns, key at km - a)
       or reverse-complemented (strand 1: rc(y) aligns, its copy of the key sits at km - j).
       No fingerprint test: the k-mer comparison is the test. */
    const uint64_t at = uint64_t(q0.z) | (uint64_t(q0.w & 0xFFu) << 32);
    const bool o = (meta & SK_STRAND) != 0;
    const uint32_t a = o ? j : km - j;
    const uint64_t w0 = uint64_t(q1.x) | (uint64_t(q1.y) << 32), w1 = uint64_t(q1.z) | (uint64_t(q1.w) << 32);
    kmer_w<W> cand;
    if constexpr (W == 1) {
        cand.w[0] = funnel_shr(w0, w1, 2 * a);
    } else {
        const uint64_t w2 = uint64_t(q2.x) | (uint64_t(q2.y) << 32), w3 = uint64_t(q2.z) | (uint64_t(q2.w) << 32);
        const bool up = 2 * a >= 64;  // a <= 62: the k-mer starts in word 0 or 1
        const uint32_t sh = (2 * a) & 63u;
        const uint64_t e0 = up ? w1 : w0, e1 = up ? w2 : w1, e2 = up ? w3 : w2;
        cand.w[0] = funnel_shr(e0, e1, sh);
        cand.w[1] = funnel_shr(e1, e2, sh);
    }
    cand = kmer_take_chars<W>(cand, d.k);
    const uint32_t left = (meta >> SK_LEFT_SHIFT) & 63u, right = (meta >> SK_RIGHT_SHIFT) & 63u;
    const kmer_w<W> target = kmer_pick<W>(o, y_rc, y);
    const bool hit = valid && !is_marker && kmer_eq<W>(cand, target) && a + left >= km && a <= right;
    /* a k-mer occurs once in the strings: at most one slot hits */
    r.kmer_offset = hit ? at + a - km : r.kmer_offset;
    r.string_id = hit ? q0.y : r.string_id;
    r.orientation = hit ? ((o != Q.s) ? int8_t(-1) : int8_t(1)) : r.orientation;
    r.outcome = hit ? int(FAST_HIT) : r.outcome;
    if constexpr (TRACK) {
        uint32_t first, last;  // the first and the last base at which the k-mer and the slot differ (only read when they do)
        bool differ;
        if constexpr (W == 1) {
            const uint64_t x0 = cand.w[0] ^ target.w[0];
            differ = x0 != 0;
            first = uint32_t(__builtin_ctzll(x0 | (uint64_t(1) << 63))) >> 1;
            last = (63u - uint32_t(__builtin_clzll(x0 | 1u))) >> 1;
        } else {
            const uint64_t x0 = cand.w[0] ^ target.w[0], x1 = cand.w[1] ^ target.w[1];
            differ = (x0 | x1) != 0;
            first = x0 ? uint32_t(__builtin_ctzll(x0)) >> 1 : 32u + (uint32_t(__builtin_ctzll(x1 | (uint64_t(1) << 63))) >> 1);
            last = x1 ? 32u + ((63u - uint32_t(__builtin_clzll(x1))) >> 1) : (63u - uint32_t(__builtin_clzll(x0 | 1u))) >> 1;
        }
        const bool along = o == Q.s;  // target is the read's own k-mer: base i of it is base i of the read's k-mer; else base k - 1 - i
        uint32_t mine = along ? last : d.k - 1 - first;
        mine = (differ && !is_marker) ? mine : 0u;
        lasts = (same_fingerprint && mine < lasts) ? mine : lasts;
        inline_seen = inline_seen || (same_fingerprint && !is_marker);  // an occurrence of (a key with) the query's fingerprint held inline
    }
}

template <int W, bool FIRST, class Piece>
__device__ __forceinline__ void sk_examine_slot(dict_view const& d, sk_query_t<W> const Q, uint32_t c, Piece piece, fast_t& r,
                                                bool& key_seen, bool& marker, sk_bucket_flags& flags) {
    uint32_t unused = 0;
    bool unused_too = false;
    sk_examine_slot_tracking<W, FIRST, false>(d, Q, c, piece, r, key_seen, marker, flags, unused, unused_too);
}

/* k <= 63: one ENTRY of the k-mers' region (device_layout.hpp: 32 bytes -- meta, string id, position | fingerprint, the
   k-mer as the strings spell it) against one query; two entries make a bucket, one 64-byte line. `piece(0)`, `piece(1)`: the
   entry's two 16-byte pieces. FIRST: entry 0 carries the bucket's go-on flags and filter, like slot 0 of the keys' region. */
template <bool FIRST, class Piece>
__device__ __forceinline__ void sk_examine_kmer_entry(sk_query_t<2> const Q, uint32_t c, Piece piece, fast_t& r, sk_bucket_flags& flags) {
    kmer_w<2> y, y_rc;
    for (int t = 0; t < 2; ++t) {
        y.w[t] = Q.y.w[t];
        y_rc.w[t] = Q.y_rc.w[t];
    }
    const uint4 q0 = piece(0), q1 = piece(1);
    const uint32_t meta = q0.x;
    if constexpr (FIRST) {
        uint32_t go_on = meta & (SK_GO_ON << c);
        if (c == 0) go_on &= 0u - ((meta >> (SK_FILTER_SHIFT + sk_filter_index(Q.fingerprint))) & 1u);
        asm volatile("" : "+v"(go_on));
        flags.go_on = go_on;
        flags.second_used = false;  // (both entries are in the line at hand)
    }
    kmer_w<2> body;
    body.w[0] = uint64_t(q1.x) | (uint64_t(q1.y) << 32);
    body.w[1] = uint64_t(q1.z) | (uint64_t(q1.w) << 32);
    const bool valid = (meta & SK_VALID) != 0;
    const bool as_y = valid && kmer_eq<2>(body, y), as_rc = valid && kmer_eq<2>(body, y_rc);
    const bool hit = as_y || as_rc;
    r.kmer_offset = hit ? (uint64_t(q0.z) | (uint64_t(q0.w & 0xFFu) << 32)) : r.kmer_offset;
    r.string_id = hit ? q0.y : r.string_id;
    r.orientation = hit ? ((as_rc != Q.s) ? int8_t(-1) : int8_t(1)) : r.orientation;
    r.outcome = hit ? int(FAST_HIT) : r.outcome;
}

/* k <= 31: one LINE of the k-mers' region (device_layout.hpp: a flags word and three 20-byte entries) against one query.
   `word(i)`: the line's i-th dword, out of LDS or out of registers. */
template <class Word>
__device__ __forceinline__ void sk_examine_kmer_line(sk_query_t<1> const Q, uint32_t c, Word word, fast_t& r, sk_bucket_flags& flags) {
    const uint64_t y = Q.y.w[0], y_rc = Q.y_rc.w[0];
    const uint32_t meta = word(0);
    uint32_t go_on = meta & (SK_GO_ON << c);
    if (c == 0) go_on &= 0u - ((meta >> (SK_FILTER_SHIFT + sk_filter_index(Q.fingerprint))) & 1u);
    flags.go_on = go_on;
    flags.second_used = false;
#pragma unroll
    for (uint32_t e = 0; e < SK_KMER_ENTRIES_NARROW; ++e) {
        const uint32_t at = 1 + SK_KMER_ENTRY_WORDS * e;
        const uint64_t kmer = uint64_t(word(at)) | (uint64_t(word(at + 1)) << 32);
        const bool valid = ((meta >> e) & 1u) != 0;
        const bool as_y = valid && kmer == y, as_rc = valid && kmer == y_rc;
        const bool hit = as_y || as_rc;
        r.kmer_offset = hit ? (uint64_t(word(at + 2)) | (uint64_t(word(at + 4) & 0xFFu) << 32)) : r.kmer_offset;
        r.string_id = hit ? word(at + 3) : r.string_id;
        r.orientation = hit ? ((as_rc != Q.s) ? int8_t(-1) : int8_t(1)) : r.orientation;
        r.outcome = hit ? int(FAST_HIT) : r.outcome;
    }
}

/* where the bucket with (global) index b starts, in bytes from d.sk.slots: the keys' region holds 64 W bytes per bucket; behind it,
   at k <= 63, the k-mers' region holds 64 (two compact entries) */
template <int W>
__device__ __forceinline__ uint64_t sk_bucket_offset(dict_view const& d, uint32_t b, bool kmer_region) {
    if constexpr (W == 2) {
        if (kmer_region) return uint64_t(d.sk.num_buckets) * 128 + uint64_t(b - d.sk.num_buckets) * 64;
    }
    return uint64_t(b) * (64 * W);
}

/* Where a probe stands: the bucket sequence it follows (its key's, or -- once it has met its key's marker -- its
   k-mer's own), the next choice of it, and the choice of the key's sequence to come back to if the k-mer's sequence
   ends without the k-mer (the marker may have been another key's with an equal fingerprint). */
constexpr uint32_t SK_NO_RETURN = 0xFFu;

struct sk_walk_t {
    sk_hash_t h;
    uint32_t c;
    uint32_t back_to;  // SK_NO_RETURN: nothing to come back to
    bool on_kmer_sequence;
};

__device__ __forceinline__ sk_walk_t sk_walk_begin(sk_hash_t const& h, uint32_t c) {
    sk_walk_t w;
    w.h = h;
    w.c = c;
    w.back_to = SK_NO_RETURN;
    w.on_kmer_sequence = false;
    return w;
}

template <int W>
__device__ __forceinline__ void sk_walk_to_kmer_sequence(dict_view const& d, kmer_w<W> const& x, kmer_w<W> const& x_rc, sk_walk_t& w,
                                                         sk_query_t<W>& Q, bool go_on) {
    w.back_to = go_on ? w.c + 1 : SK_NO_RETURN;
    w.h = sk_hash_kmer_region(sk_kmer_key<W>(x, x_rc), d.sk.num_buckets, d.sk.kmer_buckets);
    w.c = 0;
    w.on_kmer_sequence = true;
    Q.fingerprint = w.h.fingerprint;
}

/* After a bucket has been examined: true = another bucket (w.c of w.h) is to be read; false = r is final (FAST_MISS
   stands for a final miss). */
template <int W>
__device__ __forceinline__ bool sk_walk_step(dict_view const& d, kmer_w<W> const& x, kmer_w<W> const& x_rc, sk_key_t const& kk,
                                             sk_walk_t& w, sk_query_t<W>& Q, fast_t& r, uint32_t go_on, bool marker) {
    if (r.outcome != FAST_MISS) return false;
    if (marker && !w.on_kmer_sequence) {
        sk_walk_to_kmer_sequence<W>(d, x, x_rc, w, Q, go_on != 0);
        return true;
    }
    if (go_on != 0) {
        if (++w.c < SK_CHOICES) return true;
        r.outcome = FAST_DEFER;  // a key (or k-mer) that found no slot: complete path
        return false;
    }
    if (w.on_kmer_sequence && w.back_to != SK_NO_RETURN) {
        /* the k-mer is not under its own key: what is left is the rest of the key's sequence */
        w.c = w.back_to;
        w.back_to = SK_NO_RETURN;
        if (w.c >= SK_CHOICES) {
            r.outcome = FAST_DEFER;
            return false;
        }
        w.h = sk_hash(kk.key, d.sk.num_buckets);
        Q.fingerprint = w.h.fingerprint;
        return true;
    }
    return false;
}

/* ---- the same probe by a whole wave: quad-cooperative line fetch through LDS -----------------------------------
   Must be called by all 64 lanes of the wave together (lanes without a query pass need = false). Round p of a
   bucket fetch: the four lanes of every quad read the four 16-byte pieces of ONE 64-byte line -- the bucket of the
   quad's lane p & 3 -- with one load instruction, and lane L deposits its piece at region p + 16 L of the wave's
   staging area: quad q's line lands contiguously at region p + 64 q, which transposes "lane = piece" into "lane =
   owner of the whole line". 4 W rounds fetch the buckets of all 64 lanes. The memory pipeline sees one request (and
   one address translation) per line instead of one per 16-byte load (device_layout.hpp (5), HISTORY.md). */

template <int OWNER>
__device__ __forceinline__ uint32_t quad_broadcast(uint32_t v) {
    return uint32_t(__builtin_amdgcn_mov_dpp(int(v), OWNER * 0x55, 0xf, 0xf, true));
}

/* Every lane with need = true gets its bucket into its place of the wave's staging area. The broadcasts and the loads
   are executed by ALL lanes, unconditionally: a quad whose owner of the round has no use for a bucket fetches bucket 0
   (one line, shared by all such quads: an L2 hit) into a place nobody reads.

   The pieces move through registers (global_load_dwordx4 + ds_write_b128). Moving them straight into LDS
   (global_load_lds_dwordx4) measured the same speed and miscompiled under hipcc 7.2 -- the repro and the micro-benchmarks
   are kept in tools/debug/ (glds_check, m0_check, vcc_check; HISTORY.md), the code path is gone. */

/* Between two phases of a wave that talk through LDS: everything this lane has issued has completed, and the compiler moves
   no memory operation across. (The "wavefront" fences used elsewhere in this file order the accesses for the compiler and rely on
   the hardware executing a wave's DS instructions in order; here an LDS word written by one lane steers a GLOBAL load of another
   and the loaded line goes back through LDS -- the explicit wait keeps the hand-over independent of what the compiler infers about it,
   tools/debug/member_mismatch.py.) */
__device__ __forceinline__ void sk_wave_sync() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

/* LINE: which 64-byte line of the bucket (k <= 31: the bucket is one line holding both slots; k <= 63: line 0 = slot 0,
   line 1 = slot 1, fetched only by the lanes that still need it) */
template <int W, bool MIXED = false>
__device__ __forceinline__ void sk_stage_lines(dict_view const& d, uint32_t bucket, uint32_t line, bool need, uint4* wave_stage, bool kmer_region = false) {
    char const* slots = static_cast<char const*>(d.sk.slots);
    const uint32_t sub = threadIdx.x & 3u;
    /* a 32-bit line number would overflow at k <= 63 (2 lines per bucket, up to 2^32 buckets): keep bucket and line apart */
    const uint32_t mine = need ? bucket : 0u, my_line = need ? line : 0u;
    const uint32_t b0 = quad_broadcast<0>(mine), b1 = quad_broadcast<1>(mine), b2 = quad_broadcast<2>(mine), b3 = quad_broadcast<3>(mine);
    uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    if constexpr (W == 2) {
        l0 = quad_broadcast<0>(my_line);
        l1 = quad_broadcast<1>(my_line);
        l2 = quad_broadcast<2>(my_line);
        l3 = quad_broadcast<3>(my_line);
    }
    uint64_t a0 = uint64_t(b0) * (64 * W) + 64 * l0, a1 = uint64_t(b1) * (64 * W) + 64 * l1, a2 = uint64_t(b2) * (64 * W) + 64 * l2,
             a3 = uint64_t(b3) * (64 * W) + 64 * l3;
    if constexpr (W == 2 && MIXED) {
        /* some owners are on their k-mer's sequence: those buckets are single lines of the k-mers' region (sk_bucket_offset) */
        const uint32_t mine_region = need && kmer_region ? 1u : 0u;
        const uint64_t base = uint64_t(d.sk.num_buckets) * 128, first = d.sk.num_buckets;
        if (quad_broadcast<0>(mine_region)) a0 = base + uint64_t(b0 - first) * 64;
        if (quad_broadcast<1>(mine_region)) a1 = base + uint64_t(b1 - first) * 64;
        if (quad_broadcast<2>(mine_region)) a2 = base + uint64_t(b2 - first) * 64;
        if (quad_broadcast<3>(mine_region)) a3 = base + uint64_t(b3 - first) * 64;
    }
    const uint32_t lane = threadIdx.x & 63u;
    /* nontemporal: a bucket line is not read again before a few hundred million others have passed */
    const uint4 p0 = sk_load_piece(slots + a0 + 16 * sub), p1 = sk_load_piece(slots + a1 + 16 * sub),
                p2 = sk_load_piece(slots + a2 + 16 * sub), p3 = sk_load_piece(slots + a3 + 16 * sub);
    wave_stage[0 * 64 + lane] = p0;
    wave_stage[1 * 64 + lane] = p1;
    wave_stage[2 * 64 + lane] = p2;
    wave_stage[3 * 64 + lane] = p3;
    /* other lanes of this wave read what this lane wrote: order the LDS accesses for the compiler (the hardware
       executes a wave's DS instructions in order) */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* One bucket for every lane with need = true: fetch (cooperatively) and examine. Called by all 64 lanes. k <= 31: one
   line, both slots compared. k <= 63: slot 0's line; the lanes that neither hit nor can rule slot 1 out (its "in use" bit
   sits in slot 0) fetch the second line in a second round -- 1.3 lines per probe instead of 2. */
template <int W, bool MIXED = false>
__device__ __forceinline__ void sk_probe_bucket_wave(dict_view const& d, sk_query_t<W> const& Q, uint32_t bucket, uint32_t c, bool need,
                                                     uint4* wave_stage, fast_t& r, bool& key_seen, bool& marker, uint32_t& go_on,
                                                     bool kmer_region = false) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint4* mine = wave_stage + (lane & 3u) * 64 + (lane >> 2) * 4;  // region (lane & 3), line of quad (lane >> 2)
    sk_bucket_flags flags;
    flags.go_on = 0;
    flags.second_used = false;
    sk_stage_lines<W, MIXED>(d, bucket, 0u, need, wave_stage, kmer_region);
    bool compact = false;
    if constexpr (MIXED) compact = kmer_region;  // MIXED: lanes on their k-mer's sequence (compact entries) among the others
    if (need && !compact) {
        sk_examine_slot<W, true>(d, Q, c, [mine](uint32_t i) { return mine[i]; }, r, key_seen, marker, flags);
        if constexpr (W == 1) sk_examine_slot<W, false>(d, Q, c, [mine](uint32_t i) { return mine[2 + i]; }, r, key_seen, marker, flags);
    }
    if constexpr (W == 2 && MIXED) {
        if (need && compact) {
            sk_examine_kmer_entry<true>(Q, c, [mine](uint32_t i) { return mine[i]; }, r, flags);
            sk_examine_kmer_entry<false>(Q, c, [mine](uint32_t i) { return mine[2 + i]; }, r, flags);
        }
    }
    if constexpr (W == 1 && MIXED) {
        if (need && compact) {
            const uint32_t* words = reinterpret_cast<const uint32_t*>(mine);
            sk_examine_kmer_line(Q, c, [words](uint32_t i) { return words[i]; }, r, flags);
        }
    }
    if constexpr (W == 2) {
        /* Slot 1's line, for the fifth of the lanes whose key's fingerprint is there. Not another four rounds of the whole wave
           (rounds 2-3: 4 loads a lane, 8 broadcasts, 4 LDS writes, for 13 lines of 64 on average): the wave's 16 quads become
           fetch units, as in sk_finish_in_wave -- the lanes that want a line are ranked with a ballot, lane number q of them
           posts its bucket for quad q, ONE load instruction of all 64 lanes fetches those lines, the owners examine them out
           of LDS; a second turn when more than 16 lanes want one (one wave in eight). The first pass at k <= 63 is bound by
           its instructions: this takes some thirty of them off every wave. */
        bool second = need && r.outcome == FAST_MISS && flags.second_used;
        uint32_t* posted = reinterpret_cast<uint32_t*>(wave_stage + 64);  // (behind the 64 pieces of the lines fetched below; slot 0's lines are spent)
        char const* slots = static_cast<char const*>(d.sk.slots);
#pragma unroll 1
        for (;;) {
            sk_wave_sync();  // every lane is done with what the staging area held
            const uint64_t mask = __ballot(second);
            if (mask == 0) break;  // wave-uniform
            const uint32_t rank = uint32_t(__popcll(mask & ((uint64_t(1) << lane) - 1)));
            const bool served = second && rank < 16;
            if (lane < 16) posted[lane] = 0u;
            sk_wave_sync();
            if (served) posted[rank] = bucket;
            sk_wave_sync();
            const uint32_t b = posted[lane >> 2];
            wave_stage[lane] = sk_load_piece(slots + uint64_t(b) * 128 + 64 + 16 * (lane & 3u));  // quad q's line at wave_stage[4q .. 4q+3]
            sk_wave_sync();
            if (served) {
                const uint4* line = wave_stage + 4 * rank;
                sk_examine_slot<W, false>(d, Q, c, [line](uint32_t i) { return line[i]; }, r, key_seen, marker, flags);
                second = false;
            }
        }
    }
    go_on = flags.go_on;
    __builtin_amdgcn_wave_barrier();
}

/* what the first pass hands to the second in a queue entry, above the query's index */
constexpr uint32_t RESUME_CHOICE_SHIFT = 27;  // bits 27-29: choice of the key's sequence the entry refers to (the index: 27 bits)
constexpr uint32_t RESUME_HEAVY = 1u << 30;   // the key's marker was met there: start on the k-mer's own sequence ...
constexpr uint32_t RESUME_GO_ON = 1u << 31;   // ... and that bucket's go-on flag was set (the key's sequence continues)

/* First pass: ONE bucket read per query (choice 0 of its key's sequence). Settled: FAST_HIT / FAST_MISS (final);
   otherwise FAST_CONTINUE with r.kmer_offset = the flags of a queue entry (above), or FAST_DEFER. Called by all 64
   lanes (active = false: no query). `allow_rc` false (regular dictionary, check_reverse_complement off:
   src/dictionary.cpp:70-71) turns a hit on the other strand into a miss. `miss_orientation`: what a miss reports (-1
   after a regular dictionary's reverse-complement probe, src/dictionary.cpp:74-75). */
template <int W>
__device__ __forceinline__ fast_t sk_first_pass_wave(dict_view const& d, kmer_w<W> const& x, bool active, bool allow_rc,
                                                     int8_t miss_orientation, uint4* wave_stage) {
    const kmer_w<W> x_rc = kmer_revcomp<W>(x, d.k);
    const sk_key_t kk = sk_key<W>(x, x_rc, d.k, d.sk.m);
    const bool usable = active && sk_usable(d, kk);  // else: no strand-symmetric key, or a key of another table shard
    const sk_hash_t h = sk_hash(kk.key, d.sk.num_buckets);
    const sk_query_t<W> Q = sk_make_query<W>(x, x_rc, kk, h.fingerprint);
    fast_t r = fast_unsettled(active && !usable);
    uint32_t go_on = 0;
    bool marker = false, key_seen = false;
    sk_probe_bucket_wave<W>(d, Q, usable ? h.bucket[0] : 0u, 0u, usable, wave_stage, r, key_seen, marker, go_on);
    if constexpr (W == 2) {
        /* k <= 63, the heavy keys (a tenth of C4's k-mers): the k-mer's OWN first bucket, fetched in the wave (round 4). At k <= 63 the
           stragglers keep their compacted pass -- finishing every probe in the wave costs the registers of the whole walk and with them
           three waves per SIMD, -5.5 % --, but two thirds of the stragglers are lanes that met their key's marker, and what those need
           next is one line of two compact entries under a key that takes three registers to make: the k-mer's key, the first bucket of
           its sequence, its fingerprint. So that one hop is made here, with ranked fetches (as for slot 1's line), and only what is still
           open behind it -- the k-mer's first choice was full, or the marker was another key's with an equal fingerprint and the key's own
           sequence goes on -- joins the resume queue: fewer entries, fewer masked id rewrites. */
        bool hop = usable && r.outcome == FAST_MISS && marker;
        if (__ballot(hop) != 0) {  // wave-uniform
            const sk_hash_t hk = sk_hash_kmer_region(sk_kmer_key<W>(x, x_rc), d.sk.num_buckets, d.sk.kmer_buckets);  // (bucket[0] and the fingerprint: the rest is not computed)
            sk_query_t<W> Qk = Q;
            Qk.fingerprint = hk.fingerprint;
            const uint32_t lane = threadIdx.x & 63u;
            uint32_t* posted = reinterpret_cast<uint32_t*>(wave_stage + 64);
            char const* slots = static_cast<char const*>(d.sk.slots);
            uint32_t kmer_go_on = 0;
            bool pending = hop;
#pragma unroll 1
            for (;;) {
                sk_wave_sync();
                const uint64_t mask = __ballot(pending);
                if (mask == 0) break;  // wave-uniform
                const uint32_t rank = uint32_t(__popcll(mask & ((uint64_t(1) << lane) - 1)));
                const bool served = pending && rank < 16;
                if (lane < 16) posted[lane] = 0u;
                sk_wave_sync();
                if (served) posted[rank] = 2 * d.sk.num_buckets + (hk.bucket[0] - d.sk.num_buckets);  // a 64-byte line NUMBER: the keys' region is two lines a bucket
                sk_wave_sync();
                const uint32_t line = posted[lane >> 2];
                wave_stage[lane] = sk_load_piece(slots + uint64_t(line) * 64 + 16 * (lane & 3u));
                sk_wave_sync();
                if (served) {
                    const uint4* mine = wave_stage + 4 * rank;
                    sk_bucket_flags flags;
                    sk_examine_kmer_entry<true>(Qk, 0u, [mine](uint32_t i) { return mine[i]; }, r, flags);
                    sk_examine_kmer_entry<false>(Qk, 0u, [mine](uint32_t i) { return mine[2 + i]; }, r, flags);
                    kmer_go_on = flags.go_on;
                    pending = false;
                }
            }
            if (hop && r.outcome == FAST_MISS && kmer_go_on == 0) {
                /* the k-mer is not under its own key (sk_walk_step): what is left is the rest of the key's sequence, if it goes on */
                marker = false;
            }
        }
        /* (The other third of the stragglers -- the key's first bucket was full when the key was placed -- given one look at slot 0 of the
           key's second choice in the wave as well: measured in round 4 and not kept, 31.35 / 31.34 / 31.26 against 31.35 / 31.34 / 31.30 /
           31.34 G lookups/s, profiles/r04/second_choice_in_wave_k63_ab.txt: a 128-base slot comparison for the whole wave costs what the
           smaller resume pass gives back.) */
    }
    if (usable) {
        if (r.outcome == FAST_MISS) {
            if (marker) {
                r.outcome = FAST_CONTINUE;
                r.kmer_offset = RESUME_HEAVY | (go_on ? RESUME_GO_ON : 0u);
            } else if (go_on) {
                r.outcome = FAST_CONTINUE;
                r.kmer_offset = 1u << RESUME_CHOICE_SHIFT;
            } else {
                r.orientation = miss_orientation;
            }
        } else if (r.orientation < 0 && !allow_rc) {
            r = fast_unsettled(false);
            r.orientation = miss_orientation;
        }
    }
    return r;
}

/* The lanes of a wave that need ANOTHER bucket after the first (a twentieth of them: the key's first bucket was full when it
   was placed, or the key is heavy), served inside the wave that owns them. The quads of the wave become fetch units: the
   needy lanes are ranked with a ballot, needy lane number q (up to 16 a turn) posts its bucket's index for quad q, ONE load
   instruction of all 64 lanes fetches those lines (16 bytes a lane: one request and one translation a line, as in the first
   fetch; quads without a customer fetch bucket 0, an L2 hit), the pieces land in LDS and the owners examine their lines. No
   queue entry written and read back, no second kernel, and above all no placeholder id that a later pass rewrites: an 8-byte
   store into a line that has left the caches costs the DRAM a masked write -- a random access of its own, one of the 2.5 a
   resumed query cost in the resume pass (HISTORY.md). */
/* (k <= 31 only. At k <= 63 finishing in the wave measured as a loss -- round 3: 6 %, round 4, with ranked fetches everywhere: 5.5 %,
   profiles/r04/inwave_k63_ab.txt: 27.1 -> 25.6 G lookups/s. The k <= 63 first pass runs at the chip's random-line rate and what it
   needs for that is waves in flight: the loop's state takes the kernel from 60 to 86 registers, eight waves per SIMD to five. There
   the stragglers keep their own, compacted pass: resume_lookup_kernel.) */
__device__ __forceinline__ void sk_finish_in_wave(dict_view const& d, kmer_w<1> const& x, kmer_w<1> const& x_rc, sk_key_t const& kk, sk_walk_t& w,
                                                  sk_query_t<1>& Q, fast_t& r, bool need, uint4* wave_stage) {
    const uint32_t lane = threadIdx.x & 63u;
    char const* slots = static_cast<char const*>(d.sk.slots);
    uint32_t* posted = reinterpret_cast<uint32_t*>(wave_stage + 64);  // 16 bucket indices, behind the 64 pieces of the fetched lines
#pragma unroll 1
    for (;;) {
        const uint64_t mask = __ballot(need);
        if (mask == 0) break;  // wave-uniform
        const uint32_t rank = uint32_t(__popcll(mask & ((uint64_t(1) << lane) - 1)));
        const bool served = need && rank < 16;
        if (lane < 16) posted[lane] = 0u;
        sk_wave_sync();
        if (served) posted[rank] = sk_choice(w.h, w.c);
        sk_wave_sync();
        const uint32_t b = posted[lane >> 2];
        wave_stage[lane] = sk_load_piece(slots + uint64_t(b) * 64 + 16 * (lane & 3u));  // quad q's line at wave_stage[4q .. 4q+3]
        sk_wave_sync();
        if (served) {
            const uint4* mine = wave_stage + 4 * rank;
            sk_bucket_flags flags;
            flags.go_on = 0;
            flags.second_used = false;
            bool marker = false, key_seen = false;
            if (sk_choice(w.h, w.c) >= d.sk.num_buckets) {  // a bucket of the k-mers' region: three compact entries
                const uint32_t* words = reinterpret_cast<const uint32_t*>(mine);
                sk_examine_kmer_line(Q, w.c, [words](uint32_t i) { return words[i]; }, r, flags);
            } else {
                sk_examine_slot<1, true>(d, Q, w.c, [mine](uint32_t i) { return mine[i]; }, r, key_seen, marker, flags);
                sk_examine_slot<1, false>(d, Q, w.c, [mine](uint32_t i) { return mine[2 + i]; }, r, key_seen, marker, flags);
            }
            const uint32_t go_on = flags.go_on;
            need = sk_walk_step<1>(d, x, x_rc, kk, w, Q, r, go_on, marker);
        }
        sk_wave_sync();  // the staging area is rewritten by the next turn
    }
}

/* The whole table lookup of a wave in one call: the first bucket by all lanes (sk_probe_bucket_wave), whatever is left by
   sk_finish_in_wave. Returns FAST_HIT / FAST_MISS (final) / FAST_DEFER; same arguments as sk_first_pass_wave. */
template <int W>
__device__ __forceinline__ fast_t sk_lookup_in_wave(dict_view const& d, kmer_w<W> const& x, bool active, bool allow_rc, int8_t miss_orientation,
                                                    uint4* wave_stage) {
    const kmer_w<W> x_rc = kmer_revcomp<W>(x, d.k);
    const sk_key_t kk = sk_key<W>(x, x_rc, d.k, d.sk.m);
    const bool usable = active && sk_usable(d, kk);
    sk_walk_t w = sk_walk_begin(sk_hash(kk.key, d.sk.num_buckets), 0);
    sk_query_t<W> Q = sk_make_query<W>(x, x_rc, kk, w.h.fingerprint);
    fast_t r = fast_unsettled(active && !usable);
    uint32_t go_on = 0;
    bool marker = false, key_seen = false;
    sk_probe_bucket_wave<W>(d, Q, usable ? w.h.bucket[0] : 0u, 0u, usable, wave_stage, r, key_seen, marker, go_on);
    sk_wave_sync();
    bool need = false;
    if (usable) need = sk_walk_step<W>(d, x, x_rc, kk, w, Q, r, go_on, marker);
    sk_finish_in_wave(d, x, x_rc, kk, w, Q, r, need, wave_stage);
    if (usable && (r.outcome == FAST_MISS || (r.outcome == FAST_HIT && r.orientation < 0 && !allow_rc))) {
        r = fast_unsettled(false);
        r.orientation = miss_orientation;
    }
    return r;
}

/* Second pass: a queue entry of the first pass, walked to the end. Same contract; never returns FAST_CONTINUE. */
template <int W>
__device__ __forceinline__ fast_t sk_second_pass_wave(dict_view const& d, kmer_w<W> const& x, bool active, uint32_t entry, bool allow_rc,
                                                      int8_t miss_orientation, uint4* wave_stage) {
    const kmer_w<W> x_rc = kmer_revcomp<W>(x, d.k);
    const sk_key_t kk = sk_key<W>(x, x_rc, d.k, d.sk.m);
    sk_walk_t w = sk_walk_begin(sk_hash(kk.key, d.sk.num_buckets), (entry >> RESUME_CHOICE_SHIFT) & 7u);
    sk_query_t<W> Q = sk_make_query<W>(x, x_rc, kk, w.h.fingerprint);
    if (entry & RESUME_HEAVY) sk_walk_to_kmer_sequence<W>(d, x, x_rc, w, Q, (entry & RESUME_GO_ON) != 0);
    fast_t r = fast_unsettled(false);
    bool need = active;
#pragma unroll 1
    while (__ballot(need) != 0) {  // wave-uniform
        uint32_t go_on = 0;
        bool marker = false, key_seen = false;
        const uint32_t bucket = need ? sk_choice(w.h, w.c) : 0u;
        sk_probe_bucket_wave<W, true>(d, Q, bucket, w.c, need, wave_stage, r, key_seen, marker, go_on, bucket >= d.sk.num_buckets);
        if (need) need = sk_walk_step<W>(d, x, x_rc, kk, w, Q, r, go_on, marker);
    }
    if (r.outcome == FAST_MISS || (r.outcome == FAST_HIT && r.orientation < 0 && !allow_rc)) {
        r = fast_unsettled(false);
        r.orientation = miss_orientation;
    }
    return r;
}

/* Append to one of the sharded queues of the multi-pass lookup: one atomic per wave, the pushing lanes take
   consecutive places. Returns the place (>= capacity: the queue is full, nothing may be written). */
__device__ __forceinline__ uint32_t wave_queue_place(bool push, uint32_t* counter) {
    const uint64_t mask = __ballot(push);
    if (mask == 0) return 0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t leader = uint32_t(__ffsll((unsigned long long)mask)) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, uint32_t(__popcll(mask)));
    base = __shfl(base, int(leader), 64);
    return base + uint32_t(__popcll(mask & ((uint64_t(1) << lane) - 1)));
}

/* hit -> lookup_result fields (include/offsets.hpp:138-154, spss.hpp:226-228) */
template <bool FULL>
__device__ __forceinline__ void store_result(dict_view const& d, result_view const& out, uint64_t i, hit_t const& h) {
    __builtin_nontemporal_store(h.found ? h.kmer_offset - uint64_t(h.string_id) * (d.k - 1) : INVALID_U64, out.kmer_id + i);
    if constexpr (FULL) {
        uint64_t begin = INVALID_U64, end = INVALID_U64;
        if (h.found && (out.string_begin || out.string_end || out.kmer_id_in_string)) {
            begin = d.endpoints[h.string_id];
            end = d.endpoints[h.string_id + 1];
        }
        if (out.kmer_id_in_string) __builtin_nontemporal_store(h.found ? h.kmer_offset - begin : INVALID_U64, out.kmer_id_in_string + i);
        if (out.kmer_offset) __builtin_nontemporal_store(h.found ? h.kmer_offset : INVALID_U64, out.kmer_offset + i);
        if (out.string_id) __builtin_nontemporal_store(h.found ? uint64_t(h.string_id) : INVALID_U64, out.string_id + i);
        if (out.string_begin) __builtin_nontemporal_store(begin, out.string_begin + i);
        if (out.string_end) __builtin_nontemporal_store(end, out.string_end + i);
        if (out.kmer_orientation) __builtin_nontemporal_store(h.orientation, out.kmer_orientation + i);
        if (out.minimizer_found) __builtin_nontemporal_store(uint8_t(h.minimizer_found ? 1 : 0), out.minimizer_found + i);
    }
}

}  // namespace sshash_amd
