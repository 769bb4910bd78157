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

__device__ __forceinline__ uint64_t funnel_shr(uint64_t lo, uint64_t hi, uint32_t s /* 0..63 */) {
    return (lo >> s) | ((hi << 1) << (63 - s));
}

template <int W>
__device__ __forceinline__ window_t<W> read_window(granule const* __restrict__ granules, uint64_t off, uint32_t k) {
    const uint4* G = reinterpret_cast<const uint4*>(granules) + (off >> 5);
    const uint32_t r = uint32_t(off) & 31u;
    const uint32_t s = 2 * r;
    const uint4 g0 = G[0];
    const uint4 g1 = G[1];
    const uint64_t b0 = uint64_t(g0.z) | (uint64_t(g0.w) << 32);
    const uint64_t b1 = uint64_t(g1.z) | (uint64_t(g1.w) << 32);
    window_t<W> w;
    uint64_t following;  // mark bits of the positions off+1, off+2, ...
    if constexpr (W == 1) {
        w.kmer.w[0] = funnel_shr(b0, b1, s) & low_mask(2 * k);
        const uint64_t marks = uint64_t(g0.y) | (uint64_t(g1.y) << 32);
        following = marks >> (r + 1);
    } else {
        const uint4 g2 = G[2];
        const uint64_t b2 = uint64_t(g2.z) | (uint64_t(g2.w) << 32);
        w.kmer.w[0] = funnel_shr(b0, b1, s);
        w.kmer.w[1] = funnel_shr(b1, b2, s);
        w.kmer = kmer_take_chars<2>(w.kmer, k);
        const uint64_t m_lo = uint64_t(g0.y) | (uint64_t(g1.y) << 32);
        const uint64_t m_hi = uint64_t(g2.y);
        following = (m_lo >> (r + 1)) | (m_hi << (63 - r));
    }
    w.crosses = (following & low_mask(k - 1)) != 0;
    w.string_id = g0.x + __popc(g0.y & uint32_t((uint64_t(2) << r) - 1)) - 1;
    return w;
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
    bool other_key;  // the codeword's fingerprint proves the bucket belongs to another minimizer
};

/* minimizer -> MPHF -> control codeword -> bucket (include/sparse_and_skew_index.hpp:112-137) */
template <int W>
__device__ __forceinline__ bucket_t resolve_bucket(dict_view const& d, skew_part_dev const* __restrict__ skew,
                                                   uint64_t minimizer, kmer_w<W> const& skew_key) {
    bucket_t b;
    b.begin = 0;
    b.size = 1;
    b.heavy = false;
    b.valid = true;
    b.other_key = false;
    b.first_offset = 0;
    const uint64_t id = mphf_eval(d.minimizers, city128_u64(minimizer, d.minimizers.seed));
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
    if ((code & 1) == 0) {  // SINGLETON
        b.first_offset = code >> 1;
    } else if ((code & 3) == 1) {  // MIDLOAD
        b.size = uint32_t((code >> 2) & (MAX_BUCKET_SMALL - 1)) + 2;
        b.begin = uint64_t(d.begin_buckets_of_size[b.size]) + (code >> (2 + MIN_L)) * b.size;
        b.first_offset = packed_get(d.mid_load, b.begin, d.off_width);
    } else {  // HEAVYLOAD: second MPHF keyed by the k-mer (:34-44)
        b.heavy = true;
        const skew_part_dev sp = skew[(code >> 2) & 7];
        const uint64_t kid = mphf_eval(sp.f, city128_kmer<W>(skew_key, sp.f.seed));
        const uint64_t at = (code >> 5) + packed_get(sp.positions, kid, sp.pos_width);
        /* for a k-mer that is not a key the position is arbitrary and may fall outside the
           array (spectrum_preserving_string_set.hpp:51-64): treat as a miss */
        b.valid = at < d.heavy_size;
        b.first_offset = b.valid ? packed_get(d.heavy_load, at, d.off_width) : 0;
    }
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
__device__ __forceinline__ hit_t probe_regular(dict_view const& d, skew_part_dev const* __restrict__ skew,
                                               kmer_w<W> const& x, minimizer_t mini) {
    const bucket_t b = resolve_bucket<W>(d, skew, mini.value, x);
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

/* spss::lookup_canonical (include/spectrum_preserving_string_set.hpp:75-112,237-275) */
template <int W>
__device__ __forceinline__ hit_t probe_canonical(dict_view const& d, skew_part_dev const* __restrict__ skew,
                                                 kmer_w<W> const& x, kmer_w<W> const& x_rc, minimizer_t mini) {
    const kmer_w<W> key = kmer_less<W>(x_rc, x) ? x_rc : x;  // src/dictionary.cpp:53
    const bucket_t b = resolve_bucket<W>(d, skew, mini.value, key);
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

/* dictionary::lookup(Kmer, bool) -- src/dictionary.cpp:64-78 and :24-42 */
template <int W, bool CANON>
__device__ __forceinline__ hit_t lookup_one(dict_view const& d, skew_part_dev const* __restrict__ skew,
                                            kmer_w<W> const& x, bool check_rc) {
    if constexpr (CANON) {
        const kmer_w<W> x_rc = kmer_revcomp<W>(x, d.k);
        const minimizer_t mf = compute_minimizer<W>(x, d.k, d.m, d.hash_magic);
        const minimizer_t mr = compute_minimizer<W>(x_rc, d.k, d.m, d.hash_magic);
        if (mf.value < mr.value) return probe_canonical<W>(d, skew, x, x_rc, mf);
        if (mr.value < mf.value) return probe_canonical<W>(d, skew, x, x_rc, mr);
        hit_t h = probe_canonical<W>(d, skew, x, x_rc, mf);
        if (!h.found) h = probe_canonical<W>(d, skew, x, x_rc, mr);
        return h;
    } else {
        hit_t h = probe_regular<W>(d, skew, x, compute_minimizer<W>(x, d.k, d.m, d.hash_magic));
        if (!h.found && check_rc) {
            const kmer_w<W> x_rc = kmer_revcomp<W>(x, d.k);
            h = probe_regular<W>(d, skew, x_rc, compute_minimizer<W>(x_rc, d.k, d.m, d.hash_magic));
            h.orientation = -1;
        }
        return h;
    }
}

/* hit -> lookup_result fields (include/offsets.hpp:138-154, spss.hpp:226-228) */
template <bool FULL>
__device__ __forceinline__ void store_result(dict_view const& d, result_view const& out, uint64_t i, hit_t const& h) {
    if (h.found) {
        out.kmer_id[i] = h.kmer_offset - uint64_t(h.string_id) * (d.k - 1);
    } else {
        out.kmer_id[i] = INVALID_U64;
    }
    if constexpr (FULL) {
        uint64_t begin = INVALID_U64, end = INVALID_U64;
        if (h.found && (out.string_begin || out.string_end || out.kmer_id_in_string)) {
            begin = d.endpoints[h.string_id];
            end = d.endpoints[h.string_id + 1];
        }
        if (out.kmer_id_in_string) out.kmer_id_in_string[i] = h.found ? h.kmer_offset - begin : INVALID_U64;
        if (out.kmer_offset) out.kmer_offset[i] = h.found ? h.kmer_offset : INVALID_U64;
        if (out.string_id) out.string_id[i] = h.found ? uint64_t(h.string_id) : INVALID_U64;
        if (out.string_begin) out.string_begin[i] = begin;
        if (out.string_end) out.string_end[i] = end;
        if (out.kmer_orientation) out.kmer_orientation[i] = h.orientation;
        if (out.minimizer_found) out.minimizer_found[i] = h.minimizer_found ? 1 : 0;
    }
}

}  // namespace sshash_amd
