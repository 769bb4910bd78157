// streaming.hip -- batched streaming query: one read per lane (gfx950 only).
//
// Reference: streaming_query<Dict,canonical>::lookup / seed (include/streaming_query.hpp:56-109,
// 144-197) driven per read by src/query.cpp:78-108. The state machine is sequential inside a
// read and independent across reads, so reads are the parallel dimension. Per lane:
//   * k-mer and its reverse complement are rolled one base at a time (:68-80);
//   * while the previous k-mer matched at string offset `off`, the next k-mer is first compared
//     with the string's own next k-mer ("extension", :86-100). With the granule layout the
//     reference's `remaining_string_bases` counter is implicit: the window at off+-1 reports
//     whether it runs across a string boundary, which is exactly remaining == 0;
//   * otherwise seed(): the negative short-cut (:150-157), then the point lookups (:159-180).
//     Minimizers are only needed inside seed(), so they are computed there, statelessly
//     (the rolling iterators of include/minimizer_iterator.hpp return the same values: :56-57).
//     When the replica has a super-k-mer table seed() goes through it (device_layout.hpp (5)): one slot read per
//     seed instead of directory + window per strand. The negative short-cut becomes "same table key as
//     the previous k-mer, and that key is provably not in the table" -- the same k-mers are negative
//     either way, so the counters are unchanged; queries the table defers take the path above.
// Output: the six counters of streaming_query_report (include/util.hpp:21-36).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <thread>

#include "engine.hpp"
#include "replica.hpp"

namespace sshash_amd {

namespace {

template <int W, bool CANON>
__device__ __forceinline__ hit_t seed_lookup(dict_view const& d, skew_part_dev const* __restrict__ skew,
                                             kmer_w<W> const& x, kmer_w<W> const& x_rc, minimizer_t mf, minimizer_t mr) {
    if constexpr (CANON) {  // include/streaming_query.hpp:159-169
        if (mf.value < mr.value) return probe_canonical<W, true>(d, skew, x, x_rc, mf);
        if (mr.value < mf.value) return probe_canonical<W, true>(d, skew, x, x_rc, mr);
        hit_t h = probe_canonical<W, true>(d, skew, x, x_rc, mf);
        if (!h.found) h = probe_canonical<W, true>(d, skew, x, x_rc, mr);
        return h;
    } else {  // :170-180
        hit_t h = probe_regular<W, true>(d, skew, x, mf);
        if (!h.found) {
            const bool mf_found = h.minimizer_found;
            h = probe_regular<W, true>(d, skew, x_rc, mr);
            h.orientation = -1;
            h.minimizer_found = h.minimizer_found || mf_found;
        }
        return h;
    }
}

__device__ __forceinline__ uint64_t wave_sum(uint64_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

template <int W, bool CANON, bool SK>
__global__ void __launch_bounds__(256)
streaming_kernel(const dict_view d, const skew_part_dev* __restrict__ skew, const char* __restrict__ bases,
                 const uint64_t* __restrict__ offsets, const uint64_t n_reads, uint64_t* __restrict__ report) {
    uint64_t c_kmers = 0, c_invalid = 0, c_negative = 0, c_searches = 0, c_extensions = 0;
    const uint32_t k = d.k;
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    for (uint64_t r = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n_reads; r += stride) {
        const uint64_t begin = offsets[r], len = offsets[r + 1] - begin;
        if (len < k) continue;
        c_kmers += len - k + 1;
        kmer_w<W> x = kmer_zero<W>(), x_rc = kmer_zero<W>();
        uint32_t valid_len = 0;
        bool in_run = false;        // previous k-mer was found (remaining bases tracked via `off`)
        bool neg_unknown_mini = false;  // previous k-mer: seed() said "negative, minimizer not in index"
        uint64_t prev_f = 0, prev_r = 0;  // SK: prev_f holds the previous table key
        uint64_t off = 0;
        int ori = 1;
        const char* p = bases + begin;
        for (uint64_t j = 0; j < len; ++j) {
            const char c = p[j];
            const uint64_t code = base_code(c);
            x = kmer_roll<W>(x, code, k);
            x_rc = kmer_roll_rc<W>(x_rc, code, k);
            valid_len = base_is_valid(c) ? valid_len + 1 : 0;
            if (j + 1 < k) continue;
            if (valid_len < k) {  // :59-65 -- invalid k-mer resets the whole state
                ++c_invalid;
                in_run = false;
                neg_unknown_mini = false;
                continue;
            }
            if (in_run && !(ori < 0 && off == 0)) {  // :86-100
                const uint64_t next = ori > 0 ? off + 1 : off - 1;
                const window_t<W> w = read_window<W>(d.granules, next, k);
                if (!w.crosses && (kmer_eq<W>(w.kmer, x) || kmer_eq<W>(w.kmer, x_rc))) {
                    ++c_extensions;
                    off = next;
                    continue;
                }
            }
            /* seed() */
            if constexpr (SK) {
                const sk_key_t kk = sk_key<W>(x, x_rc, k, d.m);
                if (sk_usable(d, kk)) {
                    if (neg_unknown_mini && kk.key == prev_f) {
                        ++c_negative;
                        in_run = false;
                        continue;
                    }
                    bool key_seen;
                    const fast_t r = sk_probe<W>(d, x, x_rc, kk, key_seen);
                    if (r.outcome != FAST_DEFER) {
                        if (r.outcome == FAST_HIT) {
                            ++c_searches;
                            in_run = true;
                            off = r.kmer_offset;
                            ori = r.orientation;
                            neg_unknown_mini = false;
                        } else {
                            ++c_negative;
                            in_run = false;
                            neg_unknown_mini = !key_seen;
                            prev_f = kk.key;
                        }
                        continue;
                    }
                }
                /* tie / unplaced key / over-long list: the complete seed() below (its own short-cut state is
                   not carried across table seeds) */
                neg_unknown_mini = false;
            }
            const minimizer_t mf = compute_minimizer<W>(x, k, d.m, d.hash_magic);
            const minimizer_t mr = compute_minimizer<W>(x_rc, k, d.m, d.hash_magic);
            if (neg_unknown_mini && mf.value == prev_f && mr.value == prev_r) {  // :150-157
                ++c_negative;
                in_run = false;
                continue;
            }
            prev_f = mf.value;
            prev_r = mr.value;
            const hit_t h = seed_lookup<W, CANON>(d, skew, x, x_rc, mf, mr);
            if (h.found) {
                ++c_searches;
                in_run = true;
                off = h.kmer_offset;
                ori = h.orientation;
                neg_unknown_mini = false;
            } else {
                ++c_negative;
                in_run = false;
                neg_unknown_mini = SK ? false : !h.minimizer_found;
            }
        }
    }
    c_kmers = wave_sum(c_kmers);
    c_invalid = wave_sum(c_invalid);
    c_negative = wave_sum(c_negative);
    c_searches = wave_sum(c_searches);
    c_extensions = wave_sum(c_extensions);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(report + 0), (unsigned long long)c_kmers);
        atomicAdd(reinterpret_cast<unsigned long long*>(report + 1), (unsigned long long)(c_searches + c_extensions));
        atomicAdd(reinterpret_cast<unsigned long long*>(report + 2), (unsigned long long)c_negative);
        atomicAdd(reinterpret_cast<unsigned long long*>(report + 3), (unsigned long long)c_invalid);
        atomicAdd(reinterpret_cast<unsigned long long*>(report + 4), (unsigned long long)c_searches);
        atomicAdd(reinterpret_cast<unsigned long long*>(report + 5), (unsigned long long)c_extensions);
    }
}

template <int W, bool CANON>
void launch_streaming(dict_view const& d, skew_part_dev const* skew, char const* bases, uint64_t const* offsets,
                      uint64_t n_reads, uint64_t* report, hipStream_t s) {
    const uint32_t block = 256;
    uint64_t blocks = (n_reads + block - 1) / block;
    if (blocks > (uint64_t(1) << 20)) blocks = uint64_t(1) << 20;
    if (d.sk.enabled)
        hipLaunchKernelGGL((streaming_kernel<W, CANON, true>), dim3(uint32_t(blocks)), dim3(block), 0, s, d, skew, bases,
                           offsets, n_reads, report);
    else
        hipLaunchKernelGGL((streaming_kernel<W, CANON, false>), dim3(uint32_t(blocks)), dim3(block), 0, s, d, skew, bases,
                           offsets, n_reads, report);
    HIP_CHECK(hipGetLastError());
}

}  // namespace

void engine::streaming_query_device(int device, char const* d_bases, uint64_t const* d_read_offsets, uint64_t n_reads,
                                    uint64_t /*total_bases*/, uint64_t* d_report, void* stream) const {
    device_replica const* rep = replica(device);
    if (n_reads == 0) return;
    int prev = 0;
    HIP_CHECK(hipGetDevice(&prev));
    if (prev != device) HIP_CHECK(hipSetDevice(device));
    dict_view const& d = rep->view;
    hipStream_t s = hipStream_t(stream);
    const bool wide = d.k > 31;
    if (!wide && !d.canonical) launch_streaming<1, false>(d, rep->d_skew, d_bases, d_read_offsets, n_reads, d_report, s);
    else if (!wide && d.canonical) launch_streaming<1, true>(d, rep->d_skew, d_bases, d_read_offsets, n_reads, d_report, s);
    else if (wide && !d.canonical) launch_streaming<2, false>(d, rep->d_skew, d_bases, d_read_offsets, n_reads, d_report, s);
    else launch_streaming<2, true>(d, rep->d_skew, d_bases, d_read_offsets, n_reads, d_report, s);
    if (prev != device) HIP_CHECK(hipSetDevice(prev));
}

/* Host buffers: the reads are cut into pieces of at most ~32 MiB of bases; per replica up to eight lanes (the
   pooled pinned pipelines of the lookup host path, replica.hpp) pull pieces from a shared counter and run
   copy-in -> H2D -> kernel, accumulating the six counters in device memory; one read-back per lane. */
streaming_report engine::streaming_query_host(char const* bases, uint64_t const* read_offsets, uint64_t n_reads) const {
    streaming_report total;
    if (n_reads == 0) return total;
    const std::vector<int> devs = devices();
    if (devs.empty()) throw error(error_kind::no_device, "dictionary is not resident on any device (call sshash_to_device first)");
    const uint64_t G = devs.size();
    /* pieces: [first read, last read) with a bounded number of bases (a single longer read is its own piece) */
    const uint64_t piece_bases = uint64_t(32) << 20, piece_reads = uint64_t(1) << 20;
    std::vector<uint64_t> cuts{0};
    uint64_t max_bases = 0, max_reads = 0;
    for (uint64_t at = 0; at < n_reads;) {
        uint64_t end = at + 1;
        while (end < n_reads && end - at < piece_reads && read_offsets[end + 1] - read_offsets[at] <= piece_bases) ++end;
        max_bases = std::max(max_bases, read_offsets[end] - read_offsets[at]);
        max_reads = std::max(max_reads, end - at);
        cuts.push_back(end);
        at = end;
    }
    const uint64_t num_pieces = cuts.size() - 1;
    const uint64_t off_bytes = (max_reads + 1) * sizeof(uint64_t);
    const uint64_t bases_at = (off_bytes + 255) & ~uint64_t(255);
    const uint64_t report_at = (bases_at + max_bases + 255) & ~uint64_t(255);
    const uint64_t lane_bytes = report_at + 6 * sizeof(uint64_t);

    std::atomic<uint64_t> next{0};
    const uint64_t hw = std::max(1u, std::thread::hardware_concurrency());
    const uint64_t lanes_per_device = std::min<uint64_t>({(num_pieces + G - 1) / G, 8, std::max<uint64_t>(1, hw / G)});
    const uint64_t num_lanes = lanes_per_device * G;
    std::vector<std::exception_ptr> errors(num_lanes);
    std::vector<streaming_report> partial(num_lanes);

    auto run_lane = [&](uint64_t li) {
        try {
            const int device = devs[li % G];
            device_replica const* rep = replica(device);
            HIP_CHECK(hipSetDevice(device));
            host_lane* lane = rep->acquire_lane(lane_bytes);
            struct give_back {
                device_replica const* rep;
                host_lane* lane;
                ~give_back() { rep->release_lane(lane); }
            } guard{rep, lane};
            hipStream_t s = lane->stream;
            char* hp = static_cast<char*>(lane->pinned);
            char* dp = static_cast<char*>(lane->device);
            uint64_t* d_report = reinterpret_cast<uint64_t*>(dp + report_at);
            HIP_CHECK(hipMemsetAsync(d_report, 0, 6 * sizeof(uint64_t), s));
            for (;;) {
                const uint64_t piece = next.fetch_add(1);
                if (piece >= num_pieces) break;
                const uint64_t first = cuts[piece], last = cuts[piece + 1];
                const uint64_t nb = read_offsets[last] - read_offsets[first];
                uint64_t* rel = reinterpret_cast<uint64_t*>(hp);
                for (uint64_t i = first; i <= last; ++i) rel[i - first] = read_offsets[i] - read_offsets[first];
                std::memcpy(hp + bases_at, bases + read_offsets[first], nb);
                HIP_CHECK(hipMemcpyAsync(dp, hp, (last - first + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
                HIP_CHECK(hipMemcpyAsync(dp + bases_at, hp + bases_at, nb, hipMemcpyHostToDevice, s));
                streaming_query_device(device, dp + bases_at, reinterpret_cast<uint64_t const*>(dp), last - first, nb, d_report, s);
                HIP_CHECK(hipStreamSynchronize(s));  // the pinned block is reused by the next piece
            }
            uint64_t h[6];
            HIP_CHECK(hipMemcpyAsync(h, d_report, sizeof(h), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            partial[li].num_kmers = h[0];
            partial[li].num_positive_kmers = h[1];
            partial[li].num_negative_kmers = h[2];
            partial[li].num_invalid_kmers = h[3];
            partial[li].num_searches = h[4];
            partial[li].num_extensions = h[5];
        } catch (...) { errors[li] = std::current_exception(); }
    };

    int prev = 0;
    HIP_CHECK(hipGetDevice(&prev));
    if (num_lanes == 1) {
        run_lane(0);
    } else {
        std::vector<std::thread> workers;
        for (uint64_t li = 0; li < num_lanes; ++li) workers.emplace_back(run_lane, li);
        for (auto& w : workers) w.join();
    }
    (void)hipSetDevice(prev);
    for (auto const& e : errors)
        if (e) std::rethrow_exception(e);
    for (auto const& p : partial) {
        total.num_kmers += p.num_kmers;
        total.num_positive_kmers += p.num_positive_kmers;
        total.num_negative_kmers += p.num_negative_kmers;
        total.num_invalid_kmers += p.num_invalid_kmers;
        total.num_searches += p.num_searches;
        total.num_extensions += p.num_extensions;
    }
    return total;
}

}  // namespace sshash_amd
