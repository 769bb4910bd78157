// replica.hpp -- one dictionary replica resident in the HBM of one device (internal header).
#pragma once

#include <atomic>

#include <cstdio>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "lookup_device.hpp"
#include "mphf_build.hpp"

namespace sshash_amd {

inline void hip_check(hipError_t e, char const* what) {
    if (e != hipSuccess) throw error(error_kind::hip, std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}
#define HIP_CHECK(x) ::sshash_amd::hip_check((x), #x)

struct device_guard {
    int prev = 0;
    int target;
    explicit device_guard(int dev) : target(dev) {
        HIP_CHECK(hipGetDevice(&prev));
        if (prev != dev) HIP_CHECK(hipSetDevice(dev));
    }
    ~device_guard() {
        if (prev != target) (void)hipSetDevice(prev);
    }
};

/* one pipeline of the host-buffer entry points: stream + pinned staging block + device block (engine.hip) */
struct host_lane {
    hipStream_t stream = nullptr;
    void* pinned = nullptr;
    void* device = nullptr;
    size_t capacity = 0;
};

struct device_replica {
    int device = -1;
    uint64_t bytes = 0;
    uint64_t bytes_still_to_come = 0;  // during the upload: what follows the super-k-mer table (its budget test counts it in)
    dict_view view{};
    skew_part_dev* d_skew = nullptr;
    uint64_t directory_overflowed = 0;  // sectors carrying the overflow flag
    uint64_t directory_entries = 0;     // keys resident in the directory
    uint64_t sk_keys = 0, sk_heavy_keys = 0, sk_heavy_kmers = 0, sk_unplaced = 0, sk_slots_used = 0, sk_bytes = 0;  // super-k-mer table
    uint32_t sk_absent_reason = 1;  // SK_ABSENT_* (0 = the table is there)
    /* keys of the table by number of occurrences (bins SK_HIST_BINS: 1, 2, 3, 4, 5-8, 9-16, 17-64, 65-1024, > 1024):
       [0..9) keys per bin, [9..18) occurrences (= super-k-mers) per bin; [18] super-k-mers, [19] slots asked for */
    uint64_t sk_histogram[20] = {0};
    std::vector<void*> allocations;
    /* Stream-ordered scratch of the batched calls (streaming lookup, sharded lookup, neighbours) comes out of a memory pool of
       this replica's own, which keeps what it has used (a pool hands everything back to the driver at the next
       synchronisation by default: 13 vs 190-260 ms per 3 x 10^8-base streaming lookup). Round 2 raised the threshold of the
       DEVICE'S DEFAULT pool instead -- a process-wide side effect on the embedding application, whose own hipMallocAsync
       blocks were then never returned either. */
    hipMemPool_t scratch_pool = nullptr;
    void create_scratch_pool() {
        hipMemPoolProps props{};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = device;
        if (hipMemPoolCreate(&scratch_pool, &props) != hipSuccess) {
            scratch_pool = nullptr;  // (fall back to the default pool with its default behaviour)
            (void)hipGetLastError();
            /* not silently: the default pool hands its memory back at every synchronisation -- 190-260 ms instead of 13 per batched
               streaming call (HISTORY.md) */
            if (std::getenv("SSHASH_AMD_VERBOSE"))
                fprintf(stderr, "[sshash_amd] device %d: no private memory pool (hipMemPoolCreate failed): stream-ordered scratch comes out of the default pool\n", device);
            return;
        }
        uint64_t keep = ~uint64_t(0);
        (void)hipMemPoolSetAttribute(scratch_pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    void* stream_alloc(size_t bytes, hipStream_t s) const {
        void* p = nullptr;
        if (scratch_pool) HIP_CHECK(hipMallocFromPoolAsync(&p, bytes ? bytes : 8, scratch_pool, s));
        else HIP_CHECK(hipMallocAsync(&p, bytes ? bytes : 8, s));
        return p;
    }

    /* Per-stream scratch for the resume and deferred queues of the multi-pass lookup (engine.hip). Work on one stream is
       ordered, so a buffer keyed by the stream can be reused without synchronisation; it only grows
       (hipFree of the old block synchronises implicitly). At most SCRATCH_STREAMS_MAX streams keep a block. */
    /* held while one launch sequence (queue reset, first / resume / deferred pass) is enqueued: two host threads that share a
       stream (the null stream, typically) must not interleave their sequences, which share that stream's scratch */
    mutable std::mutex launch_mutex;
    mutable std::mutex scratch_mutex;
    struct stream_scratch {
        void* block = nullptr;
        size_t bytes = 0;
        uint64_t last_use = 0;
        /* the tail passes (resume, deferred) of one launch piece run on `aux` while the caller's stream already runs the
           first pass of the next piece (engine.hip: launch): one auxiliary stream and two event pairs per caller stream */
        hipStream_t aux = nullptr;
        hipEvent_t first_done[2] = {nullptr, nullptr};
        hipEvent_t tail_done[2] = {nullptr, nullptr};
        void release() {
            if (aux) (void)hipStreamSynchronize(aux);
            for (int i = 0; i < 2; ++i) {
                if (first_done[i]) (void)hipEventDestroy(first_done[i]);
                if (tail_done[i]) (void)hipEventDestroy(tail_done[i]);
                first_done[i] = tail_done[i] = nullptr;
            }
            if (aux) (void)hipStreamDestroy(aux);
            aux = nullptr;
            if (block) (void)hipFree(block);
            block = nullptr;
            bytes = 0;
        }
    };
    mutable std::unordered_map<void*, stream_scratch> scratch;
    mutable uint64_t scratch_clock = 0;
    static constexpr size_t SCRATCH_STREAMS_MAX = 16;  // an application that makes a stream per request must not keep a
                                                       // gigabyte of queues per stream it ever used: the
                                                       // least recently used stream's block goes first
    stream_scratch& scratch_for(void* stream, size_t bytes, bool with_aux_stream) const {
        std::lock_guard<std::mutex> lock(scratch_mutex);
        auto it = scratch.find(stream);
        if (it == scratch.end()) {
            if (scratch.size() >= SCRATCH_STREAMS_MAX) {
                auto oldest = scratch.begin();
                for (auto jt = scratch.begin(); jt != scratch.end(); ++jt)
                    if (jt->second.last_use < oldest->second.last_use) oldest = jt;
                oldest->second.release();  // (synchronises: nothing in flight still uses it)
                scratch.erase(oldest);
            }
            it = scratch.emplace(stream, stream_scratch{}).first;
        }
        stream_scratch& slot = it->second;
        slot.last_use = ++scratch_clock;
        if (slot.bytes < bytes) {
            if (slot.aux) HIP_CHECK(hipStreamSynchronize(slot.aux));  // a tail pass of an earlier call may still read the old block
            if (slot.block) HIP_CHECK(hipFree(slot.block));
            slot.block = nullptr;
            slot.bytes = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            HIP_CHECK(hipMalloc(&slot.block, want));
            slot.bytes = want;
        }
        if (with_aux_stream && !slot.aux) {
            HIP_CHECK(hipStreamCreateWithFlags(&slot.aux, hipStreamNonBlocking));
            for (int i = 0; i < 2; ++i) {
                HIP_CHECK(hipEventCreateWithFlags(&slot.first_done[i], hipEventDisableTiming));
                HIP_CHECK(hipEventCreateWithFlags(&slot.tail_done[i], hipEventDisableTiming));
            }
        }
        return slot;
    }

    /* Per-stream scratch of the streaming query (streaming.hip: the 2-bit packed copy of a call's reads and their validity bits).
       Round 5 took it from the pool above and gave it back at every call -- two stream-ordered allocations of ~0.4 GB each per call
       of 3 x 10^9 bases, on the host side of a step that the device then waits for (profiles/r06/). Kept per stream instead: calls
       on one stream are ordered, so the block is reused without synchronisation and only grows (hipFree of the old block
       synchronises). An application that makes a stream per request does not keep a block per stream it ever used: the least
       recently used ones go once more than READ_SCRATCH_STREAMS_MAX streams or READ_SCRATCH_BYTES_MAX bytes are held. */
    struct read_scratch {
        void* block = nullptr;
        size_t bytes = 0;
        uint64_t last_use = 0;
    };
    mutable std::unordered_map<void*, read_scratch> read_scratches;
    static constexpr size_t READ_SCRATCH_STREAMS_MAX = 64;  // (the file query runs up to 32 lanes, each on its own stream)
    static constexpr size_t READ_SCRATCH_BYTES_MAX = size_t(8) << 30;
    void* read_scratch_for(void* stream, size_t bytes) const {
        std::lock_guard<std::mutex> lock(scratch_mutex);
        auto it = read_scratches.find(stream);
        if (it == read_scratches.end()) it = read_scratches.emplace(stream, read_scratch{}).first;
        read_scratch& slot = it->second;
        slot.last_use = ++scratch_clock;
        if (slot.bytes < bytes) {
            if (slot.block) HIP_CHECK(hipFree(slot.block));  // (synchronises: nothing in flight still reads it)
            slot.block = nullptr;
            slot.bytes = 0;
            const size_t want = bytes + bytes / 8 + 4096;
            size_t held = want;
            for (auto const& kv : read_scratches) held += kv.second.bytes;
            while (read_scratches.size() > 1 && (read_scratches.size() > READ_SCRATCH_STREAMS_MAX || held > READ_SCRATCH_BYTES_MAX)) {
                auto oldest = read_scratches.end();
                for (auto jt = read_scratches.begin(); jt != read_scratches.end(); ++jt)
                    if (jt != it && (oldest == read_scratches.end() || jt->second.last_use < oldest->second.last_use)) oldest = jt;
                if (oldest == read_scratches.end()) break;
                if (oldest->second.block) (void)hipFree(oldest->second.block);
                held -= oldest->second.bytes;
                read_scratches.erase(oldest);
            }
            HIP_CHECK(hipMalloc(&slot.block, want));
            slot.bytes = want;
        }
        return slot.block;
    }

    /* pooled lanes of the host-buffer path */
    mutable std::mutex lanes_mutex;
    mutable std::vector<host_lane*> idle_lanes;
    host_lane* acquire_lane(size_t bytes) const;
    void release_lane(host_lane* lane) const;

    template <typename T>
    T* put(std::vector<T> const& v) {
        T* d = nullptr;
        const size_t n = (v.empty() ? 1 : v.size()) * sizeof(T);
        HIP_CHECK(hipMalloc(&d, n));
        allocations.push_back(d);
        if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        bytes += n;
        return d;
    }
    mphf_view put_mphf(mphf_host const& f) {
        mphf_view v = f.view();
        v.parts = put(f.parts);
        v.pilots = put(f.pilots);
        v.free_slots = put(f.free_slots);
        return v;
    }
    device_replica() = default;
    device_replica(device_replica const&) = delete;
    device_replica& operator=(device_replica const&) = delete;
    ~device_replica() {
        if (device < 0) return;
        int prev = 0;
        if (hipGetDevice(&prev) != hipSuccess) return;
        (void)hipSetDevice(device);
        for (void* p : allocations) (void)hipFree(p);
        if (scratch_pool) {
            (void)hipDeviceSynchronize();  // (stream-ordered frees of this pool must have run)
            (void)hipMemPoolDestroy(scratch_pool);
        }
        for (auto& kv : scratch) kv.second.release();
        for (auto& kv : read_scratches)
            if (kv.second.block) (void)hipFree(kv.second.block);
        for (host_lane* lane : idle_lanes) {
            if (lane->pinned) (void)hipHostFree(lane->pinned);
            if (lane->device) (void)hipFree(lane->device);
            if (lane->stream) (void)hipStreamDestroy(lane->stream);
            delete lane;
        }
        (void)hipSetDevice(prev);
    }
};

}  // namespace sshash_amd
