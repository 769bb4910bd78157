// sharded.cpp -- lookup against a dictionary partitioned over several GPUs (BASELINE.json
// configs[4]): route -> exchange -> lookup -> return -> combine as ONE call. The only step of the whole lookup path
// that needs a collective is the exchange; it is handed in as a pair of callbacks, so the same code runs over RCCL
// (sshash_sharded_lookup_rccl: grouped ncclSend/ncclRecv = an all-to-all-v over xGMI), over torch.distributed
// (sshash_amd/sharded.py wraps all_to_all_single) or over anything a host application already has.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is resolved at the first RCCL call (no link-time dependency)

#include <dlfcn.h>

#include <mutex>
#include <numeric>
#include <vector>

#include "engine.hpp"
#include "replica.hpp"

namespace sshash_amd {

namespace {

struct stream_buffers {  // stream-ordered scratch out of the replica's own pool, released on every exit path
    hipStream_t s;
    device_replica const* rep;
    std::vector<void*> owned;
    template <typename T>
    T* get(uint64_t n) {
        void* p = rep->stream_alloc(std::max<uint64_t>(n, 1) * sizeof(T), s);
        owned.push_back(p);
        return static_cast<T*>(p);
    }
    ~stream_buffers() {
        for (void* p : owned) (void)hipFreeAsync(p, s);
    }
};

void call(int rc, char const* what) {
    if (rc != 0) throw error(error_kind::internal, std::string("sharded lookup: the ") + what + " exchange failed (" + std::to_string(rc) + ")");
}

}  // namespace

void engine::sharded_lookup_device(int device, uint32_t num_ranks, bool by_table_key, uint64_t const* d_kmers, uint64_t n,
                                   bool check_rc, uint64_t* d_out, exchange_ops const& x, void* stream) const {
    device_replica const* rep = replica(device);
    if (num_ranks == 0 || !x.counts || !x.data) throw error(error_kind::argument, "sharded lookup: ranks and both exchange callbacks are required");
    if (n && (!d_kmers || !d_out)) throw error(error_kind::argument, "null argument");
    device_guard guard(device);
    hipStream_t s = hipStream_t(stream);
    const uint32_t W = rep->view.k <= 31 ? 1 : 2, R = num_ranks;
    stream_buffers buf{s, rep, {}};

    /* 1. route: messages per owner, then the messages themselves in per-owner regions (engine.hip: route_bucket_kernel) */
    uint64_t* d_cursors = buf.get<uint64_t>(R);
    HIP_CHECK(hipMemsetAsync(d_cursors, 0, R * sizeof(uint64_t), s));
    uint32_t* d_owners = buf.get<uint32_t>(n);  // elected once, by the counting launch
    route_bucket_device(device, d_kmers, n, R, check_rc, by_table_key, d_cursors, nullptr, nullptr, s, d_owners);
    std::vector<uint64_t> send_counts(R), recv_counts(R), first(R);
    HIP_CHECK(hipMemcpyAsync(send_counts.data(), d_cursors, R * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    std::exclusive_scan(send_counts.begin(), send_counts.end(), first.begin(), uint64_t(0));
    const uint64_t total = first.back() + send_counts.back();
    HIP_CHECK(hipMemcpyAsync(d_cursors, first.data(), R * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    uint64_t* d_send = buf.get<uint64_t>(total * W);
    uint32_t* d_slots = buf.get<uint32_t>(total);
    route_bucket_device(device, d_kmers, n, R, check_rc, by_table_key, d_cursors, d_send, d_slots, s, d_owners);

    /* 2. exchange: one packed k-mer per message. Every rank must elect table keys of the same length: the length steers which rank owns
       a key (route_bucket_kernel BY_KEY, sk_owner) and is read from the environment when a replica is built (SSHASH_AMD_SK_M:
       sktable.hip), so ranks started under different environments would route keys to ranks that do not hold them -- complete-path
       fallbacks at best, misses on table shards -- with nothing to show for it. The length travels in the top byte of
       every count of THIS exchange, on every call: round 5 compared it in an exchange of its own that a rank ran or skipped by its
       local state, so a rank with a fresh replica paired its key-length exchange with its peers' count exchange. */
    constexpr uint32_t KEY_SHIFT = 56;
    const uint64_t my_key = by_table_key ? uint64_t(rep->view.sk.m) & 0xFFu : 0;
    {
        std::vector<uint64_t> tagged(send_counts);
        for (uint64_t& c : tagged) {
            if (c >> KEY_SHIFT) throw error(error_kind::argument, "sharded lookup: more than 2^56 messages for one rank");
            c |= my_key << KEY_SHIFT;
        }
        call(x.counts(x.ctx, tagged.data(), recv_counts.data()), "count");
    }
    for (uint32_t r = 0; r < R; ++r) {
        const uint64_t their_key = recv_counts[r] >> KEY_SHIFT;
        recv_counts[r] &= (uint64_t(1) << KEY_SHIFT) - 1;
        if (their_key != my_key)
            throw error(error_kind::argument, "sharded lookup: rank " + std::to_string(r) + " elects table keys of " + std::to_string(their_key) +
                                                  " bases, this rank of " + std::to_string(my_key) + " (SSHASH_AMD_SK_M must be the same on every rank, and every rank must shard the same way)");
    }
    const uint64_t m = std::accumulate(recv_counts.begin(), recv_counts.end(), uint64_t(0));
    uint64_t* d_recv = buf.get<uint64_t>(m * W);
    call(x.data(x.ctx, d_send, send_counts.data(), d_recv, recv_counts.data(), W * 8, stream), "k-mer");

    /* 3. the ordinary batched lookup on what arrived (a probe whose structures live elsewhere simply misses) */
    uint64_t* d_ids = buf.get<uint64_t>(m);
    result_view ids{};
    ids.kmer_id = d_ids;
    if (m) lookup_packed_device(device, d_recv, m, check_rc, out_mode::ids, ids, nullptr, stream);

    /* 4. the ids travel back, aligned with the messages; 5. a reply that found its k-mer settles its query */
    uint64_t* d_replies = buf.get<uint64_t>(total);
    call(x.data(x.ctx, d_ids, recv_counts.data(), d_replies, send_counts.data(), 8, stream), "id");
    const bool one_reply_per_query = total == n;  // table keys, canonical minimizers, no reverse complements: every query has ONE owner
    if (n && !one_reply_per_query) HIP_CHECK(hipMemsetAsync(d_out, 0xFF, n * sizeof(uint64_t), s));
    route_combine_device(device, d_replies, d_slots, total, d_out, s, one_reply_per_query);
}

/* ---- the exchange over RCCL ------------------------------------------------------------------------------------ */

namespace {

struct rccl_api {
    ncclResult_t (*group_start)() = nullptr;
    ncclResult_t (*group_end)() = nullptr;
    ncclResult_t (*send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*count)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*user_rank)(const ncclComm_t, int*) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
};

rccl_api const& rccl() {
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        /* the process may already hold an RCCL (torch ships one): use whatever is loaded, else the ROCm one */
        void* lib = nullptr;
        if (!dlsym(RTLD_DEFAULT, "ncclSend")) {
            lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) throw error(error_kind::no_device, "RCCL is not available (librccl.so could not be loaded)");
        }
        auto sym = [&](char const* name) {
            void* p = lib ? dlsym(lib, name) : dlsym(RTLD_DEFAULT, name);
            if (!p) throw error(error_kind::no_device, std::string("RCCL symbol missing: ") + name);
            return p;
        };
        api.group_start = reinterpret_cast<decltype(api.group_start)>(sym("ncclGroupStart"));
        api.group_end = reinterpret_cast<decltype(api.group_end)>(sym("ncclGroupEnd"));
        api.send = reinterpret_cast<decltype(api.send)>(sym("ncclSend"));
        api.recv = reinterpret_cast<decltype(api.recv)>(sym("ncclRecv"));
        api.count = reinterpret_cast<decltype(api.count)>(sym("ncclCommCount"));
        api.user_rank = reinterpret_cast<decltype(api.user_rank)>(sym("ncclCommUserRank"));
        api.error_string = reinterpret_cast<decltype(api.error_string)>(sym("ncclGetErrorString"));
    });
    return api;
}

struct rccl_ctx {
    ncclComm_t comm;
    uint32_t ranks;
    uint32_t self;  // this rank: its own share never leaves the device
    hipStream_t stream;
    uint64_t* d_scratch;  // 2 * ranks uint64 for the counts
};

int nccl_ok(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + int(r); }

/* all-to-all-v as grouped sends and receives: one message per peer in each direction, point to point over xGMI */
int rccl_data(void* c, const void* send, const uint64_t* send_counts, void* recv, const uint64_t* recv_counts, uint32_t elem_bytes,
              void* stream) {
    auto* ctx = static_cast<rccl_ctx*>(c);
    rccl_api const& api = rccl();
    int rc = nccl_ok(api.group_start());
    uint64_t so = 0, ro = 0;
    for (uint32_t p = 0; p < ctx->ranks && rc == 0; ++p) {
        if (p == ctx->self) {  // 1/R of the messages: a copy on the stream (RCCL's send-to-self took 0.71 ms for the 0.8 GB of 10^8 messages, the copy takes 0.3)
            if (send_counts[p] != recv_counts[p]) rc = 4;
            else if (send_counts[p] && hipMemcpyAsync(static_cast<char*>(recv) + ro * elem_bytes, static_cast<const char*>(send) + so * elem_bytes,
                                                      send_counts[p] * elem_bytes, hipMemcpyDeviceToDevice, hipStream_t(stream)) != hipSuccess)
                rc = 5;
            so += send_counts[p];
            ro += recv_counts[p];
            continue;
        }
        if (send_counts[p]) rc = nccl_ok(api.send(static_cast<const char*>(send) + so * elem_bytes, send_counts[p] * elem_bytes, ncclUint8, int(p), ctx->comm, hipStream_t(stream)));
        if (rc == 0 && recv_counts[p]) rc = nccl_ok(api.recv(static_cast<char*>(recv) + ro * elem_bytes, recv_counts[p] * elem_bytes, ncclUint8, int(p), ctx->comm, hipStream_t(stream)));
        so += send_counts[p];
        ro += recv_counts[p];
    }
    const int end = nccl_ok(api.group_end());
    return rc ? rc : end;
}

int rccl_counts(void* c, const uint64_t* send, uint64_t* recv) {
    auto* ctx = static_cast<rccl_ctx*>(c);
    const uint32_t R = ctx->ranks;
    if (hipMemcpyAsync(ctx->d_scratch, send, R * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return 1;
    std::vector<uint64_t> ones(R, 1);
    const int rc = rccl_data(c, ctx->d_scratch, ones.data(), ctx->d_scratch + R, ones.data(), 8, ctx->stream);
    if (rc) return rc;
    if (hipMemcpyAsync(recv, ctx->d_scratch + R, R * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return 2;
    return hipStreamSynchronize(ctx->stream) == hipSuccess ? 0 : 3;
}

}  // namespace

void engine::sharded_lookup_rccl(int device, void* nccl_comm, bool by_table_key, uint64_t const* d_kmers, uint64_t n, bool check_rc,
                                 uint64_t* d_out, void* stream) const {
    if (!nccl_comm) throw error(error_kind::argument, "null RCCL communicator");
    (void)replica(device);
    device_guard guard(device);
    int ranks = 0;
    if (rccl().count(static_cast<ncclComm_t>(nccl_comm), &ranks) != ncclSuccess || ranks < 1)
        throw error(error_kind::argument, "ncclCommCount failed on the given communicator");
    int self = 0;
    if (rccl().user_rank(static_cast<ncclComm_t>(nccl_comm), &self) != ncclSuccess || self < 0 || self >= ranks)
        throw error(error_kind::argument, "ncclCommUserRank failed on the given communicator");
    rccl_ctx ctx{static_cast<ncclComm_t>(nccl_comm), uint32_t(ranks), uint32_t(self), hipStream_t(stream), nullptr};
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx.d_scratch), 2 * uint64_t(ranks) * 8));
    struct release {
        uint64_t* p;
        ~release() { (void)hipFree(p); }
    } scratch{ctx.d_scratch};
    exchange_ops x{&ctx, rccl_counts, rccl_data};
    sharded_lookup_device(device, uint32_t(ranks), by_table_key, d_kmers, n, check_rc, d_out, x, stream);
    HIP_CHECK(hipStreamSynchronize(hipStream_t(stream)));  // ctx and scratch go out of scope
}

}  // namespace sshash_amd
