// check_sharded.cpp -- the sharded lookup (BASELINE.json configs[4]) driven from C++ through the C
// ABI alone, the way a host application of the reference (tools/query.cpp-shaped) would:
//   [A] two ranks = two host threads, each holding one MINIMIZER shard (sshash_build_config.num_shards = 2) on device 0,
//       exchanging through callbacks of their own (sshash_exchange): ids must equal the whole dictionary's;
//   [B] the same with TABLE shards (sshash_to_device_table_shard) of the complete dictionary;
//   [C] sshash_sharded_lookup_rccl over a real RCCL communicator (one rank: the only size one GPU allows).
// Usage: check_sharded <input.fa[.gz]> <k> <m>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sshash_amd.h"

#define REQUIRE(cond, msg)                                             \
    do {                                                               \
        if (!(cond)) {                                                 \
            printf("ERROR: %s (%s)\n", msg, sshash_last_error());     \
            exit(1);                                                   \
        }                                                              \
    } while (0)
#define HIP(x) REQUIRE((x) == hipSuccess, #x)

struct hub {  // a two-party meeting point
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0, generation = 0;
    const uint64_t* counts_send[2];
    const void* data_send[2];
    const uint64_t* data_send_counts[2];
    void meet() {
        std::unique_lock<std::mutex> lock(m);
        const int gen = generation;
        if (++arrived == 2) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lock, [&] { return generation != gen; });
        }
    }
};
struct party {
    hub* h;
    int rank;
};

static int exchange_counts(void* ctx, const uint64_t* send, uint64_t* recv) {
    party* p = static_cast<party*>(ctx);
    p->h->counts_send[p->rank] = send;
    p->h->meet();
    for (int peer = 0; peer < 2; ++peer) recv[peer] = p->h->counts_send[peer][p->rank];
    p->h->meet();
    return 0;
}

static int exchange_data(void* ctx, const void* send, const uint64_t* send_counts, void* recv, const uint64_t* recv_counts,
                         uint32_t elem_bytes, void* stream) {
    party* p = static_cast<party*>(ctx);
    if (hipStreamSynchronize(hipStream_t(stream)) != hipSuccess) return 1;  // my send buffer is complete
    p->h->data_send[p->rank] = send;
    p->h->data_send_counts[p->rank] = send_counts;
    p->h->meet();
    uint64_t at = 0;
    for (int peer = 0; peer < 2; ++peer) {
        uint64_t before = 0;  // where my block starts inside the peer's send buffer
        for (int r = 0; r < p->rank; ++r) before += p->h->data_send_counts[peer][r];
        if (p->h->data_send_counts[peer][p->rank] != recv_counts[peer]) return 2;
        if (recv_counts[peer] &&
            hipMemcpyAsync(static_cast<char*>(recv) + at * elem_bytes, static_cast<const char*>(p->h->data_send[peer]) + before * elem_bytes,
                           recv_counts[peer] * elem_bytes, hipMemcpyDeviceToDevice, hipStream_t(stream)) != hipSuccess)
            return 3;
        at += recv_counts[peer];
    }
    if (hipStreamSynchronize(hipStream_t(stream)) != hipSuccess) return 4;
    p->h->meet();  // nobody releases a send buffer before everybody has read it
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) return printf("Usage: %s <input.fa[.gz]> <k> <m>\n", argv[0]), 2;
    sshash_build_config cfg;
    sshash_build_config_default(&cfg);
    cfg.k = uint32_t(atoi(argv[2]));
    cfg.m = uint32_t(atoi(argv[3]));
    cfg.num_threads = 8;
    sshash_dict* whole = nullptr;
    REQUIRE(sshash_build_from_fasta(argv[1], &cfg, &whole) == SSHASH_OK, "build");
    REQUIRE(sshash_to_device(whole, 0) == SSHASH_OK, "to_device");
    sshash_info info;
    REQUIRE(sshash_get_info(whole, &info) == SSHASH_OK, "info");
    const uint64_t W = info.words_per_kmer, n = 60000;

    /* two local batches: positives (every other one left as it is, the rest with scrambled low bits = negatives) */
    std::vector<std::vector<uint64_t>> batch(2), expected(2);
    srand(7);
    for (int r = 0; r < 2; ++r) {
        std::vector<uint64_t> ids(n);
        for (auto& id : ids) id = (uint64_t(rand()) * 2147483647ull + uint64_t(rand())) % info.num_kmers;
        batch[r].resize(n * W);
        REQUIRE(sshash_access_packed(whole, ids.data(), n, batch[r].data()) == SSHASH_OK, "access");
        for (uint64_t i = 1; i < n; i += 2) batch[r][i * W] ^= 0x5DEECE66Dull & ((uint64_t(1) << 40) - 1);
        expected[r].resize(n);
        sshash_results out{};
        out.kmer_id = expected[r].data();
        REQUIRE(sshash_lookup_packed(whole, batch[r].data(), n, 1, &out) == SSHASH_OK, "lookup");
    }

    for (int by_table = 0; by_table < 2; ++by_table) {
        printf("checking the sharded lookup over two %s shards...\n", by_table ? "table" : "minimizer");
        sshash_dict* shard[2] = {nullptr, nullptr};
        for (uint32_t r = 0; r < 2; ++r) {
            sshash_build_config c = cfg;
            if (!by_table) {
                c.num_shards = 2;
                c.shard_id = r;
            }
            REQUIRE(sshash_build_from_fasta(argv[1], &c, &shard[r]) == SSHASH_OK, "build shard");
            if (by_table) REQUIRE(sshash_to_device_table_shard(shard[r], 0, 2, r) == SSHASH_OK, "to_device_table_shard");
            else REQUIRE(sshash_to_device(shard[r], 0) == SSHASH_OK, "to_device shard");
        }
        hub h;
        std::vector<std::vector<uint64_t>> got(2, std::vector<uint64_t>(n));
        int failed[2] = {0, 0};
        auto run = [&](int r) {
            if (hipSetDevice(0) != hipSuccess) { failed[r] = 1; return; }
            hipStream_t s;
            uint64_t *d_q = nullptr, *d_ids = nullptr;
            if (hipStreamCreate(&s) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&d_q), n * W * 8) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&d_ids), n * 8) != hipSuccess ||
                hipMemcpy(d_q, batch[r].data(), n * W * 8, hipMemcpyHostToDevice) != hipSuccess) { failed[r] = 2; return; }
            party me{&h, r};
            sshash_exchange x{&me, exchange_counts, exchange_data};
            if (sshash_sharded_lookup_device(shard[r], 0, 2, by_table, d_q, n, 1, d_ids, &x, s) != SSHASH_OK) { failed[r] = 3; return; }
            if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(got[r].data(), d_ids, n * 8, hipMemcpyDeviceToHost) != hipSuccess) failed[r] = 4;
            (void)hipFree(d_q);
            (void)hipFree(d_ids);
            (void)hipStreamDestroy(s);
        };
        std::thread t0(run, 0), t1(run, 1);
        t0.join();
        t1.join();
        for (int r = 0; r < 2; ++r) {
            REQUIRE(failed[r] == 0, "a rank failed");
            uint64_t found = 0;
            for (uint64_t i = 0; i < n; ++i) {
                REQUIRE(got[r][i] == expected[r][i], "sharded id differs from the whole dictionary's");
                found += got[r][i] != UINT64_MAX;
            }
            REQUIRE(found >= n / 2 && found < n, "the batch was meant to mix positives and negatives");
        }
        sshash_free(shard[0]);
        sshash_free(shard[1]);
    }

    printf("checking the sharded lookup over an RCCL communicator...\n");
    {
        ncclUniqueId id;
        ncclComm_t comm;
        REQUIRE(ncclGetUniqueId(&id) == ncclSuccess, "ncclGetUniqueId");
        REQUIRE(ncclCommInitRank(&comm, 1, id, 0) == ncclSuccess, "ncclCommInitRank");
        hipStream_t s;
        uint64_t *d_q = nullptr, *d_ids = nullptr;
        HIP(hipStreamCreate(&s));
        HIP(hipMalloc(reinterpret_cast<void**>(&d_q), n * W * 8));
        HIP(hipMalloc(reinterpret_cast<void**>(&d_ids), n * 8));
        HIP(hipMemcpy(d_q, batch[0].data(), n * W * 8, hipMemcpyHostToDevice));
        std::vector<uint64_t> got(n);
        for (int by_table = 0; by_table < 2; ++by_table) {  // one rank: the whole dictionary is its own only shard
            REQUIRE(sshash_sharded_lookup_rccl(whole, 0, comm, by_table, d_q, n, 1, d_ids, s) == SSHASH_OK, "sshash_sharded_lookup_rccl");
            HIP(hipStreamSynchronize(s));
            HIP(hipMemcpy(got.data(), d_ids, n * 8, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < n; ++i) REQUIRE(got[i] == expected[0][i], "id over RCCL differs");
        }
        REQUIRE(sshash_sharded_lookup_rccl(whole, 0, comm, 0, nullptr, 0, 1, nullptr, s) == SSHASH_OK, "empty local batch");
        ncclCommDestroy(comm);
    }
    sshash_free(whole);
    printf("EVERYTHING OK!\n");
    return 0;
}
