// engine.hpp -- host-side owner of the device-resident dictionary replicas and kernel launchers.
#pragma once

#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "device_layout.hpp"
#include "index.hpp"

namespace sshash_amd {

struct device_replica;  // defined in engine.hip

enum class out_mode : int { ids = 0, full = 1, member = 2 };

/* the one collective step of the sharded lookup, supplied by the caller (sshash_exchange in include/sshash_amd.h) */
struct exchange_ops {
    void* ctx;
    int (*counts)(void* ctx, const uint64_t* send, uint64_t* recv);
    int (*data)(void* ctx, const void* send, const uint64_t* send_counts, void* recv, const uint64_t* recv_counts, uint32_t elem_bytes,
                void* hip_stream);
};

struct streaming_report {  // streaming_query_report, include/util.hpp:21-36
    uint64_t num_kmers = 0, num_positive_kmers = 0, num_negative_kmers = 0, num_invalid_kmers = 0,
             num_searches = 0, num_extensions = 0;
};

class engine {
public:
    explicit engine(std::shared_ptr<host_index> idx);
    ~engine();

    host_index const& index() const { return *m_idx; }

    /* Copy the dictionary into the HBM of `device` (no-op when already there). */
    /* table_shards > 1: this replica's super-k-mer table holds only its share of the keys (device_layout.hpp (5)) */
    void to_device(int device, uint32_t table_shards = 1, uint32_t table_shard_id = 0);
    bool on_device(int device) const;
    std::vector<int> devices() const;
    uint64_t device_bytes(int device) const;
    /* {bytes in HBM, directory sectors, directory sectors flagged overflow, keys held by the directory,
        super-k-mer table slots (0 = disabled), its keys, its inline keys, its keys left to the complete path} */
    void device_stats(int device, uint64_t out[16]) const;
    void device_table_histogram(int device, uint64_t out[32]) const;

    /* Device-pointer entry points: queries and outputs already live in the HBM of `device`;
       the launch is asynchronous on `stream` (a hipStream_t, may be null = default stream).
       `packed`: n*W u64 words (W = 1 for k<=31, 2 otherwise). `ascii`: n*k chars, no NUL. */
    void lookup_packed_device(int device, uint64_t const* d_kmers, uint64_t n, bool check_rc, out_mode mode,
                              result_view const& d_out, uint8_t* d_member, void* stream) const;
    void lookup_ascii_device(int device, char const* d_kmers, uint64_t n, bool check_rc, out_mode mode,
                             result_view const& d_out, uint8_t* d_member, void* stream) const;

    /* kmer_neighbours (src/dictionary.cpp:111-187) for a batch: the 8 lookups per query, results at 8*i + which
       (which = 0..3 forward with A,C,T,G -- the character's 2-bit code --; 4..7 backward likewise). Device buffers / host buffers. */
    void neighbours_packed_device(int device, uint64_t const* d_kmers, uint64_t n, bool check_rc, out_mode mode,
                                  result_view const& d_out, void* stream) const;
    void neighbours_packed_host(uint64_t const* h_kmers, uint64_t n, bool check_rc, out_mode mode,
                                result_view const& h_out) const;

    /* string_neighbours (src/dictionary.cpp:189-201): forward neighbours of the last k-mer and backward neighbours
       of the first k-mer of every string; same 8-entry layout. Host buffers. */
    void string_neighbours_host(uint64_t const* h_string_ids, uint64_t n, bool check_rc, out_mode mode,
                                result_view const& h_out) const;

    /* access(kmer_id) for a batch of ids, device buffers: out gets n*W packed words
       (all-ones for an id >= num_kmers). */
    void access_packed_device(int device, uint64_t const* d_ids, uint64_t n, uint64_t* d_out, void* stream) const;

    /* weight(kmer_id) for a batch of ids, device buffers (all-ones for an id >= num_kmers); throws when the
       dictionary stores no weights */
    void weight_device(int device, uint64_t const* d_ids, uint64_t n, uint64_t* d_out, void* stream) const;

    /* owner shards of every query's forward / reverse-complement minimizer (shard_of_minimizer) */
    void route_packed_device(int device, uint64_t const* d_kmers, uint64_t n, uint32_t num_shards, uint32_t* d_owner_fwd,
                             uint32_t* d_owner_rc, void* stream) const;

    /* the same routing, bucketed on the device (see sshash_route_bucket_device in include/sshash_amd.h) */
    void route_bucket_device(int device, uint64_t const* d_kmers, uint64_t n, uint32_t num_shards, bool check_rc,
                             bool by_table_key, uint64_t* d_cursors, uint64_t* d_send, uint32_t* d_slots, void* stream,
                             uint32_t* d_known_owners = nullptr) const;  // n words kept between the counting and the scattering launch
    void route_combine_device(int device, uint64_t const* d_replies, uint32_t const* d_slots, uint64_t m, uint64_t* d_out,
                              void* stream, bool one_reply_per_query = false) const;

    /* lookup_packed_device over the places i with bit 0 of d_lane_valid[i] set; the outputs of the other places are
       left untouched (the position-parallel streaming lookup, streaming.hip) */
    void lookup_packed_masked_device(int device, uint64_t const* d_kmers, uint8_t const* d_lane_valid, uint64_t n, bool check_rc,
                                     out_mode mode, result_view const& d_out, void* stream) const;

    /* Host-buffer entry points: shard the batch over every replica, stream chunks through
       pinned staging buffers, results land in the caller's arrays. */
    void lookup_packed_host(uint64_t const* h_kmers, uint64_t n, bool check_rc, out_mode mode,
                            result_view const& h_out, uint8_t* h_member) const;
    void lookup_ascii_host(char const* h_kmers, uint64_t n, bool check_rc, out_mode mode, result_view const& h_out,
                           uint8_t* h_member) const;

    /* Batched streaming query over `n_reads` reads stored back to back in `bases`
       (read r = bases[read_offsets[r] .. read_offsets[r+1])). Host buffers. */
    streaming_report streaming_query_host(char const* bases, uint64_t const* read_offsets, uint64_t n_reads) const;
    /* An uncompressed FASTQ file, read and parsed by the lanes themselves (reads.hpp: fastq_pieces): every lane takes pieces of
       the file from a shared counter, parses a piece straight into its pinned block, uploads it and runs the streaming
       kernels -- no single reader thread, no intermediate batch. Returns false when the file turned out not to be four lines
       per record (or holds reads longer than a piece): the caller then takes the sequential reader, `total` is untouched. */
    bool streaming_query_fastq_pieces(std::string const& filename, streaming_report& total) const;
    /* Device buffers, asynchronous; `d_report` receives 6 u64 counters (accumulated). */
    void streaming_query_device(int device, char const* d_bases, uint64_t const* d_read_offsets, uint64_t n_reads,
                                uint64_t total_bases, uint64_t* d_report, void* stream) const;

    /* Per-k-mer results of the streaming query (streaming_query::lookup for every k-mer of every read,
       include/streaming_query.hpp:56-109): entry read_offsets[r] + j of every non-null array of `d_out` = the k-mer
       starting at base j of read r; places where no k-mer starts are left untouched. `d_report` (nullable): the six
       counters, accumulated. Device buffers, asynchronous. */
    void streaming_lookup_device(int device, char const* d_bases, uint64_t const* d_read_offsets, uint64_t n_reads,
                                 uint64_t total_bases, result_view const& d_out, uint64_t* d_report, void* stream) const;
    streaming_report streaming_lookup_host(char const* bases, uint64_t const* read_offsets, uint64_t n_reads,
                                           result_view const& h_out) const;

    /* Lookup against a dictionary partitioned over `num_ranks` GPUs (minimizer shards, or table shards with
       by_table_key): route -> exchange -> lookup -> return -> combine (sharded.cpp). Collective: every rank calls it, with
       its own local batch (n may be 0). d_out: n ids. */
    void sharded_lookup_device(int device, uint32_t num_ranks, bool by_table_key, uint64_t const* d_kmers, uint64_t n, bool check_rc,
                               uint64_t* d_out, exchange_ops const& x, void* stream) const;
    void sharded_lookup_rccl(int device, void* nccl_comm, bool by_table_key, uint64_t const* d_kmers, uint64_t n, bool check_rc,
                             uint64_t* d_out, void* stream) const;

    /* the replica resident on `device` (throws when there is none); internal to the .hip files */
    device_replica const* replica(int device) const;

private:
    std::shared_ptr<host_index> m_idx;
    /* to_device may run while other host threads query: readers share, the upload's final push_back is exclusive */
    mutable std::shared_mutex m_replicas_mutex;
    std::mutex m_upload_mutex;
    std::vector<std::unique_ptr<device_replica>> m_replicas;
};

int visible_device_count();  // 0 when no GPU / no driver
char const* isa_guard_state();  // "guarded": device code built through tools/isa_guard.py; "plain": by hipcc alone (engine.hip)

}  // namespace sshash_amd
