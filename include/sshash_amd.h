/* sshash_amd.h -- C ABI of the MI355X-native batched k-mer Lookup engine for SSHash.
 *
 * The reference (jermp/sshash) has no FFI layer: its boundary for this path is the C++ class
 * `sshash::dictionary<Kmer, Offsets>` (reference include/dictionary.hpp:10-181). Each entry
 * point below names the reference interface it stands in for. Conventions:
 *   - opaque handle, plain pointers and sizes, no exceptions across the ABI;
 *   - every function returns an sshash_status; sshash_last_error() gives the message of the
 *     last failure on the calling thread;
 *   - "not found" is NOT an error: kmer_id == SSHASH_INVALID_U64 (include/constants.hpp:5);
 *   - lookups run on the GPU only. Without a visible HIP device they fail with
 *     SSHASH_ERR_NO_DEVICE; there is no CPU fallback.
 *   - k-mers are 2-bit packed, first base in the least-significant bits, A=0 C=1 T=2 G=3
 *     (include/kmer.hpp:80,194); W = 1 64-bit word per k-mer for k <= 31, W = 2 for k <= 63.
 */
#ifndef SSHASH_AMD_H
#define SSHASH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSHASH_INVALID_U64 UINT64_MAX

typedef enum sshash_status {
    SSHASH_OK = 0,
    SSHASH_ERR_ARGUMENT = 1,  /* null pointer, bad k/m, id out of range ...                    */
    SSHASH_ERR_IO = 2,        /* "error in opening the file" (src/query.cpp:128, tools/build.cpp) */
    SSHASH_ERR_FORMAT = 3,    /* not an index file / corrupt                                     */
    SSHASH_ERR_VERSION = 4,   /* "MAJOR index version mismatch" (include/util.hpp:191-195)       */
    SSHASH_ERR_NO_DEVICE = 5, /* no HIP device visible, or dictionary not resident on the device  */
    SSHASH_ERR_HIP = 6,       /* a HIP runtime call failed                                       */
    SSHASH_ERR_BUILD = 7,     /* index construction failed                                       */
    SSHASH_ERR_INTERNAL = 8
} sshash_status;

typedef struct sshash_dict sshash_dict; /* stands for sshash::dictionary_type (include/dictionary_types.hpp:9) */

/* build_configuration (include/util.hpp:143-159); zero-initialise then override. */
typedef struct sshash_build_config {
    uint32_t k;           /* default 31 */
    uint32_t m;           /* default 20 */
    uint64_t seed;        /* default 1  */
    uint32_t canonical;   /* default 0  */
    uint32_t num_threads; /* default 1; 0 = hardware concurrency */
    double lambda;        /* default 5.0 */
    uint32_t verbose;
    uint32_t weighted;    /* build_configuration::weighted: FASTA headers carry '>[id] LN:i:[len] ab:Z:[weights]'
                             (src/builder/encode_strings.cpp:83-135); sshash_build_from_fasta only */
    /* minimizer-sharded build for dictionaries larger than one GPU's HBM: keep only the buckets of the
     * minimizers owned by shard `shard_id` of `num_shards` (strings stay complete). 0/1 = whole index. */
    uint32_t num_shards;
    uint32_t shard_id;
} sshash_build_config;

/* accessors of dictionary (include/dictionary.hpp:31-38) */
typedef struct sshash_info {
    uint8_t version[3];
    uint8_t canonical;
    uint32_t k, m;
    uint32_t words_per_kmer;
    uint64_t num_kmers, num_strings, num_bases, num_minimizers;
    uint64_t num_bits;      /* dictionary::num_bits(), host representation */
    uint32_t skew_partitions;
    uint32_t weighted;   /* dictionary::weighted() */
    uint32_t num_shards; /* 1 unless built as one shard of a minimizer-partitioned index */
    uint32_t shard_id;
} sshash_info;

/* lookup_result (include/util.hpp:38-62) as a struct of arrays: one entry per query.
 * kmer_id is mandatory for sshash_lookup_*; any other pointer may be NULL (field not produced).
 * A query that is not found has every u64 field == SSHASH_INVALID_U64. */
typedef struct sshash_results {
    uint64_t* kmer_id;
    uint64_t* kmer_id_in_string;
    uint64_t* kmer_offset;
    uint64_t* string_id;
    uint64_t* string_begin;
    uint64_t* string_end;
    int8_t* kmer_orientation;  /* +1 forward, -1 backward (include/constants.hpp:17-18) */
    uint8_t* minimizer_found;
} sshash_results;

/* streaming_query_report (include/util.hpp:21-36) */
typedef struct sshash_streaming_report {
    uint64_t num_kmers;
    uint64_t num_positive_kmers;
    uint64_t num_negative_kmers;
    uint64_t num_invalid_kmers;
    uint64_t num_searches;
    uint64_t num_extensions;
} sshash_streaming_report;

const char* sshash_last_error(void);
/* how the library was built, "key=value;..." (no reference counterpart): isa_guard=guarded|plain -- whether the device code went
 * through tools/isa_guard.py, which keeps a gfx950 register hazard out of the kernels (a plain build also warns at the first upload) */
const char* sshash_build_info(void);
void sshash_build_config_default(sshash_build_config* cfg);

/* ---- construction / persistence: dictionary::build (src/builder/build.cpp:10-28),
 *      essentials::save / load (tools/build.cpp:90-95, tools/common.hpp:19-22) ------------- */
sshash_status sshash_build_from_fasta(const char* filename, const sshash_build_config* cfg, sshash_dict** out);
/* strings already 2-bit packed back to back: `words` holds endpoints[num_strings] bases. */
sshash_status sshash_build_from_packed(const uint64_t* words, const uint64_t* endpoints, uint64_t num_strings,
                                       const sshash_build_config* cfg, sshash_dict** out);
sshash_status sshash_save(const sshash_dict* d, const char* filename);
sshash_status sshash_load(const char* filename, sshash_dict** out);
void sshash_free(sshash_dict* d);
sshash_status sshash_get_info(const sshash_dict* d, sshash_info* info);

/* The bucket statistics `sshash build --verbose` prints (src/builder/build_sparse_and_skew_index.cpp:64-99,
 * include/buckets_statistics.hpp: "num_buckets_larger_than_1_not_in_skew_index", "num_buckets_in_skew_index",
 * "max_bucket_size", "num kmers in skew index", "buckets with s minimizer positions"), recomputed from the finished index:
 * out = { [0] minimizers, [1] minimizer positions, [2] buckets of 2..64 positions, [3] positions in them, [4] buckets in the
 *         skew index, [5] positions in them, [6] k-mers in the skew index, [7] largest bucket, [8..15] k-mers per skew
 *         partition, [16..31] buckets of exactly 1..16 positions, [32] k-mers, [33] strings, [34] bases, [35] skew
 *         partitions, [36] longest string, [37..63] zero } */
sshash_status sshash_bucket_stats(const sshash_dict* d, uint64_t out[64]);

/* ---- device residency (no reference counterpart: the reference is host-only) ------------- */
int sshash_device_count(void);
sshash_status sshash_to_device(sshash_dict* d, int device);
/* The same, with the super-k-mer table (the largest device structure) partitioned over several
 * GPUs: this replica builds the slots of the keys it owns only (shard `table_shard_id` of `num_table_shards`); queries
 * are meant to reach it through sshash_route_bucket_by_key_device, but any query is still answered correctly (a key
 * of another shard takes the complete path). Everything else is resident in full. */
sshash_status sshash_to_device_table_shard(sshash_dict* d, int device, uint32_t num_table_shards, uint32_t table_shard_id);
sshash_status sshash_device_bytes(const sshash_dict* d, int device, uint64_t* bytes);
/* out = { [0] bytes in HBM, [1] minimizer-directory sectors (0 = disabled), [2] sectors flagged overflow, [3] keys in the
 *         directory, [4] super-k-mer table slots (0 = no table), [5] its keys, [6] keys held inline (<= 4 occurrences),
 *         [7] items left to the complete path (no free slot in any of their buckets), [8] slots in use (load factor =
 *         [8] / [4]), [9] heavy keys (a marker + one slot per k-mer), [10] k-mers entered one by one, [11] why there is no
 *         table: 0 = there is one, 1 = disabled (SSHASH_AMD_SKTABLE=0), 2 = minimizer shard (keeps the directory path),
 *         3 = more than 2^39 bases, 4 = more items than one build pass holds (2^31 super-k-mers), 5 = not enough free HBM
 *         -- in all these cases lookups take the directory / MPHF path, about half as fast --, [12] bytes of the table,
 *         [13..15] reserved } */
sshash_status sshash_device_stats(const sshash_dict* d, int device, uint64_t out[16]);
/* The keys of the super-k-mer table by number of occurrences -- the table-side counterpart of sshash_bucket_stats (a key
 * with more than 4 occurrences is "heavy": one slot per k-mer, one more bucket read per lookup):
 * out = { [0..9) keys with 1, 2, 3, 4, 5-8, 9-16, 17-64, 65-1024, > 1024 occurrences, [9..18) the occurrences (super-k-mers)
 *         in each of these bins, [18] super-k-mers in all, [19] slots asked for (inline occurrences + markers + k-mers of heavy
 *         keys), [20..31] zero }; all zero when the replica has no table. */
sshash_status sshash_device_table_histogram(const sshash_dict* d, int device, uint64_t out[32]);

/* ---- dictionary::lookup(Kmer, bool) / lookup(char const*, bool): include/dictionary.hpp:41-42,
 *      src/dictionary.cpp:58-78. Batched. ------------------------------------------------------
 * *_device: `kmers` and every non-NULL array of `out` are DEVICE pointers in the HBM of `device`;
 *           the launch is asynchronous on `hip_stream` (hipStream_t as void*, NULL = default stream).
 * host variants: caller-owned host buffers; the batch is sharded over all resident devices. Page-locked buffers (hipHostMalloc /
 *           hipHostRegister; input AND every requested output) are copied from and to directly: 4.3 G lookups/s over PCIe
 *           against 1.5 G/s for pageable memory, which is staged through the library's own pinned lanes.
 * Cost of the fields: NULL arrays are skipped. kmer_id alone is answered by the device's super-k-mer table at full speed (DESIGN.md
 * section 6: 38-40 G lookups/s); the position fields come from the same probe (all seven: 25 G/s at 50 % positives -- seven output
 * streams, and string_begin / string_end cost a hit one more random read). `minimizer_found` is true for a hit; for a miss only the
 * MPHF can reproduce it -- the flag of an absent minimizer depends on which bucket the MPHF maps that minimizer to
 * (include/spectrum_preserving_string_set.hpp:46-65) --, so the misses of a batch that asks for it go through the MPHF path in a
 * last pass that stores that one byte (one probe: the reference's result for a miss is that of its last probe): 20 G/s at 100 %
 * positives, 14 at 50 %, 11 at 0 %. The eight-field result is an extension: the reference's own comparator of lookup results ignores
 * that flag (include/util.hpp:107-141), and for an absent minimizer its value is an artefact of this build's MPHF. */
sshash_status sshash_lookup_packed_device(const sshash_dict* d, int device, const uint64_t* kmers, uint64_t n,
                                          int check_reverse_complement, const sshash_results* out, void* hip_stream);
sshash_status sshash_lookup_ascii_device(const sshash_dict* d, int device, const char* kmers, uint64_t n,
                                         int check_reverse_complement, const sshash_results* out, void* hip_stream);
sshash_status sshash_lookup_packed(const sshash_dict* d, const uint64_t* kmers, uint64_t n, int check_reverse_complement,
                                   const sshash_results* out);
sshash_status sshash_lookup_ascii(const sshash_dict* d, const char* kmers, uint64_t n, int check_reverse_complement,
                                  const sshash_results* out);

/* ---- dictionary::is_member: include/dictionary.hpp:75-76, src/dictionary.cpp:80-88 -------- */
sshash_status sshash_is_member_packed_device(const sshash_dict* d, int device, const uint64_t* kmers, uint64_t n,
                                             int check_reverse_complement, uint8_t* out, void* hip_stream);
sshash_status sshash_is_member_packed(const sshash_dict* d, const uint64_t* kmers, uint64_t n, int check_reverse_complement,
                                      uint8_t* out);
sshash_status sshash_is_member_ascii(const sshash_dict* d, const char* kmers, uint64_t n, int check_reverse_complement,
                                     uint8_t* out);

/* ---- dictionary::access (include/dictionary.hpp:69, src/dictionary.cpp:90-94): host side,
 *      used to draw positive queries as tools/perf.hpp:38-51 does ---------------------------- */
sshash_status sshash_access(const sshash_dict* d, uint64_t kmer_id, char* out_k_chars);
sshash_status sshash_access_packed(const sshash_dict* d, const uint64_t* kmer_ids, uint64_t n, uint64_t* out_words);
/* the same on the GPU: device pointers, asynchronous; an id >= num_kmers yields all-ones words */
sshash_status sshash_access_packed_device(const sshash_dict* d, int device, const uint64_t* kmer_ids, uint64_t n,
                                          uint64_t* out_words, void* hip_stream);

/* ---- dictionary::weight(kmer_id): include/dictionary.hpp (weight), src/dictionary.cpp:96-100,
 *      include/weights.hpp:147-152. Batched; SSHASH_ERR_ARGUMENT when the dictionary stores no weights
 *      (the reference's checker refuses it the same way, test/check_from_file.hpp:234-237) or an id is
 *      >= num_kmers (host variant; the device variant writes UINT64_MAX for such an id). */
sshash_status sshash_weight(const sshash_dict* d, const uint64_t* kmer_ids, uint64_t n, uint64_t* out_weights);
sshash_status sshash_weight_device(const sshash_dict* d, int device, const uint64_t* kmer_ids, uint64_t n,
                                   uint64_t* out_weights, void* hip_stream);

/* ---- dictionary::kmer_neighbours(Kmer, bool): include/dictionary.hpp:59-61, src/dictionary.cpp:111-126,176-187.
 *      Batched: every array of `out` holds 8*n entries; entry 8*i + c (c = 0..3) is the lookup of the forward
 *      neighbour suffix(kmer i) + "ACTG"[c], entry 8*i + 4 + c the backward neighbour "ACTG"[c] + prefix(kmer i): c is
 *      the 2-bit code of the character, the reference's alphabet order (include/kmer.hpp:118; its own checker indexes
 *      forward[char_to_uint(next)], test/check_from_file.hpp:198-216)
 *      (neighbourhood::forward / ::backward, include/util.hpp:78-81). Same pointer rules as the lookups. */
sshash_status sshash_neighbours_packed_device(const sshash_dict* d, int device, const uint64_t* kmers, uint64_t n,
                                              int check_reverse_complement, const sshash_results* out, void* hip_stream);
sshash_status sshash_neighbours_packed(const sshash_dict* d, const uint64_t* kmers, uint64_t n,
                                       int check_reverse_complement, const sshash_results* out);

/* dictionary::string_size(string_id): include/dictionary.hpp:44-46, src/dictionary.cpp:102-109 -- the number of
 * k-mers of each string (its length is size + k - 1). Host arrays. */
sshash_status sshash_string_size(const sshash_dict* d, const uint64_t* string_ids, uint64_t n, uint64_t* out_sizes);
/* dictionary::string_offsets(string_id) -> [begin, end) in bases (include/dictionary.hpp:105-108): what lookup_result's
 * string_begin / string_end refer to */
sshash_status sshash_string_offsets(const sshash_dict* d, const uint64_t* string_ids, uint64_t n, uint64_t* out_begin, uint64_t* out_end);

/* dictionary::string_neighbours(string_id, bool): include/dictionary.hpp:62, src/dictionary.cpp:189-201 -- the
 * forward neighbours of the string's last k-mer and the backward neighbours of its first one, same layout. */
sshash_status sshash_string_neighbours(const sshash_dict* d, const uint64_t* string_ids, uint64_t n,
                                       int check_reverse_complement, const sshash_results* out);

/* ---- dictionary::streaming_query_from_file (include/dictionary.hpp:81-82, src/query.cpp:118-175)
 *      and streaming_query<Dict,canonical> over reads in memory (include/streaming_query.hpp) --
 * The file (.fa/.fasta/.fq/.fastq, optionally .gz) is read ~256 MiB of bases at a time by a reader thread while the
 * devices work on the previous batch: host memory stays bounded whatever the size of the file. The call runs at the reader's
 * pace (the devices are busy a tenth of the time): a plain file at ~9 GB/s, a .gz at one thread's inflate (0.28 G bases/s), a BGZF
 * .gz (bgzip's gzip members, recognised by the first header) inflated on min(16, cores) threads (SSHASH_AMD_READER_THREADS). */
sshash_status sshash_streaming_query_from_file(const sshash_dict* d, const char* filename, int multiline,
                                               sshash_streaming_report* report);
/* reads stored back to back: read r = bases[read_offsets[r] .. read_offsets[r+1]) ; host buffers */
sshash_status sshash_streaming_query(const sshash_dict* d, const char* bases, const uint64_t* read_offsets,
                                     uint64_t num_reads, sshash_streaming_report* report);
/* device buffers; `report` is a device pointer to 6 uint64 counters, accumulated into. `total_bases` = read_offsets[num_reads] (as
 * sshash_streaming_lookup_device takes it): the size of the 2-bit packed copy of the reads the call makes in scratch the replica keeps
 * per stream. With it the call only enqueues work on `hip_stream` and returns. 0 = "not known to the caller": the call then reads
 * read_offsets[num_reads] back (8 bytes) and so waits for what the stream holds at that moment -- its one synchronisation. */
sshash_status sshash_streaming_query_device(const sshash_dict* d, int device, const char* bases,
                                            const uint64_t* read_offsets, uint64_t num_reads, uint64_t total_bases,
                                            uint64_t* report, void* hip_stream);

/* ---- streaming_query<Dict,canonical>::lookup for EVERY k-mer of every read (include/streaming_query.hpp:56-109),
 *      batched: what the reference returns k-mer by k-mer while it streams a read. Every non-NULL array of `out` has one
 *      entry per BASE of `bases` (total_bases = read_offsets[num_reads] entries): entry read_offsets[r] + j is the result
 *      of the k-mer starting at base j of read r; entries of places where no k-mer starts (the last k-1 bases of a
 *      read, reads shorter than k) are left untouched. A k-mer holding a character other than ACGTacgt gets the
 *      default result the reference returns after its reset() (every u64 field SSHASH_INVALID_U64, orientation +1).
 *      Results equal the point lookups' (the reference asserts exactly that, :107); minimizer_found is not produced
 *      (SSHASH_ERR_ARGUMENT if asked for). `report` (may be NULL; device variant: 6 uint64 counters, accumulated into)
 *      receives the same counters as sshash_streaming_query. ---- */
sshash_status sshash_streaming_lookup_device(const sshash_dict* d, int device, const char* bases, const uint64_t* read_offsets,
                                             uint64_t num_reads, uint64_t total_bases, const sshash_results* out, uint64_t* report,
                                             void* hip_stream);
sshash_status sshash_streaming_lookup(const sshash_dict* d, const char* bases, const uint64_t* read_offsets, uint64_t num_reads,
                                      const sshash_results* out, sshash_streaming_report* report);

/* ---- routing for a minimizer-sharded index: owner shard of the forward
 *      minimizer and of the reverse-complement minimizer of every query (equal for canonical
 *      dictionaries: the smaller-valued minimizer decides). Device pointers, asynchronous. ----------- */
sshash_status sshash_route_packed_device(const sshash_dict* d, int device, const uint64_t* kmers, uint64_t n,
                                         uint32_t num_shards, uint32_t* owner_forward, uint32_t* owner_reverse,
                                         void* hip_stream);

/* The same routing with the bucketing done on the device, two calls on one stream:
 *   1. send == slots == NULL: cursors[s] += number of messages for shard s (one message per query and distinct
 *      owner; `cursors`: num_shards device uint64, zeroed by the caller) -- the send counts of the all-to-all;
 *   2. cursors[s] = index of the first message of shard s (exclusive prefix sum of the counts): message t gets
 *      send[t*W .. t*W+W) = the packed k-mer and slots[t] = the index of its query. Messages of one shard are
 *      contiguous; their order inside the shard is unspecified. n < 2^32, num_shards <= 1024.
 * sshash_route_combine_device: out[slots[t]] = replies[t] for every reply != UINT64_MAX (`out` pre-filled with
 * UINT64_MAX by the caller); owners that both find a k-mer return the same id. */
sshash_status sshash_route_bucket_device(const sshash_dict* d, int device, const uint64_t* kmers, uint64_t n,
                                         uint32_t num_shards, int check_reverse_complement, uint64_t* cursors,
                                         uint64_t* send, uint32_t* slots, void* hip_stream);
/* Bucketing for table shards (sshash_to_device_table_shard): the owner of a query is the owner of its table key --
 * strand-symmetric, so exactly ONE message per query. Same two-call protocol as above. */
sshash_status sshash_route_bucket_by_key_device(const sshash_dict* d, int device, const uint64_t* kmers, uint64_t n,
                                                uint32_t num_shards, uint64_t* cursors, uint64_t* send, uint32_t* slots,
                                                void* hip_stream);
sshash_status sshash_route_combine_device(const sshash_dict* d, int device, const uint64_t* replies, const uint32_t* slots,
                                          uint64_t m, uint64_t* out, void* hip_stream);

/* ---- the whole sharded lookup as ONE call (BASELINE.json configs[4]): route -> exchange -> lookup
 *      -> return -> combine against a dictionary partitioned over `num_ranks` GPUs -- minimizer shards
 *      (sshash_build_config.num_shards / shard_id; by_table_key = 0) or table shards (sshash_to_device_table_shard;
 *      by_table_key = 1). Collective: every rank calls it with its own local batch of n packed k-mers (n may be 0) and
 *      receives the n ids, identical to the unpartitioned dictionary's. The one step that needs communication is handed
 *      in as two callbacks (return 0 on success):
 *        counts  all-to-all of ONE uint64 per peer: send[p] goes to rank p, recv[p] comes from rank p (host arrays of
 *                num_ranks entries). The words are opaque to the callback: the library carries the length of its table keys in
 *                their top byte, so that ranks built under different SSHASH_AMD_SK_M refuse to work together (SSHASH_ERR_ARGUMENT
 *                on every rank, after this exchange and before the data exchange) -- in the same exchange on every call, so no rank
 *                ever runs a collective step its peers skip;
 *        data    all-to-all-v of DEVICE buffers: the block for rank p starts after the blocks of the ranks before it,
 *                send_counts[p] / recv_counts[p] elements of elem_bytes each (host arrays); it must be complete, or ordered
 *                on hip_stream, when it returns.
 *      sshash_sharded_lookup_rccl supplies them over an RCCL communicator (grouped ncclSend/ncclRecv: the all-to-all
 *      over xGMI); `nccl_comm` is an ncclComm_t whose rank r holds shard r. RCCL is resolved when first used.
 *      Like any collective, the call completes only if every rank makes it: a rank that fails before the exchange (an argument
 *      error, an allocation failure: it returns its status without having called `counts`) leaves its peers waiting in theirs --
 *      the host application's job control has to take the group down, as it would for a failed ncclAllReduce. ---- */
typedef struct sshash_exchange {
    void* ctx;
    int (*counts)(void* ctx, const uint64_t* send, uint64_t* recv);
    int (*data)(void* ctx, const void* send, const uint64_t* send_counts, void* recv, const uint64_t* recv_counts,
                uint32_t elem_bytes, void* hip_stream);
} sshash_exchange;
sshash_status sshash_sharded_lookup_device(const sshash_dict* d, int device, uint32_t num_ranks, int by_table_key,
                                           const uint64_t* kmers, uint64_t n, int check_reverse_complement, uint64_t* kmer_ids,
                                           const sshash_exchange* exchange, void* hip_stream);
sshash_status sshash_sharded_lookup_rccl(const sshash_dict* d, int device, void* nccl_comm, int by_table_key, const uint64_t* kmers,
                                         uint64_t n, int check_reverse_complement, uint64_t* kmer_ids, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* SSHASH_AMD_H */
