/* sshash_oracle.h -- CPU restatement of the SSHash Lookup path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product (sshash_amd/, include/) may include, link or call this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker / CPU baseline.
 *
 * Pinning status: the reference's lookup path cannot be compiled here (its PTHash dependency is
 * an empty submodule), so this restatement is pinned by
 *   (1) golden vectors produced in the build container from the two reference pieces that DO
 *       compile -- include/kmer.hpp and external/cityhash -- by oracle/ref_vectors.cpp
 *       (tests/golden/ref_vectors.json), plus python-xxhash for the m-mer hash magic;
 *   (2) the reference's own result contract, test/check_from_file.hpp:66-155: k-mer ids are the
 *       rank of the k-mer in the input file, checked against an independent ground truth built
 *       straight from the FASTA (oracle/ground_truth.py);
 *   (3) the known answers of test/test_alphabet.cpp:58-117 and README.md:222-223.
 * The MPHF is this repo's own PTHash-style function (values never leak into results); the
 * reference's .sshash byte format and concrete MPHF values are "parity unpinned" (no vector for
 * them exists in the reference).
 */
#ifndef SSHASH_ORACLE_H
#define SSHASH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_index oracle_index;

/* lookup_result, reference include/util.hpp:38-62 */
typedef struct oracle_result {
    uint64_t kmer_id;
    uint64_t kmer_id_in_string;
    uint64_t kmer_offset;
    int64_t kmer_orientation;
    uint64_t string_id;
    uint64_t string_begin;
    uint64_t string_end;
    uint8_t minimizer_found;
    uint8_t pad[7];
} oracle_result;

typedef struct oracle_info {
    uint32_t k, m, canonical, words_per_kmer;
    uint64_t num_kmers, num_strings, num_bases, num_minimizers, hash_magic;
} oracle_info;

/* ---- index (own parser of the file written by sshash_save) ---- */
int oracle_load(const char* filename, oracle_index** out, char* err, int err_len);
void oracle_free(oracle_index* idx);
void oracle_get_info(const oracle_index* idx, oracle_info* info);

/* ---- lookups ---- */
void oracle_lookup_packed(const oracle_index* idx, const uint64_t* kmers, uint64_t n, int check_rc, oracle_result* out);
void oracle_lookup_ascii(const oracle_index* idx, const char* kmers, uint64_t n, int check_rc, oracle_result* out);
/* ids only, `num_threads` pthreads over contiguous chunks (test/check.hpp:63-71); used for timing */
void oracle_lookup_ids(const oracle_index* idx, const uint64_t* kmers, uint64_t n, int check_rc, uint64_t* ids,
                       int num_threads);
/* algorithmic bytes of a batch under the counting rule of SURVEY.md section 8(d): 8 bytes per
 * distinct 64-bit index word dereferenced per query + 8 B query in + 8 B id out; a string
 * endpoint `locate` is charged 24 B. */
uint64_t oracle_count_bytes(const oracle_index* idx, const uint64_t* kmers, uint64_t n, int check_rc);
void oracle_access(const oracle_index* idx, uint64_t kmer_id, char* out_k_chars);
/* weights::weight, include/weights.hpp:147-152; *ok = 0 when there are no weights / the id is out of range */
uint64_t oracle_weight(const oracle_index* idx, uint64_t kmer_id, int* ok);
int oracle_weights(const oracle_index* idx, const uint64_t* kmer_ids, uint64_t n, uint64_t* out); /* 1 = all ok */

/* ---- streaming query: report = {num_kmers, positive, negative, invalid, searches, extensions} ---- */
void oracle_streaming_query(const oracle_index* idx, const char* bases, const uint64_t* read_offsets, uint64_t num_reads,
                            uint64_t report[6]);
/* per-k-mer results of ONE read (len - k + 1 entries), as streaming_query::lookup returns them */
void oracle_streaming_read(const oracle_index* idx, const char* read, uint64_t len, oracle_result* out);
/* algorithmic bytes of a streaming query: 8 bytes per distinct index word the reference's state machine dereferences per k-mer + 1 per base */
uint64_t oracle_streaming_count_bytes(const oracle_index* idx, const char* bases, const uint64_t* read_offsets, uint64_t num_reads);

/* ---- primitives, exposed so that they can be pinned against golden vectors ---- */
void oracle_encode_kmer(const char* s, uint32_t k, uint64_t out[2]);
void oracle_revcomp(const uint64_t in[2], uint32_t k, int words, uint64_t out[2]);
void oracle_minimizer(const uint64_t kmer[2], uint32_t k, uint32_t m, uint64_t magic, int words, uint64_t* value,
                      uint64_t* pos);
void oracle_city128(const void* key, int len /* 8 or 16 */, uint64_t seed, uint64_t out[2]);
uint64_t oracle_xxh64_u64(uint64_t value, uint64_t seed);
int oracle_is_valid_base(char c);

#ifdef __cplusplus
}
#endif
#endif
