/* sshash_oracle.c -- CPU restatement of the SSHash Lookup path.  TEST INFRASTRUCTURE ONLY
 * (see sshash_oracle.h for who may use it and how it is pinned).
 *
 * Plain C, one query at a time, the reference's data layout (a 2-bit bit-vector for the strings,
 * a sorted endpoint sequence, bit-packed compact vectors) and the reference's control flow.
 * Every function names the reference lines it follows (paths relative to the reference checkout).
 * The GPU engine uses a different memory layout and different code, so agreement between the two
 * is meaningful.
 */
#include "sshash_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

#define INVALID_U64 UINT64_MAX /* include/constants.hpp:5 */
#define MIN_L 6                /* include/constants.hpp:13 */

/* ------------------------------------------------------------------------------------------ */
/* containers                                                                                  */

typedef struct {
    uint64_t size;
    uint32_t width;
    uint64_t* words;
    uint64_t num_words;
} cvec; /* bits::compact_vector */

typedef struct {
    uint64_t key_offset, pilot_base, free_base;
    uint32_t num_keys, table_size, dense_buckets, sparse_buckets;
    uint64_t reserved;
} mphf_part;

typedef struct {
    uint64_t seed, num_keys;
    uint32_t pilot_width;
    uint64_t num_parts;
    mphf_part* parts;
    uint64_t* pilots;
    uint64_t num_pilot_words;
    uint32_t* free_slots;
    uint64_t num_free;
} mphf;

struct oracle_index {
    uint32_t k, m, canonical, W, skew_parts, num_shards, shard_id;
    uint64_t num_kmers, num_strings, num_bases, hash_magic, build_seed;
    uint64_t* strings;
    uint64_t strings_words, strings_num_bits;
    uint64_t* endpoints;
    mphf minimizers;
    cvec codewords;
    uint32_t* begin_buckets_of_size;
    cvec mid_load;
    mphf skew_f[8];
    cvec skew_pos[8];
    cvec heavy_load;
    /* weights (include/weights.hpp): interval i = ids [weight_starts[i], weight_starts[i+1]) -> weight_values[i] */
    uint64_t* weight_starts;
    uint64_t* weight_values;
    uint64_t num_weight_intervals;
};

/* optional per-query instrumentation: distinct 64-bit words touched */
typedef struct {
    int n;
    struct {
        const void* base;
        uint64_t word;
    } e[512];
    uint64_t extra_bytes;
} touch_ctx;

static __thread touch_ctx* g_touch = NULL;

static inline void touch(const void* base, uint64_t word) {
    touch_ctx* t = g_touch;
    if (!t) return;
    for (int i = 0; i < t->n; ++i)
        if (t->e[i].base == base && t->e[i].word == word) return;
    if (t->n < 512) {
        t->e[t->n].base = base;
        t->e[t->n].word = word;
        ++t->n;
    }
}

static inline uint64_t low_mask(uint32_t bits) { return bits >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << bits) - 1); }

static inline uint64_t cvec_get(const cvec* v, uint64_t i) {
    const uint64_t bit = i * v->width, word = bit >> 6;
    const uint32_t sh = (uint32_t)(bit & 63);
    touch(v->words, word);
    uint64_t x = v->words[word] >> sh;
    if (sh + v->width > 64) {
        touch(v->words, word + 1);
        x |= v->words[word + 1] << (64 - sh);
    }
    return x & low_mask(v->width);
}

/* ------------------------------------------------------------------------------------------ */
/* k-mer primitives                                                                            */

/* include/kmer.hpp:194 */
static inline uint64_t char_to_uint(char c) { return ((uint64_t)(unsigned char)c >> 1) & 3; }

/* include/kmer.hpp:209-219,253-255: A C G T a c g t */
int oracle_is_valid_base(char c) {
    switch (c) {
        case 'A': case 'C': case 'G': case 'T': case 'a': case 'c': case 'g': case 't': return 1;
        default: return 0;
    }
}

static inline u128 take_bits(u128 x, uint32_t bits) { return bits >= 128 ? x : (x & (((u128)1 << bits) - 1)); }

/* util::string_to_uint_kmer, include/util.hpp:207-213 */
static u128 string_to_kmer(const char* s, uint32_t k) {
    u128 x = 0;
    for (uint32_t i = 0; i < k; ++i) x |= (u128)char_to_uint(s[i]) << (2 * i);
    return x;
}

/* crc64, include/kmer.hpp:141-157 */
static uint64_t crc64(uint64_t x) {
    uint64_t c = x ^ 0xaaaaaaaaaaaaaaaaULL;
    uint64_t r = __builtin_bswap64(c);
    r = ((r & 0x0f0f0f0f0f0f0f0fULL) << 4) | ((r & 0xf0f0f0f0f0f0f0f0ULL) >> 4);
    r = ((r & 0x3333333333333333ULL) << 2) | ((r & 0xccccccccccccccccULL) >> 2);
    return r;
}

/* reverse_complement_inplace, include/kmer.hpp:159-165 (W = words of the k-mer type) */
static u128 revcomp(u128 x, uint32_t k, int W) {
    if (W == 1) return (u128)(crc64((uint64_t)x) >> (64 - 2 * k));
    u128 r = ((u128)crc64((uint64_t)x) << 64) | crc64((uint64_t)(x >> 64));
    return r >> (128 - 2 * k);
}

/* mixer_64::hash, include/hash_util.hpp:91 */
static inline uint64_t mmer_hash(uint64_t x, uint64_t magic) { return (x * 0x517cc1b727220a95ULL) ^ magic; }

typedef struct {
    uint64_t minimizer;
    uint64_t pos_in_kmer;
} mini_info;

/* util::compute_minimizer, include/util.hpp:262-283 */
static mini_info compute_minimizer(u128 kmer, uint32_t k, uint32_t m, uint64_t magic) {
    uint64_t min_hash = INVALID_U64;
    mini_info r = {INVALID_U64, 0};
    for (uint32_t i = 0; i != k - m + 1; ++i) {
        const uint64_t mmer = (uint64_t)take_bits(kmer, 2 * m);
        const uint64_t h = mmer_hash(mmer, magic);
        if (h < min_hash) {
            min_hash = h;
            r.minimizer = mmer;
            r.pos_in_kmer = i;
        }
        kmer >>= 2;
    }
    return r;
}

/* XXH64 of 8 bytes; the reference seeds its m-mer hasher with it (include/hash_util.hpp:88) */
uint64_t oracle_xxh64_u64(uint64_t value, uint64_t seed) {
    const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL,
                   P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    uint64_t h = seed + P5 + 8;
    uint64_t k1 = value * P2;
    k1 = (k1 << 31) | (k1 >> 33);
    k1 *= P1;
    h ^= k1;
    h = ((h << 27) | (h >> 37)) * P1 + P4;
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return h;
}

/* ------------------------------------------------------------------------------------------ */
/* CityHash128WithSeed for short keys (external/cityhash/cityhash.cpp:116-135,238-269)          */

static const uint64_t CK1 = 0xb492b66fbe98f273ULL;

static inline uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }

static inline uint64_t hash_len16(uint64_t u, uint64_t v) { /* cityhash.hpp:90-99 */
    const uint64_t mul = 0x9ddfea08eb382d69ULL;
    uint64_t a = (u ^ v) * mul;
    a ^= (a >> 47);
    uint64_t b = (v ^ a) * mul;
    b ^= (b >> 47);
    b *= mul;
    return b;
}

static uint64_t hash_len_0_to_16(const unsigned char* s, size_t len) { /* cityhash.cpp:116-135 */
    if (len > 8) {
        uint64_t a, b;
        memcpy(&a, s, 8);
        memcpy(&b, s + len - 8, 8);
        const uint64_t t = b + len;
        return hash_len16(a, (t >> len) | (t << (64 - len))) ^ b;
    }
    if (len >= 4) {
        uint32_t a, b;
        memcpy(&a, s, 4);
        memcpy(&b, s + len - 4, 4);
        return hash_len16(len + ((uint64_t)a << 3), b);
    }
    return 0; /* shorter keys never occur on this path */
}

void oracle_city128(const void* key, int len, uint64_t seed, uint64_t out[2]) { /* cityhash.cpp:238-269 */
    const unsigned char* s = (const unsigned char*)key;
    uint64_t a = seed, b = ~seed, c, d, first;
    memcpy(&first, s, 8);
    a = shift_mix(a * CK1) * CK1;
    c = b * CK1 + hash_len_0_to_16(s, (size_t)len);
    d = shift_mix(a + first);
    a = hash_len16(a, c);
    b = hash_len16(d, b);
    out[0] = a ^ b;
    out[1] = hash_len16(b, a);
}

/* ------------------------------------------------------------------------------------------ */
/* MPHF evaluation (this repo's PTHash-style function; sshash_amd/csrc/mphf.hpp is the spec)    */

static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

static uint64_t mphf_eval(const mphf* f, const uint64_t h[2]) {
    const uint32_t pi = mulhi32((uint32_t)((h[0] ^ h[1]) >> 32), (uint32_t)f->num_parts);
    const mphf_part* p = &f->parts[pi];
    /* partition table: a few KB, cache-resident -> not counted (SURVEY.md 8(d) rule) */
    const uint32_t sel = (uint32_t)(h[0] >> 32), v = (uint32_t)h[0];
    const uint32_t bucket = sel < 0x9999999Au ? mulhi32(v, p->dense_buckets) : p->dense_buckets + mulhi32(v, p->sparse_buckets);
    cvec pv = {0, f->pilot_width, f->pilots, f->num_pilot_words};
    const uint64_t pilot = cvec_get(&pv, p->pilot_base + bucket);
    uint64_t x = h[1] ^ (pilot * 0x9E3779B97F4A7C15ULL);
    x = (x ^ (x >> 32)) * 0xD6E8FEB86659FD93ULL;
    uint32_t pos = mulhi32((uint32_t)(x >> 32), p->table_size);
    if (pos >= p->num_keys) {
        const uint64_t at = p->free_base + (pos - p->num_keys);
        touch(f->free_slots, at >> 1);
        pos = f->free_slots[at];
    }
    return p->key_offset + pos;
}

/* ------------------------------------------------------------------------------------------ */
/* strings                                                                                     */

/* bits::bit_vector::get_word64: 64 bits starting at an arbitrary bit position */
static inline uint64_t get_word64(const oracle_index* d, uint64_t pos) {
    const uint64_t block = pos >> 6;
    const uint32_t shift = (uint32_t)(pos & 63);
    touch(d->strings, block);
    uint64_t w = d->strings[block] >> shift;
    if (shift && block + 1 < d->strings_words) {
        touch(d->strings, block + 1);
        w |= d->strings[block + 1] << (64 - shift);
    }
    return w;
}

/* util::read_kmer_at, include/util.hpp:248-257 */
static u128 read_kmer_at(const oracle_index* d, uint32_t k, uint64_t pos) {
    u128 kmer = 0;
    for (int i = 64 * (int)d->W - 64; i >= 0; i -= 64) {
        if (pos + (uint64_t)i < d->strings_num_bits) kmer = (d->W == 1 ? 0 : (kmer << 64)) | get_word64(d, pos + (uint64_t)i);
    }
    return take_bits(kmer, 2 * k);
}

/* endpoints_sequence::locate as used by decoded_offsets::offset_to_id (include/offsets.hpp:138-154):
   string i with endpoints[i] <= x < endpoints[i+1] */
static uint64_t locate(const oracle_index* d, uint64_t x) {
    uint64_t lo = 0, hi = d->num_strings; /* endpoints[lo] <= x < endpoints[hi] */
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (d->endpoints[mid] <= x) lo = mid;
        else hi = mid;
    }
    if (g_touch) g_touch->extra_bytes += 24; /* Elias-Fano locate: high-bits word + two low-bits words */
    return lo;
}

/* ------------------------------------------------------------------------------------------ */
/* lookup                                                                                      */

static void result_init(oracle_result* r, int minimizer_found) { /* include/util.hpp:39-49 */
    r->kmer_id = r->kmer_id_in_string = r->kmer_offset = INVALID_U64;
    r->kmer_orientation = 1;
    r->string_id = r->string_begin = r->string_end = INVALID_U64;
    r->minimizer_found = (uint8_t)minimizer_found;
    memset(r->pad, 0, sizeof(r->pad));
}

enum { SINGLETON = 0, MIDLOAD = 1, HEAVYLOAD = 3 }; /* include/util.hpp:13-17 */

typedef struct {
    uint64_t size, offset, begin;
    int type, valid;
} bucket;

/* minimizers_control_map::lookup (include/minimizers_control_map.hpp:36-39) then
   sparse_and_skew_index::lookup (include/sparse_and_skew_index.hpp:112-137), skew :34-44 */
static bucket ssi_lookup(const oracle_index* d, u128 kmer, uint64_t minimizer) {
    bucket b = {1, 0, 0, SINGLETON, 1};
    uint64_t h[2];
    oracle_city128(&minimizer, 8, d->minimizers.seed, h);
    uint64_t code = cvec_get(&d->codewords, mphf_eval(&d->minimizers, h));
    if ((code & 1) == SINGLETON) {
        b.offset = code >> 1;
        return b;
    }
    if ((code & 3) == MIDLOAD) {
        code >>= 2;
        b.size = (code & ((1u << MIN_L) - 1)) + 2;
        b.begin = d->begin_buckets_of_size[b.size] + (code >> MIN_L) * b.size;
        b.type = MIDLOAD;
        return b;
    }
    b.type = HEAVYLOAD;
    code >>= 2;
    const uint64_t partition_id = code & 7, begin = code >> 3;
    uint64_t key[2] = {(uint64_t)kmer, (uint64_t)(kmer >> 64)};
    oracle_city128(key, d->W == 1 ? 8 : 16, d->skew_f[partition_id].seed, h);
    const uint64_t pos = cvec_get(&d->skew_pos[partition_id], mphf_eval(&d->skew_f[partition_id], h));
    if (begin + pos >= d->heavy_load.size) { /* absent k-mer: arbitrary position (spss.hpp:51-64) */
        b.valid = 0;
        return b;
    }
    b.offset = cvec_get(&d->heavy_load, begin + pos);
    return b;
}

static inline uint64_t bucket_at(const oracle_index* d, const bucket* b, uint64_t i) {
    return b->size == 1 ? b->offset : cvec_get(&d->mid_load, b->begin + i);
}

/* decoded_offsets::offset_to_id, include/offsets.hpp:138-154 */
static void offset_to_id(const oracle_index* d, oracle_result* r) {
    const uint64_t s = locate(d, r->kmer_offset);
    r->string_id = s;
    r->string_begin = d->endpoints[s];
    r->string_end = d->endpoints[s + 1];
    r->kmer_id = r->kmer_offset - s * (d->k - 1);
    r->kmer_id_in_string = r->kmer_offset - r->string_begin;
}

/* _lookup_regular, include/spectrum_preserving_string_set.hpp:213-235 */
static int try_regular(const oracle_index* d, oracle_result* r, uint64_t p, u128 kmer, mini_info mi) {
    if (p < mi.pos_in_kmer) return 0;
    r->kmer_offset = p - mi.pos_in_kmer;
    if (kmer != read_kmer_at(d, d->k, 2 * r->kmer_offset)) return 0;
    offset_to_id(d, r);
    return r->kmer_offset < r->string_end - d->k + 1;
}

/* lookup_regular, include/spectrum_preserving_string_set.hpp:29-73 */
static oracle_result spss_lookup_regular(const oracle_index* d, const bucket* b, u128 kmer, mini_info mi) {
    oracle_result r;
    uint64_t v[1u << MIN_L] = {0};
    if (!b->valid) {
        result_init(&r, 1);
        return r;
    }
    for (uint64_t i = 0; i < b->size; ++i) v[i] = bucket_at(d, b, i);
    if ((uint64_t)read_kmer_at(d, d->m, 2 * v[0]) != mi.minimizer) {
        result_init(&r, b->type != HEAVYLOAD ? 0 : 1);
        return r;
    }
    for (uint64_t i = 0; i < b->size; ++i) {
        result_init(&r, 1);
        if (try_regular(d, &r, v[i], kmer, mi)) return r;
    }
    result_init(&r, 1);
    return r;
}

/* __lookup_canonical, include/spectrum_preserving_string_set.hpp:249-275 */
static int try_canonical(const oracle_index* d, oracle_result* r, uint64_t p, u128 kmer, u128 kmer_rc, uint64_t pos_in_kmer) {
    if (p < pos_in_kmer) return 0;
    r->kmer_offset = p - pos_in_kmer;
    const u128 read = read_kmer_at(d, d->k, 2 * r->kmer_offset);
    if (read != kmer && read != kmer_rc) return 0;
    r->kmer_orientation = read == kmer_rc ? -1 : 1;
    offset_to_id(d, r);
    return r->kmer_offset < r->string_end - d->k + 1;
}

/* lookup_canonical, include/spectrum_preserving_string_set.hpp:75-112 and :237-247 */
static oracle_result spss_lookup_canonical(const oracle_index* d, const bucket* b, u128 kmer, u128 kmer_rc, mini_info mi) {
    oracle_result r;
    uint64_t v[1u << MIN_L] = {0};
    if (!b->valid) {
        result_init(&r, 1);
        return r;
    }
    for (uint64_t i = 0; i < b->size; ++i) v[i] = bucket_at(d, b, i);
    const uint64_t read_mmer = (uint64_t)read_kmer_at(d, d->m, 2 * v[0]);
    if (read_mmer != mi.minimizer) {
        const uint64_t mini_rc = (uint64_t)revcomp(mi.minimizer, d->m, (int)d->W);
        if (read_mmer != mini_rc) {
            result_init(&r, b->type != HEAVYLOAD ? 0 : 1);
            return r;
        }
    }
    for (uint64_t i = 0; i < b->size; ++i) {
        result_init(&r, 1);
        if (try_canonical(d, &r, v[i], kmer, kmer_rc, mi.pos_in_kmer)) return r;
        result_init(&r, 1);
        if (try_canonical(d, &r, v[i], kmer, kmer_rc, d->k - d->m - mi.pos_in_kmer)) return r;
    }
    result_init(&r, 1);
    return r;
}

/* dictionary::lookup_regular(kmer, mini_info), src/dictionary.cpp:13-22 */
static oracle_result dict_lookup_regular(const oracle_index* d, u128 kmer, mini_info mi) {
    const bucket b = ssi_lookup(d, kmer, mi.minimizer);
    return spss_lookup_regular(d, &b, kmer, mi);
}

/* dictionary::lookup_canonical(kmer, kmer_rc, mini_info), src/dictionary.cpp:44-56 */
static oracle_result dict_lookup_canonical3(const oracle_index* d, u128 kmer, u128 kmer_rc, mini_info mi) {
    const u128 canon = kmer < kmer_rc ? kmer : kmer_rc;
    const bucket b = ssi_lookup(d, canon, mi.minimizer);
    return spss_lookup_canonical(d, &b, kmer, kmer_rc, mi);
}

/* dictionary::lookup(Kmer, bool), src/dictionary.cpp:64-78 with lookup_canonical(Kmer) :24-42 */
static oracle_result dict_lookup(const oracle_index* d, u128 kmer, int check_rc) {
    if (d->canonical) {
        const u128 rc = revcomp(kmer, d->k, (int)d->W);
        const mini_info mi = compute_minimizer(kmer, d->k, d->m, d->hash_magic);
        const mini_info mr = compute_minimizer(rc, d->k, d->m, d->hash_magic);
        if (mi.minimizer < mr.minimizer) return dict_lookup_canonical3(d, kmer, rc, mi);
        if (mr.minimizer < mi.minimizer) return dict_lookup_canonical3(d, kmer, rc, mr);
        oracle_result r = dict_lookup_canonical3(d, kmer, rc, mi);
        if (r.kmer_id == INVALID_U64) r = dict_lookup_canonical3(d, kmer, rc, mr);
        return r;
    }
    oracle_result r = dict_lookup_regular(d, kmer, compute_minimizer(kmer, d->k, d->m, d->hash_magic));
    if (check_rc && r.kmer_id == INVALID_U64) {
        const u128 rc = revcomp(kmer, d->k, (int)d->W);
        r = dict_lookup_regular(d, rc, compute_minimizer(rc, d->k, d->m, d->hash_magic));
        r.kmer_orientation = -1;
    }
    return r;
}

static inline u128 load_packed(const oracle_index* d, const uint64_t* q, uint64_t i) {
    u128 x = q[i * d->W];
    if (d->W == 2) x |= (u128)q[i * 2 + 1] << 64;
    return take_bits(x, 2 * d->k);
}

void oracle_lookup_packed(const oracle_index* d, const uint64_t* kmers, uint64_t n, int check_rc, oracle_result* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = dict_lookup(d, load_packed(d, kmers, i), check_rc);
}

void oracle_lookup_ascii(const oracle_index* d, const char* kmers, uint64_t n, int check_rc, oracle_result* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = dict_lookup(d, string_to_kmer(kmers + i * d->k, d->k), check_rc);
}

typedef struct {
    const oracle_index* d;
    const uint64_t* kmers;
    uint64_t begin, end;
    int check_rc;
    uint64_t* ids;
} ids_job;

static void* ids_worker(void* arg) {
    ids_job* j = (ids_job*)arg;
    for (uint64_t i = j->begin; i < j->end; ++i) j->ids[i] = dict_lookup(j->d, load_packed(j->d, j->kmers, i), j->check_rc).kmer_id;
    return NULL;
}

void oracle_lookup_ids(const oracle_index* d, const uint64_t* kmers, uint64_t n, int check_rc, uint64_t* ids, int num_threads) {
    if (num_threads < 1) num_threads = 1;
    if (num_threads == 1 || n < (uint64_t)num_threads) {
        ids_job j = {d, kmers, 0, n, check_rc, ids};
        ids_worker(&j);
        return;
    }
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)num_threads);
    ids_job* jobs = (ids_job*)malloc(sizeof(ids_job) * (size_t)num_threads);
    const uint64_t chunk = (n + (uint64_t)num_threads - 1) / (uint64_t)num_threads;
    for (int t = 0; t < num_threads; ++t) {
        uint64_t b = (uint64_t)t * chunk, e = b + chunk;
        if (b > n) b = n;
        if (e > n) e = n;
        jobs[t] = (ids_job){d, kmers, b, e, check_rc, ids};
        pthread_create(&th[t], NULL, ids_worker, &jobs[t]);
    }
    for (int t = 0; t < num_threads; ++t) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

uint64_t oracle_count_bytes(const oracle_index* d, const uint64_t* kmers, uint64_t n, int check_rc) {
    touch_ctx ctx;
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        ctx.n = 0;
        ctx.extra_bytes = 0;
        g_touch = &ctx;
        (void)dict_lookup(d, load_packed(d, kmers, i), check_rc);
        g_touch = NULL;
        total += 8ull * (uint64_t)ctx.n + ctx.extra_bytes + 8ull * d->W /* query in */ + 8 /* id out */;
    }
    return total;
}

/* spss::access, include/spectrum_preserving_string_set.hpp:114-118 with id_to_offset (include/offsets.hpp:41-65) */
/* dictionary::weight (src/dictionary.cpp:96-100) -> weights::weight (include/weights.hpp:147-152):
   prev_leq(kmer_id) over the cumulative interval lengths, then the interval's value. Returns 0 and sets
   *ok = 0 when the dictionary stores no weights (test/check_from_file.hpp:234-237) or the id is out of range. */
uint64_t oracle_weight(const oracle_index* d, uint64_t kmer_id, int* ok) {
    *ok = 0;
    if (d->num_weight_intervals == 0 || kmer_id >= d->num_kmers) return 0;
    uint64_t lo = 0, hi = d->num_weight_intervals - 1; /* largest i with weight_starts[i] <= kmer_id */
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (d->weight_starts[mid] <= kmer_id) lo = mid;
        else hi = mid - 1;
    }
    *ok = 1;
    return d->weight_values[lo];
}

int oracle_weights(const oracle_index* d, const uint64_t* kmer_ids, uint64_t n, uint64_t* out) {
    for (uint64_t i = 0; i < n; ++i) {
        int ok;
        out[i] = oracle_weight(d, kmer_ids[i], &ok);
        if (!ok) return 0;
    }
    return 1;
}

void oracle_access(const oracle_index* d, uint64_t kmer_id, char* out) {
    uint64_t lo = 0, hi = d->num_strings - 1;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (d->endpoints[mid] - mid * (d->k - 1) <= kmer_id) lo = mid;
        else hi = mid - 1;
    }
    const uint64_t offset = kmer_id + lo * (d->k - 1);
    u128 x = read_kmer_at(d, d->k, 2 * offset);
    for (uint32_t i = 0; i < d->k; ++i, x >>= 2) out[i] = "ACTG"[(unsigned)(x & 3)]; /* include/kmer.hpp:118 */
}

/* ------------------------------------------------------------------------------------------ */
/* streaming query (include/streaming_query.hpp)                                               */

typedef struct {
    const oracle_index* d;
    oracle_result res;
    int start;
    u128 kmer, kmer_rc;
    mini_info curr, prev, curr_rc, prev_rc;
    uint64_t it_pos; /* base offset of the k-mer the string iterator currently points at */
    uint64_t remaining_string_bases;
    uint64_t num_searches, num_extensions, num_invalid, num_negative;
} sq_state;

static void sq_reset(sq_state* s) { /* :48-54 */
    s->start = 1;
    s->remaining_string_bases = 0;
    result_init(&s->res, 1);
}

static char complement_char(char c) { /* canonicalize_basepair_reverse_map, include/kmer.hpp:233-243 */
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default: return 0;
    }
}

static void sq_seed(sq_state* s) { /* :144-197 */
    const oracle_index* d = s->d;
    s->remaining_string_bases = 0;
    if (s->curr.minimizer == s->prev.minimizer && s->curr_rc.minimizer == s->prev_rc.minimizer && !s->res.minimizer_found) {
        s->num_negative += 1;
        return;
    }
    if (d->canonical) {
        if (s->curr.minimizer < s->curr_rc.minimizer) {
            s->res = dict_lookup_canonical3(d, s->kmer, s->kmer_rc, s->curr);
        } else if (s->curr_rc.minimizer < s->curr.minimizer) {
            s->res = dict_lookup_canonical3(d, s->kmer, s->kmer_rc, s->curr_rc);
        } else {
            s->res = dict_lookup_canonical3(d, s->kmer, s->kmer_rc, s->curr);
            if (s->res.kmer_id == INVALID_U64) s->res = dict_lookup_canonical3(d, s->kmer, s->kmer_rc, s->curr_rc);
        }
    } else {
        s->res = dict_lookup_regular(d, s->kmer, s->curr);
        const int found_fwd = s->res.minimizer_found;
        if (s->res.kmer_id == INVALID_U64) {
            s->res = dict_lookup_regular(d, s->kmer_rc, s->curr_rc);
            s->res.kmer_orientation = -1;
            s->res.minimizer_found = (uint8_t)(s->res.minimizer_found || found_fwd);
        }
    }
    if (s->res.kmer_id == INVALID_U64) {
        s->num_negative += 1;
        return;
    }
    s->num_searches += 1;
    /* position the string iterator on the matched k-mer (:189-196); the reference keeps a bit
       position and a direction, here: the base offset of the matched k-mer */
    s->it_pos = s->res.kmer_id + s->res.string_id * (d->k - 1);
    s->remaining_string_bases = (s->res.string_end - s->res.string_begin - d->k) - s->res.kmer_id_in_string;
    if (s->res.kmer_orientation == -1) s->remaining_string_bases = s->res.kmer_id_in_string;
}

static oracle_result sq_lookup(sq_state* s, const char* kmer_chars) { /* :56-109 */
    const oracle_index* d = s->d;
    const uint32_t k = d->k;
    int valid;
    if (s->start) {
        valid = 1;
        for (uint32_t i = 0; i < k; ++i) valid = valid && oracle_is_valid_base(kmer_chars[i]);
    } else {
        valid = oracle_is_valid_base(kmer_chars[k - 1]);
    }
    if (!valid) {
        s->num_invalid += 1;
        sq_reset(s);
        return s->res;
    }
    if (!s->start) {
        s->kmer >>= 2;
        s->kmer |= (u128)char_to_uint(kmer_chars[k - 1]) << (2 * (k - 1));
        s->kmer_rc <<= 2;
        s->kmer_rc |= char_to_uint(complement_char(kmer_chars[k - 1]));
        s->kmer_rc = take_bits(s->kmer_rc, 2 * k);
    } else {
        s->kmer = string_to_kmer(kmer_chars, k);
        s->kmer_rc = revcomp(s->kmer, k, (int)d->W);
    }
    /* minimizer_iterator(.next) == compute_minimizer, asserted at include/minimizer_iterator.hpp:56-57,138-139 */
    s->curr = compute_minimizer(s->kmer, k, d->m, d->hash_magic);
    s->curr_rc = compute_minimizer(s->kmer_rc, k, d->m, d->hash_magic);
    if (s->remaining_string_bases == 0) {
        sq_seed(s);
    } else {
        /* next k-mer of the string in the direction of the match (kmer_iterator::next/next_reverse) */
        const uint64_t expected_pos = s->res.kmer_orientation == 1 ? s->it_pos + 1 : s->it_pos - 1;
        const u128 expected = read_kmer_at(d, k, 2 * expected_pos);
        s->it_pos = expected_pos;
        if (expected == s->kmer || expected == s->kmer_rc) {
            s->num_extensions += 1;
            s->res.kmer_id += (uint64_t)s->res.kmer_orientation;
            s->res.kmer_id_in_string += (uint64_t)s->res.kmer_orientation;
            s->remaining_string_bases -= 1;
        } else {
            sq_seed(s);
        }
    }
    s->prev = s->curr;
    s->prev_rc = s->curr_rc;
    s->start = 0;
    return s->res;
}

static void sq_init(sq_state* s, const oracle_index* d) {
    memset(s, 0, sizeof(*s));
    s->d = d;
    s->prev.minimizer = s->prev_rc.minimizer = s->curr.minimizer = s->curr_rc.minimizer = INVALID_U64;
    sq_reset(s);
}

/* src/query.cpp:78-108: reset per read, one lookup per k-mer */
void oracle_streaming_query(const oracle_index* d, const char* bases, const uint64_t* off, uint64_t num_reads, uint64_t report[6]) {
    sq_state s;
    sq_init(&s, d);
    uint64_t num_kmers = 0;
    for (uint64_t r = 0; r < num_reads; ++r) {
        const char* read = bases + off[r];
        const uint64_t len = off[r + 1] - off[r];
        sq_reset(&s);
        if (len < d->k) continue;
        num_kmers += len - d->k + 1;
        for (uint64_t i = 0; i + d->k <= len; ++i) (void)sq_lookup(&s, read + i);
    }
    report[0] = num_kmers;
    report[1] = s.num_searches + s.num_extensions;
    report[2] = s.num_negative;
    report[3] = s.num_invalid;
    report[4] = s.num_searches;
    report[5] = s.num_extensions;
}

/* Algorithmic bytes of a streaming query under the counting rule of SURVEY.md section 8(d), applied k-mer by k-mer to what the
   REFERENCE's state machine does: 8 bytes per distinct 64-bit index word it dereferences for that k-mer -- the lookups of a seed()
   that is not cut short by the unchanged-minimizer test (:150-157), the strings' next k-mer of an extension (:86-100) -- plus one byte
   per base of the reads. (No per-k-mer result is counted: the query reports six counters.) */
uint64_t oracle_streaming_count_bytes(const oracle_index* d, const char* bases, const uint64_t* off, uint64_t num_reads) {
    sq_state s;
    sq_init(&s, d);
    touch_ctx ctx;
    uint64_t total = 0;
    for (uint64_t r = 0; r < num_reads; ++r) {
        const char* read = bases + off[r];
        const uint64_t len = off[r + 1] - off[r];
        sq_reset(&s);
        total += len;
        if (len < d->k) continue;
        for (uint64_t i = 0; i + d->k <= len; ++i) {
            ctx.n = 0;
            ctx.extra_bytes = 0;
            g_touch = &ctx;
            (void)sq_lookup(&s, read + i);
            g_touch = NULL;
            total += 8ull * (uint64_t)ctx.n + ctx.extra_bytes;
        }
    }
    return total;
}

void oracle_streaming_read(const oracle_index* d, const char* read, uint64_t len, oracle_result* out) {
    sq_state s;
    sq_init(&s, d);
    for (uint64_t i = 0; i + d->k <= len; ++i) out[i] = sq_lookup(&s, read + i);
}

/* ------------------------------------------------------------------------------------------ */
/* primitives for golden-vector tests                                                          */

void oracle_encode_kmer(const char* s, uint32_t k, uint64_t out[2]) {
    const u128 x = string_to_kmer(s, k);
    out[0] = (uint64_t)x;
    out[1] = (uint64_t)(x >> 64);
}

void oracle_revcomp(const uint64_t in[2], uint32_t k, int words, uint64_t out[2]) {
    u128 x = in[0];
    if (words == 2) x |= (u128)in[1] << 64;
    const u128 r = revcomp(x, k, words);
    out[0] = (uint64_t)r;
    out[1] = (uint64_t)(r >> 64);
}

void oracle_minimizer(const uint64_t kmer[2], uint32_t k, uint32_t m, uint64_t magic, int words, uint64_t* value, uint64_t* pos) {
    u128 x = kmer[0];
    if (words == 2) x |= (u128)kmer[1] << 64;
    const mini_info mi = compute_minimizer(x, k, m, magic);
    *value = mi.minimizer;
    *pos = mi.pos_in_kmer;
}

/* ------------------------------------------------------------------------------------------ */
/* index file parser (format: sshash_amd/csrc/index.cpp, "(de)serialisation")                   */

typedef struct {
    FILE* f;
    int ok;
} rd;

static void rd_raw(rd* r, void* p, size_t n) {
    if (r->ok && n && fread(p, 1, n, r->f) != n) r->ok = 0;
}
static uint64_t rd_u64(rd* r) {
    uint64_t v = 0;
    rd_raw(r, &v, 8);
    return v;
}
static void* rd_vec(rd* r, size_t elem, uint64_t* count) {
    const uint64_t n = rd_u64(r);
    *count = n;
    if (!r->ok || n > ((uint64_t)1 << 40)) {
        r->ok = 0;
        return NULL;
    }
    void* p = malloc(n * elem + 8);
    if (!p) {
        r->ok = 0;
        return NULL;
    }
    rd_raw(r, p, n * elem);
    uint64_t pad;
    rd_raw(r, &pad, (8 - (n * elem) % 8) % 8);
    return p;
}
static void rd_cvec(rd* r, cvec* v) {
    v->size = rd_u64(r);
    v->width = (uint32_t)rd_u64(r);
    v->words = (uint64_t*)rd_vec(r, 8, &v->num_words);
}
static void rd_mphf(rd* r, mphf* f) {
    f->seed = rd_u64(r);
    f->num_keys = rd_u64(r);
    f->pilot_width = (uint32_t)rd_u64(r);
    f->parts = (mphf_part*)rd_vec(r, sizeof(mphf_part), &f->num_parts);
    f->pilots = (uint64_t*)rd_vec(r, 8, &f->num_pilot_words);
    f->free_slots = (uint32_t*)rd_vec(r, 4, &f->num_free);
}

int oracle_load(const char* filename, oracle_index** out, char* err, int err_len) {
    *out = NULL;
    FILE* f = fopen(filename, "rb");
    if (!f) {
        snprintf(err, (size_t)err_len, "error in opening the file '%s'", filename);
        return 2;
    }
    oracle_index* d = (oracle_index*)calloc(1, sizeof(oracle_index));
    rd r = {f, 1};
    char magic[8];
    uint8_t hdr[4];
    uint32_t kms[3];
    rd_raw(&r, magic, 8);
    if (!r.ok || memcmp(magic, "SSHAMD\x03\x00", 8) != 0) {
        snprintf(err, (size_t)err_len, "not an sshash_amd index file");
        fclose(f);
        free(d);
        return 3;
    }
    rd_raw(&r, hdr, 4);
    if (r.ok && hdr[0] != 5) { /* include/util.hpp:191-195 */
        snprintf(err, (size_t)err_len, "MAJOR index version mismatch: SSHash index needs rebuilding");
        fclose(f);
        free(d);
        return 4;
    }
    d->canonical = hdr[3];
    rd_raw(&r, kms, 12);
    d->k = kms[0];
    d->m = kms[1];
    d->skew_parts = kms[2];
    d->W = d->k <= 31 ? 1 : 2;
    d->num_kmers = rd_u64(&r);
    d->num_strings = rd_u64(&r);
    d->num_bases = rd_u64(&r);
    d->hash_magic = rd_u64(&r);
    d->build_seed = rd_u64(&r);
    d->strings_num_bits = rd_u64(&r);
    {
        const uint64_t sh = rd_u64(&r); /* num_shards | shard_id << 32: which minimizers this index holds */
        d->num_shards = (uint32_t)sh;
        d->shard_id = (uint32_t)(sh >> 32);
    }
    uint64_t cnt;
    d->strings = (uint64_t*)rd_vec(&r, 8, &d->strings_words);
    d->endpoints = (uint64_t*)rd_vec(&r, 8, &cnt);
    rd_mphf(&r, &d->minimizers);
    rd_cvec(&r, &d->codewords);
    d->begin_buckets_of_size = (uint32_t*)rd_vec(&r, 4, &cnt);
    rd_cvec(&r, &d->mid_load);
    for (uint32_t p = 0; p < d->skew_parts && p < 8; ++p) {
        rd_mphf(&r, &d->skew_f[p]);
        rd_cvec(&r, &d->skew_pos[p]);
    }
    rd_cvec(&r, &d->heavy_load);
    d->weight_starts = (uint64_t*)rd_vec(&r, 8, &d->num_weight_intervals);
    d->weight_values = (uint64_t*)rd_vec(&r, 8, &cnt);
    fclose(f);
    if (!r.ok) {
        snprintf(err, (size_t)err_len, "index file truncated or corrupt");
        oracle_free(d);
        return 3;
    }
    *out = d;
    return 0;
}

static void free_mphf(mphf* f) {
    free(f->parts);
    free(f->pilots);
    free(f->free_slots);
}

void oracle_free(oracle_index* d) {
    if (!d) return;
    free(d->strings);
    free(d->endpoints);
    free_mphf(&d->minimizers);
    free(d->codewords.words);
    free(d->begin_buckets_of_size);
    free(d->mid_load.words);
    for (int p = 0; p < 8; ++p) {
        free_mphf(&d->skew_f[p]);
        free(d->skew_pos[p].words);
    }
    free(d->heavy_load.words);
    free(d->weight_starts);
    free(d->weight_values);
    free(d);
}

void oracle_get_info(const oracle_index* d, oracle_info* info) {
    info->k = d->k;
    info->m = d->m;
    info->canonical = d->canonical;
    info->words_per_kmer = d->W;
    info->num_kmers = d->num_kmers;
    info->num_strings = d->num_strings;
    info->num_bases = d->num_bases;
    info->num_minimizers = d->codewords.size;
    info->hash_magic = d->hash_magic;
}
