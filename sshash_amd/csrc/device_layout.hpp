// device_layout.hpp -- how the dictionary lives in MI355X HBM.
//
// The host index keeps the reference's components (index.hpp). On upload they are re-laid
// for what bounds this workload on the GPU: the number of *dependent random 64-byte sector
// fetches* per query, not bytes. Two things differ from the host layout:
//
//  (1) `strings` and the string endpoints are fused so that ONE aligned read yields the k-mer to
//      compare, the id of the string it lies in (rank + popcount) and whether it runs across a
//      string boundary -- the information the reference obtains from `read_kmer_at` plus an
//      Elias-Fano `locate` (include/util.hpp:248-257, include/offsets.hpp:138-154: 1 + ~3
//      dependent misses):
//        k <= 31: 32-byte, 32-byte-aligned *atoms*, one per 32 bases, each holding 64 bases
//                 (its own 32 and a copy of the next 32):
//                     { u64 bases[2], u64 marks, u32 rank, u32 spare }
//                 so a k-mer starting at base o lies wholly inside atom o/32: exactly one DRAM
//                 atom per window (a 32-byte window over 16-byte granules straddles two atoms
//                 half of the time);
//        k <= 63: 16-byte granules { u32 rank, u32 marks, u64 bases } per 32 bases; a window reads
//                 three consecutive granules.
//      bases : the same 2-bit codes, base i of the block in bits [2i,2i+1]
//      marks : bit i set iff a string begins at that base (the final end offset is marked too)
//      rank  : number of marks at bases before the block
//      Costs 8 (resp. 4) bits/base instead of ~2.1; HBM capacity (288 GB) is not the constraint.
//  (2) the endpoints themselves are kept as a plain u64 array, touched only when the caller
//      asks for the full lookup_result (string_begin/string_end).
//
//  (3) control codewords are widened to one aligned u64 per minimizer id:
//          bits [0, cw_width)   the reference's control codeword, unchanged
//          bits [cw_width, 64)  a fingerprint of the bucket's minimizer (top bits of a multiplicative
//                               hash; canonical indexes fingerprint min(minimizer, its reverse complement))
//      A query whose minimizer is not the bucket's (every negative query, and the first probe of a
//      reverse-complemented positive) is rejected right after the codeword read, without touching
//      the strings: the reference's own minimizer check (spectrum_preserving_string_set.hpp:46-65)
//      answered from the codeword's sector. Equal fingerprints still go through the exact check, so
//      results are unchanged; only the number of sector fetches per miss drops (3.3 -> 2.1).
//      The fingerprints are derived on the GPU at upload time from the strings themselves.
//      A replica that holds the super-k-mer table (5) keeps the codewords bit-packed as the host index has them
//      (dict_view::cw_packed) and builds no directory (4): with the table only the ~0.05 % deferred queries (and the
//      `minimizer_found` byte of a miss) ever come this way, and 3.2 + 8.6 GB of a 50 GB human-scale replica served them.
//
//  (4) a *minimizer directory*: the minimizer -> codeword map once more, as a fingerprinted bucket
//      table whose buckets are exactly one 32-byte DRAM atom: 4 x u64 entries,
//          entry = codeword (40 bits) | fingerprint (16 bits) << 40 | meta << 56
//          meta: bit 0 = slot valid; bit 7 of entry 0 = bucket overflowed
//      bucket = hash(minimizer) range-reduced, average load 1.0 keys. What bounds this workload is
//      the number of 32-byte atoms fetched from HBM per lookup (~50 G random atoms/s):
//      MPHF + codeword array cost two dependent atoms per probe (pilot, then codeword); the
//      directory answers the same question with ONE. It is an accelerator over the MPHF, not a
//      replacement: every key stays reachable through the MPHF; a bucket whose keys did not all fit
//      (Poisson tail, ~0.4 % of buckets) or that held two keys with equal fingerprints carries the
//      overflow flag, and probes that are not settled there fall back to the MPHF path. A key absent
//      from a non-overflowed bucket is absent from the dictionary. Used by the ids-only / is_member
//      / streaming kernels; the full-result kernel keeps the MPHF path because for an absent
//      minimizer the reference's `minimizer_found` flag depends on which (arbitrary) bucket the
//      MPHF lands on. Built on the GPU at upload; SSHASH_AMD_DIRECTORY=0 disables it.
//
//  (5) a *super-k-mer table*, built on the GPU at upload from the strings alone (described for k <= 31;
//      for k <= 63 a slot is 64 bytes -- the same 16-byte head, then 128 bases, then 16 spare bytes -- and a bucket two 64-byte
//      lines; slot 1's line is fetched only by the probes whose key's fingerprint matches slot 1's, which slot 0 keeps in its
//      spare bytes). Structures
//      (1)-(4) answer a positive lookup with two dependent random reads per probe (minimizer ->
//      position, then the strings) and a regular index probes both strands (src/dictionary.cpp:70-75);
//      what bounds the batch is the number of such reads. The table answers most lookups with ONE:
//        key    a strand-symmetric minimizer of the k-mer (sk_key below): the m-mer with the smallest hash
//               over both strands (equal minima = tie, left to the structures above) -- one key for both
//               strands, whatever the dictionary's own minimizer flavour. Built and probed with the same
//               function, the table need not follow the reference's minimizer hash: it uses a 32-bit one;
//        bucket one 64-byte line = the unit the memory system fetches (TCC_EA0_RDREQ is always a 64-byte request,
//               profiles/r02/tlb_probe_counters.txt) = TWO 32-byte slots. A key lives in one of SK_CHOICES = 5
//               hashed buckets, in the first of them that had a free slot when it was placed; 2.5 slots per item
//               (load factor 0.4). The four lanes of a quad fetch the line of one of them together -- 16 bytes each,
//               ONE load instruction, transposed through LDS (lookup_device.hpp) -- so that the memory pipeline sees
//               one request and ONE address translation per lookup: with one lane issuing the 16-byte loads of its
//               own slot every load is a separate UTCL1 miss once the table outgrows the ~2 GiB the per-CU
//               translation cache covers, and the translation-request rate (75 G/s chip-wide), not DRAM, is what
//               capped round 1's table at 38 G reads/s (HISTORY.md);
//        slot   32 bytes:
//                 d0  bit0 valid | bit1 marker | bit2 strand | bits 3-7 go-on flags of the BUCKET, one per choice
//                     (kept in slot 0 only) | bits 8-13 left | bits 14-19 right | bit 20 (slot 0 only) slot 1 is in use |
//                     bits 21-31 (slot 0 only) filter over the fingerprints of the keys that went on from their FIRST choice
//                 d1  string id (marker: number of occurrences of the key)
//                 d2,d3  position of the key occurrence (40 bits) | fingerprint of the key << 40
//                 d4-d7  the 64 bases starting k-m bases before the occurrence, i.e. every k-mer of the super-k-mer
//               An *inline* slot holds one occurrence of its key: the super-k-mer around it, how far it may extend
//               inside its string (left/right), and the string id, so a lookup that finds its k-mer there ends
//               there. A key with up to SK_INLINE_MAX = 4 occurrences in the strings (one for ~96 % of the keys)
//               holds one inline slot per occurrence, spread along its bucket sequence (two per bucket). A key with
//               more (a *heavy* key: repeats, low-complexity sequence) holds ONE *marker* slot, and every k-mer of
//               every one of its super-k-mers is entered a second time under a key of its own -- a hash of the
//               canonical k-mer (sk_kmer_key) -- as a copy of its super-k-mer's inline slot: a probe that meets its
//               key's marker switches to the k-mer's own bucket sequence and finds the k-mer after one more read,
//               however many thousand occurrences the key has (what the reference's skew index does for its heavy
//               buckets, include/sparse_and_skew_index.hpp:34-44, with the table's own machinery).
//               The k-mers' copies live in a range of buckets of their own behind the keys' (sk_view::kmer_buckets), packed
//               tighter: only the probes that met a marker pay for its second choices. At k <= 63 an entry there is not a copy
//               of the 64-byte slot but 32 bytes -- d0 valid + (entry 0) the bucket's flags and filter | d1 string id |
//               d2,d3 position of the K-MER | fingerprint << 40 | d4-d7 the k-mer as the strings spell it -- two to a bucket,
//               one 64-byte line: the region takes half the bytes and a probe never needs a second line.
//        flags  go-on flag c of a bucket says "a key whose c-th choice is this bucket lives further along its
//               sequence"; a probe that finds neither its k-mer nor that flag is a final miss -- negative
//               queries end after ~1.1 line reads. The last choice's flag (a key or k-mer that found no slot at all)
//               and ties send the query to the complete path through (3)/(4).
//      Ids are positions in the strings, so results are identical to the reference's; the table only
//      changes how many reads it takes to find the position. SSHASH_AMD_SKTABLE=0 disables it.
//      The table is the largest structure (~14 bytes per k-mer at k = 31, m = 21); with several GPUs it can be
//      partitioned by key (sk_owner): each replica then builds the slots of its own keys only, queries are
//      routed to the owner of their key (one message per query, sharded.py), and a replica that meets a key
//      it does not own simply takes the complete path -- every replica stays correct on its own.
//
// The remaining packed vectors (bucket offset lists, pilots, skew positions) keep their bit-packed
// form: one 8-byte read, sometimes two adjacent ones.
#pragma once

#include "mphf.hpp"

namespace sshash_amd {

struct alignas(16) granule {
    uint32_t rank;
    uint32_t marks;
    uint64_t bases;
};
static_assert(sizeof(granule) == 16, "granule layout");

struct alignas(32) atom32 {  // k <= 31
    uint64_t bases[2];
    uint64_t marks;
    uint32_t rank;
    uint32_t spare;
};
static_assert(sizeof(atom32) == 32, "atom layout");

constexpr uint32_t GRANULE_BASES = 32;
constexpr uint32_t GRANULE_PAD = 4;  // zero granules after the last real one

constexpr uint64_t FINGERPRINT_MUL = 0x9E3779B97F4A7C15ULL;

/* fingerprint of a minimizer, comparable with (entry >> cw_width) */
SSH_HD uint64_t minimizer_fingerprint(uint64_t minimizer, uint32_t m, bool canonical, uint32_t cw_width) {
    uint64_t key = minimizer;
    if (canonical) {
        const uint64_t rc = mmer_revcomp(minimizer, m);
        key = rc < key ? rc : key;
    }
    return cw_width >= 64 ? 0 : (key * FINGERPRINT_MUL) >> cw_width;
}

constexpr uint32_t DIR_SLOTS = 4;         // entries per 32-byte bucket
constexpr uint32_t DIR_CODE_BITS = 40;    // widest control codeword the directory can hold
constexpr double DIR_LOAD = 1.0;          // average keys per bucket (1.5 until round 3: 1.9 % of the buckets overflowed and every probe they left
                                          // open went to the complete path -- 0.64 of the directory path's 7.4 ms per 10^8 queries; at 1.0: 0.4 %)

struct directory_view {
    uint64_t const* buckets;  // 4 words per bucket
    uint32_t num_buckets;
    uint32_t enabled;
};

/* hash of a minimizer for the directory: high half picks the bucket, low 16 bits are the fingerprint */
SSH_HD uint64_t directory_hash(uint64_t minimizer) {
    uint64_t x = minimizer * 0xFF51AFD7ED558CCDULL;
    x ^= x >> 32;
    x *= 0xC4CEB9FE1A85EC53ULL;
    x ^= x >> 29;
    return x;
}
SSH_HD uint32_t directory_bucket(uint64_t h, uint32_t num_buckets) { return mulhi32(uint32_t(h >> 32), num_buckets); }
SSH_HD uint32_t directory_fingerprint(uint64_t h) { return uint32_t(h) & 0xFFFFu; }
SSH_HD uint64_t directory_entry(uint64_t code, uint32_t fp) { return code | (uint64_t(fp) << 40) | (uint64_t(1) << 56); }

/* ---- super-k-mer table (5) ---- */
constexpr uint32_t SK_VALID = 1u, SK_MARKER = 2u, SK_STRAND = 4u;
constexpr uint32_t SK_GO_ON = 8u;             // << c: a key whose choice c is this bucket lives at a later choice ...
constexpr uint32_t SK_CHOICES = 5;            // ... or, for the last choice, in no slot at all (flags: bits 3-7 of slot 0)
constexpr uint32_t SK_BUCKET_SLOTS = 2;       // slots per bucket: one 64-byte line at k <= 31
constexpr uint32_t SK_LEFT_SHIFT = 8, SK_RIGHT_SHIFT = 14;
constexpr uint32_t SK_SECOND_FINGERPRINT_WORD = 12;  // k <= 63: 32-bit word of slot 0 (its last 16 bytes are spare) holding slot 1's fingerprint
constexpr uint32_t SK_SECOND_USED = 1u << 20;  // in slot 0: slot 1 of the bucket is in use (k <= 63: worth fetching its line)
/* Bits 21-31 of slot 0: WHICH keys went on from their first choice -- bit sk_filter_index(fingerprint) is set by every item that
   found this bucket, its first choice, full. The first go-on flag alone sends every query that misses in such a bucket (5 % of
   the buckets at load factor 0.4) to the resume pass, negative ones included; with the filter a query goes on only if a key
   with its own filter index did: 1/11 of those negatives. */
constexpr uint32_t SK_FILTER_SHIFT = 21, SK_FILTER_BITS = 11;
static_assert(SK_FILTER_SHIFT + SK_FILTER_BITS == 32 && (1u << SK_FILTER_SHIFT) > SK_SECOND_USED, "filter bits are the top of the word");
SSH_HD uint32_t sk_filter_index(uint32_t fingerprint24) { return (fingerprint24 * SK_FILTER_BITS) >> 24; }
static_assert((SK_GO_ON << (SK_CHOICES - 1)) < (1u << SK_LEFT_SHIFT), "go-on flags must stay below the extent fields");
static_assert(SK_CHOICES == 5, "sk_hash and sk_choice spell out five choices");
constexpr uint32_t SK_INLINE_MAX = 4;         // a key with up to this many occurrences holds one inline slot per occurrence;
                                              // a heavier key holds a marker and its k-mers are keyed one by one
constexpr double SK_SLOTS_PER_KEY = 2.5;  // slots per item: load factor 0.4 (HISTORY.md: 1.6 ... 4.0 measured)
/* k <= 63: 3.0 (load factor 0.33). A bucket there is two lines, an item in slot 1 or past its first bucket costs a second one, and since the
   table's keys are 31 bases long (sk_table_m) the k-mers' region is small enough to pay for it: same-box, four alternating rounds
   (profiles/r04/slots_per_key_c4_table_key_31.txt) 2.5 -> 3.0 -> 3.5 slots = 30.0 -> 31.4 -> 32.0 G lookups/s (medians) for 13.35 -> 15.17 ->
   16.99 B/k-mer; with the 25-base key 3.0 bought 0.5 % (slots_per_key_c4.txt) */
constexpr double SK_SLOTS_PER_KEY_WIDE = 3.0;
/* the same for the region of the heavy keys' k-mers (sk_view::kmer_buckets): fuller, since only the probes that met a marker pay for it.
   Same-box sweep (profiles/r03/kmer_region_load_ab.txt): C3 2.5 -> 2.0 -> 1.75 -> 1.5 slots per k-mer = 47.9 -> 45.7 -> 44.6 -> 43.5 GB and
   37.8 -> 37.6 -> 37.25 -> 36.6 G lookups/s; C4 (k = 63, 64-byte slots, 9.9 % of the k-mers under heavy keys) 66.7 -> 58.0 -> 53.6 ->
   49.3 GB with every rate inside the box's +-3 % run-to-run spread (the first pass is bound by instructions there, not by lines). */
/* k <= 63, round 4's last sweep (31-base table key, 3.0 slots per item in the keys' region; profiles/r04/kmer_region_places_c4_last_defaults.txt, a box that
   repeats within 0.2 %): 1.75 -> 2.5 places per k-mer = 33.78 / 33.72 / 33.77 -> 34.42 / 34.37 / 34.43 G lookups/s (+1.9 %) for 15.2 -> 16.6 B/k-mer; the
   streaming query does not see it (37.15 -> 37.2). With a third fewer k-mers there the region is cheap to loosen, and every lookup that gets to it has already
   paid for two lines. */
constexpr double SK_SLOTS_PER_KMER_NARROW = 1.75, SK_SLOTS_PER_KMER_WIDE = 2.5;  // k <= 31 (2.0 until the entries there became three to a line: profiles/r04/kmer_region_places_sweep.txt), k <= 63
/* k <= 31 (round 4): a bucket of the k-mers' region is one 64-byte line of THREE entries instead of two copies of 32-byte slots --
   a k-mer entered under its own key needs the k-mer, where it lies and in which string, not its super-k-mer's 64 bases:
     dword 0          bits 0-2 entry e in use | bits 3-7 the bucket's go-on flags | bits 21-31 the first-choice filter (as slot 0)
     dwords 1+5e ..   k-mer (as the strings spell it) lo, hi | position lo | string id | position bits 32-39
   134.9 M k-mers of C3's heavy keys: 8.6 GB at one slot each, 5.8 GB so. */
constexpr uint32_t SK_KMER_ENTRIES_NARROW = 3, SK_KMER_ENTRY_WORDS = 5;
/* why a replica was given no table (sshash_device_stats) */
constexpr uint32_t SK_ABSENT_DISABLED = 1, SK_ABSENT_MINIMIZER_SHARD = 2, SK_ABSENT_TOO_MANY_BASES = 3, SK_ABSENT_TOO_MANY_ITEMS = 4,
                   SK_ABSENT_NO_MEMORY = 5;

struct sk_view {
    void const* slots;    // num_buckets x 2 slots of 32 bytes (k <= 31) or of 64 bytes (k <= 63), then kmer_buckets x 2 entries of 32 bytes
    uint32_t num_buckets; // the region the keys hash into: inline slots and markers
    uint32_t enabled;
    /* the heavy keys' k-mers (keyed by sk_kmer_key) have a region of their own behind the keys' -- buckets num_buckets ..
       num_buckets + kmer_buckets - 1, same bucket format, same go-on flags -- so that it can be packed at its own load
       factor (SK_SLOTS_PER_KMER): a probe gets there only after its key's marker, one positive lookup in ten or twenty,
       and what a fuller region costs (more second choices) is paid by those alone */
    uint32_t kmer_buckets;
    /* length of the table's key m-mers. The table elects its own key (sk_key), so it need not be the dictionary's minimizer length:
       a shorter key makes longer super-k-mers (up to k - m + 1 k-mers an item) and so fewer items -- what the table's size is
       proportional to (HISTORY.md). Set by build_sk_table; every table-side function reads it from here, never dict_view::m */
    uint32_t m;
    /* table shard (multi-GPU, sharded.py): this replica's table holds only the keys with
       sk_owner(key, num_shards) == shard_id; lookups of other keys take the complete path */
    uint32_t num_shards;
    uint32_t shard_id;
};

/* owner of a key when the super-k-mer table is partitioned over several GPUs (independent of the slot hashes) */
SSH_HD uint32_t sk_owner(uint64_t key, uint32_t num_shards) {
    const uint64_t h = (key + 0x632BE59BD9B4E019ULL) * 0xA0761D6478BD642FULL;
    return mulhi32(uint32_t(h >> 32) ^ uint32_t(h >> 7), num_shards);
}

struct sk_hash_t {
    uint32_t bucket[SK_CHOICES];
    uint32_t fingerprint;  // 24 bits
};

SSH_HD sk_hash_t sk_hash(uint64_t key, uint32_t num_buckets) {
    uint64_t a = key * 0xFF51AFD7ED558CCDULL;
    a ^= a >> 32;
    a *= 0xC4CEB9FE1A85EC53ULL;
    a ^= a >> 29;
    uint64_t b = (a ^ key) * 0x9E3779B97F4A7C15ULL;
    b ^= b >> 31;
    sk_hash_t h;
    h.bucket[0] = mulhi32(uint32_t(a >> 32), num_buckets);
    h.bucket[1] = mulhi32(uint32_t(b >> 32), num_buckets);
    h.bucket[2] = mulhi32(uint32_t(b), num_buckets);
    const uint64_t c = (a + b) * 0xD6E8FEB86659FD93ULL;
    h.bucket[3] = mulhi32(uint32_t(c), num_buckets);
    h.bucket[4] = mulhi32(uint32_t((c ^ (c >> 31)) * 0x9E3779B1u + uint32_t(a)), num_buckets);
    /* out of `a` alone (its low 24 bits; bucket[0] comes out of its high 32): the first pass needs bucket[0] and the fingerprint
       only, and with the fingerprint taken from `c` (rounds 1-2) it had to compute b and c -- three 64-bit multiplies, a dozen
       VALU instructions each -- for every query, although only the twentieth that goes on ever looks at the other buckets. At
       k <= 63 the first pass is bound by its instructions (HISTORY.md). */
    h.fingerprint = uint32_t(a) & 0xFFFFFFu;
    return h;
}

/* choice c of a key's bucket sequence out of the key and `a` -- the first stage of sk_hash's arithmetic, whose low 24 bits are the key's
   fingerprint --, computed when it is asked for: what sk_hash lays out in advance (three more 64-bit multiplies), for a walk that nineteen
   times in twenty never leaves its first bucket (the streaming query: streaming.hip) */
SSH_HD uint64_t sk_hash_a(uint64_t key) {
    uint64_t a = key * 0xFF51AFD7ED558CCDULL;
    a ^= a >> 32;
    a *= 0xC4CEB9FE1A85EC53ULL;
    a ^= a >> 29;
    return a;
}
SSH_HD uint32_t sk_choice_of(uint64_t key, uint64_t a, uint32_t c, uint32_t num_buckets) {
    if (c == 0) return mulhi32(uint32_t(a >> 32), num_buckets);
    uint64_t b = (a ^ key) * 0x9E3779B97F4A7C15ULL;
    b ^= b >> 31;
    if (c == 1) return mulhi32(uint32_t(b >> 32), num_buckets);
    if (c == 2) return mulhi32(uint32_t(b), num_buckets);
    const uint64_t cc = (a + b) * 0xD6E8FEB86659FD93ULL;
    if (c == 3) return mulhi32(uint32_t(cc), num_buckets);
    return mulhi32(uint32_t((cc ^ (cc >> 31)) * 0x9E3779B1u + uint32_t(a)), num_buckets);
}

/* bucket sequence of a heavy key's k-mer: hashed into the k-mers' region */
SSH_HD sk_hash_t sk_hash_kmer_region(uint64_t kmer_key, uint32_t first_bucket, uint32_t kmer_buckets) {
    sk_hash_t h = sk_hash(kmer_key, kmer_buckets);
    for (uint32_t c = 0; c < SK_CHOICES; ++c) h.bucket[c] += first_bucket;
    return h;
}

/* Key under which the k-mers of a heavy key are entered one by one: a hash of the canonical k-mer, so that both
   strands agree. (sk_hash mixes it further; equal keys of different k-mers only cost a comparison.) */
template <int W>
SSH_HD uint64_t sk_kmer_key(kmer_w<W> const& x, kmer_w<W> const& x_rc) {
    kmer_w<W> const& c = kmer_less<W>(x_rc, x) ? x_rc : x;
    uint64_t v = c.w[0];
    if constexpr (W == 2) v ^= (c.w[1] + 0x9E3779B97F4A7C15ULL) * 0xD6E8FEB86659FD93ULL;
    return (v ^ (v >> 29)) * 0xBF58476D1CE4E5B9ULL + 0x632BE59BD9B4E019ULL;
}

/* Key of a k-mer for the super-k-mer table: an m-mer occurrence chosen so that a k-mer and its reverse complement
   choose the same one. The table is free to use any such function (it is built and probed with the same one), so
   this is NOT the reference's minimizer. The election looks at the first min(m, 12) bases of every m-mer occurrence
   only -- 24 bits per candidate, one funnel shift to extract them, one 24-bit multiply-add to hash them -- and carries
   the candidate's position in the low 6 bits of the hash, so that a running minimum replaces compare-and-select: two and a
   half VALU instructions per candidate (the first version hashed the whole m-mer: eleven; round 2: four; at k = 63, m = 25
   the 78 candidates made the first pass VALU-bound, HISTORY.md). Per strand the leftmost occurrence with the smallest
   26-bit hash wins; the strand with the smaller winning hash supplies the key -- the whole m-mer at the elected
   position --; equal hashes on the two strands = tie (no key: the caller takes the complete path). */
struct sk_key_t {
    uint64_t key;  // the m-mer, as read on the winning strand
    uint32_t pos;  // where it starts on that strand
    uint32_t hash; // the winning occurrence's 26-bit election hash (sk_key_persists)
    bool rc;       // the winning strand is the reverse complement of x
    bool tie;
};

constexpr uint32_t SK_POS_BITS = 6;  // positions 0 .. k - m <= 62
/* 26-bit election hash (in the top bits) of the first 12 bases of an m-mer occurrence, the candidate's position below it:
   (bases ^ salt) * constant + position. The salt keeps poly-A (all zero) from winning every election it takes part in. Two salts:
     k <= 31  SK_SELECT_SALT, irregular, applied to every candidate: consecutive candidates are shifts of one another and
              a multiplicative hash alone orders them in a correlated way -- 2 % more super-k-mers (table slots) measured
              with a salt that commutes with the shift; the lookup is bound by memory there, not by instructions;
     k <= 63  SK_SELECT_FLIP, whose pattern repeats every base and so commutes with a shift by whole bases: the k-mer's
              words are salted once instead of 78 candidates one by one -- there the election is what bounds the first
              pass (92 % VALU utilisation before), and the denser election costs less than it saves. */
constexpr uint32_t SK_SELECT_SALT = 0x6A09E667u, SK_SELECT_FLIP = 0xAAAAAAAAu;
template <int W>
SSH_HD uint32_t sk_select_salt() { return W == 1 ? SK_SELECT_SALT : SK_SELECT_FLIP; }
/* ONE instruction for hash and position (round 4; rounds 2-3: a 32-bit multiply and an or): v_mad_u32_u24 multiplies the low
   24 bits of its operands -- it cuts the 12-base window out of the funnel-shifted word by itself -- and adds the position,
   an inline constant. The multiplier carries the shift that makes room for the position: (window * odd 18-bit) << 6 -- the
   low 32 bits of that are a 26-bit multiplicative hash (a bijection on the window's low 26 bits, its best-mixed bits on
   top) above six zero bits. */
constexpr uint32_t SK_SELECT_MUL = 0x278DDu << SK_POS_BITS;  // 0x9E3740: 24 bits
static_assert(SK_SELECT_MUL < (1u << 24) && ((SK_SELECT_MUL >> SK_POS_BITS) & 1u), "an odd multiplier times 2^6, within 24 bits");
SSH_HD uint32_t sk_select_hash(uint32_t salted, uint32_t position, uint32_t mul) {  // mul: sk_select_mul()
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(salted, mul) + position;
#else
    return (salted & 0xFFFFFFu) * mul + position;
#endif
}
/* the multiplier in a scalar register, its value hidden from the compiler: knowing that its low six bits are zero hipcc turns
   the addition of the position into an OR, and for multiply-then-OR it has no single instruction (v_mul_u32_u24 + v_or_b32:
   the three instructions per candidate this was meant to leave behind) */
SSH_HD uint32_t sk_select_mul() {
    uint32_t mul = SK_SELECT_MUL;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(mul));
#endif
    return mul;
}

/* low 32 bits of (hi:lo) >> s, s in 0..31: one v_alignbit_b32 on the device (a 64-bit shift costs five times as much) */
SSH_HD uint32_t funnel32(uint32_t lo, uint32_t hi, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return uint32_t(((uint64_t(hi) << 32) | lo) >> s);
#endif
}

/* the election over one strand: f = the strand's words (SALT_EACH false: already salted), one more word behind them.
   MASKED: m < 12: the hashed window is cut to 2m bits -- shifted left by `mask` = 24 - 2m bits, out of the multiply's reach.
   Two and a half instructions per candidate at best: funnel shift, multiply-add, and a minimum shared between two (v_min3) */
template <int D, bool MASKED, bool SALT_EACH>
SSH_HD uint32_t sk_elect(uint32_t const (&f)[D + 1], uint32_t n, uint32_t mask) {
    uint32_t best = 0xFFFFFFFFu;
    const uint32_t mul = sk_select_mul();
    auto candidate = [&](uint32_t lo, uint32_t hi, uint32_t t, uint32_t i) {
        uint32_t word = funnel32(lo, hi, 2 * t);
        if constexpr (SALT_EACH) word ^= SK_SELECT_SALT;
        if constexpr (MASKED) word <<= mask;  // (what lies beyond the m-mer leaves the 24 bits the multiply looks at)
        const uint32_t h = sk_select_hash(word, i, mul);
        best = h < best ? h : best;
    };
    /* n is the same for all lanes. A word whose 16 candidates all exist is unrolled: shifts and positions are constants,
       nothing for the scalar unit to do. The last, partial word is entered at the right candidate through a switch (one
       computed jump; its candidates are constants as well, taken from the last down to the first) -- as a counted loop
       (rounds 2-3) it cost ten scalar instructions per candidate, 140 of the 304 of a k = 63 first-pass wave. (One loop over
       all candidates with a bound test per candidate costs a select per candidate on the vector side, or -- the bound folded
       into a scalar tag -- four scalar instructions per candidate: at k = 63 the CU's single scalar unit was then 72 % busy.) */
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < D; ++j) {
        if (16 * uint32_t(j + 1) <= n) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (uint32_t t = 0; t < 16; ++t) candidate(f[j], f[j + 1], t, 16 * uint32_t(j) + t);
        } else {
            const uint32_t rest = n > 16 * uint32_t(j) ? n - 16 * uint32_t(j) : 0u;
            const uint32_t lo = f[j], hi = f[j + 1], at = 16 * uint32_t(j);
            switch (rest) {  // (rest < 16: a full word was taken by the branch above)
                case 15: candidate(lo, hi, 14, at + 14); [[fallthrough]];
                case 14: candidate(lo, hi, 13, at + 13); [[fallthrough]];
                case 13: candidate(lo, hi, 12, at + 12); [[fallthrough]];
                case 12: candidate(lo, hi, 11, at + 11); [[fallthrough]];
                case 11: candidate(lo, hi, 10, at + 10); [[fallthrough]];
                case 10: candidate(lo, hi, 9, at + 9); [[fallthrough]];
                case 9: candidate(lo, hi, 8, at + 8); [[fallthrough]];
                case 8: candidate(lo, hi, 7, at + 7); [[fallthrough]];
                case 7: candidate(lo, hi, 6, at + 6); [[fallthrough]];
                case 6: candidate(lo, hi, 5, at + 5); [[fallthrough]];
                case 5: candidate(lo, hi, 4, at + 4); [[fallthrough]];
                case 4: candidate(lo, hi, 3, at + 3); [[fallthrough]];
                case 3: candidate(lo, hi, 2, at + 2); [[fallthrough]];
                case 2: candidate(lo, hi, 1, at + 1); [[fallthrough]];
                case 1: candidate(lo, hi, 0, at + 0); [[fallthrough]];
                default: break;
            }
            break;
        }
    }
    return best;
}

template <int W>
SSH_HD sk_key_t sk_key(kmer_w<W> const& x, kmer_w<W> const& x_rc, uint32_t k, uint32_t m) {
    /* the k-mer as 32-bit words (16 bases each), one more word behind it: the 16 bases starting at base i = 16 j + t are
       words j, j+1 shifted right by 2 t -- a funnel shift instead of a 64-bit (or 128-bit) shift */
    constexpr int D = 2 * W;
    constexpr bool EACH = W == 1;                       // see SK_SELECT_SALT
    constexpr uint32_t once = EACH ? 0u : SK_SELECT_FLIP;
    uint32_t f[D + 1], r[D + 1];
    for (int j = 0; j < W; ++j) {
        f[2 * j] = uint32_t(x.w[j]) ^ once;
        f[2 * j + 1] = uint32_t(x.w[j] >> 32) ^ once;
        r[2 * j] = uint32_t(x_rc.w[j]) ^ once;
        r[2 * j + 1] = uint32_t(x_rc.w[j] >> 32) ^ once;
    }
    f[D] = r[D] = once;
    const uint32_t n = k - m + 1;
    uint32_t best_f, best_r;
    if (m >= 12) {  // uniform: the 12-base window lies inside every m-mer
        best_f = sk_elect<D, false, EACH>(f, n, 0u);
        best_r = sk_elect<D, false, EACH>(r, n, 0u);
    } else {
        const uint32_t mask = 24 - 2 * m;  // (a shift count: see sk_elect)
        best_f = sk_elect<D, true, EACH>(f, n, mask);
        best_r = sk_elect<D, true, EACH>(r, n, mask);
    }
    sk_key_t out;
    out.rc = (best_r >> SK_POS_BITS) < (best_f >> SK_POS_BITS);
    out.tie = (best_r >> SK_POS_BITS) == (best_f >> SK_POS_BITS);
    out.pos = (out.rc ? best_r : best_f) & ((1u << SK_POS_BITS) - 1u);
    out.hash = (out.rc ? best_r : best_f) >> SK_POS_BITS;
    out.key = kmer_shr_chars<W>(out.rc ? x_rc : x, out.pos).w[0] & low_mask(2 * m);
    return out;
}

/* For how many of the k-mers that FOLLOW a k-mer along a read does its key last? The k-mer one base further on keeps every
   candidate of the election but two and gains two: the m-mer that ends at its last base, and -- on the other strand -- that
   m-mer's reverse complement. The elected occurrence stays elected for as long as (1) it is still inside the k-mer and (2) no
   candidate that has come in hashes below it; a newcomer that hashes EQUAL ends the count as well (it may be the leftmost of its
   strand, or tie the strands). So: the number t <= SK_PERSIST_MAX such that sk_key of the k-mers 1 .. t bases further on elects
   the same occurrence (same key, at a position that moves with the k-mer) -- never more than that, possibly fewer.
     ahead_f  the 32 bases of the read that start k - m + 1 bases behind the k-mer's first: the newcomers' first bases, forward
     ahead_r  the 32 bases that start k + 1 - min(m, 12) behind it: the last min(m, 12) bases of the following k-mers -- whose
              reverse complements are the first bases of the newcomers on the other strand
   What the streaming query does with it (streaming.hip): a k-mer whose key is proven absent from the table -- or present with
   one slot that differs from the read at a base the following k-mers still hold -- settles the k-mers behind it that share its
   key without looking at them. tests/cpp/check_table_key.cpp holds it against sk_key base by base. */
constexpr uint32_t SK_PERSIST_MAX = 20;  // 32 bases ahead hold 21 windows of 12
template <int W>
SSH_HD uint32_t sk_key_persists(sk_key_t const& kk, uint32_t k, uint32_t m, uint64_t ahead_f, uint64_t ahead_r) {
    const uint32_t inside = kk.rc ? k - m - kk.pos : kk.pos;  // (1): that many k-mers further on still hold the occurrence
    uint32_t most = inside < SK_PERSIST_MAX ? inside : SK_PERSIST_MAX;
    const uint32_t threshold = (kk.hash << SK_POS_BITS) | ((1u << SK_POS_BITS) - 1u);  // umul24(...) <= threshold  <=>  its 26-bit hash <= kk.hash
    const uint32_t mul = sk_select_mul(), salt = sk_select_salt<W>();
    const uint32_t L = m < 12 ? m : 12;
    const uint64_t behind = revcomp_word(ahead_r);  // base 31 - i = the complement of base i of ahead_r
    const uint32_t f_lo = uint32_t(ahead_f), f_hi = uint32_t(ahead_f >> 32), r_lo = uint32_t(behind), r_hi = uint32_t(behind >> 32);
    uint32_t first = SK_PERSIST_MAX + 1;  // the first k-mer (1-based) with a newcomer at or below the threshold
    auto newcomer = [&](uint32_t word, uint32_t t) {
        word ^= salt;
        if (m < 12) word <<= 24 - 2 * m;  // uniform (sk_elect: MASKED)
        first = sk_select_hash(word, 0u, mul) <= threshold ? t : first;
    };
    if (m >= 12) {  // uniform: every shift below is a constant
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t t = SK_PERSIST_MAX; t >= 1; --t) {
            if (t > k - m) continue;  // uniform: no key lasts longer than k - m k-mers (k = 31, m = 21: ten steps of the twenty)
            /* forward: the 12 bases from base t - 1 of ahead_f; other strand: the reverse complement of the 12 bases from base t - 1 of
               ahead_r = the 12 bases from base 32 - 12 - (t - 1) of `behind` */
            newcomer(t - 1 < 16 ? funnel32(f_lo, f_hi, 2 * (t - 1)) : f_hi >> (2 * (t - 1 - 16)), t);
            const uint32_t at = 21 - t;
            newcomer(at < 16 ? funnel32(r_lo, r_hi, 2 * at) : r_hi >> (2 * (at - 16)), t);
        }
    } else {
        for (uint32_t t = SK_PERSIST_MAX; t >= 1; --t) {
            newcomer(uint32_t(ahead_f >> (2 * (t - 1))) & uint32_t(low_mask(2 * L)), t);
            newcomer(uint32_t(behind >> (2 * (32 - L - (t - 1)))) & uint32_t(low_mask(2 * L)), t);
        }
    }
    return first - 1 < most ? first - 1 : most;
}

struct dict_view {
    uint32_t k, m;
    uint32_t canonical;
    uint32_t skew_parts;
    uint64_t hash_magic;
    uint64_t num_kmers, num_strings, num_bases;

    void const* granules;  // atom32[] when k <= 31, granule[] otherwise
    uint64_t const* endpoints;  // num_strings + 1

    mphf_view minimizers;
    uint64_t const* codewords;  // one u64 per minimizer id: code | fingerprint << cw_width -- or, cw_packed, the reference's
                                // bit-packed control codewords as they are (no fingerprint: the strings settle a foreign minimizer)
    uint32_t cw_width;
    uint32_t cw_packed;
    uint32_t off_width;  // width of the entries of mid_load / heavy_load
    uint32_t const* begin_buckets_of_size;  // 65 entries
    uint64_t const* mid_load;
    uint64_t const* heavy_load;
    uint64_t heavy_size;
    directory_view directory;
    sk_view sk;
    uint64_t const* weight_starts;  // run-length intervals of the weights over the k-mer ids (index.hpp), or null
    uint64_t const* weight_values;
    uint64_t num_weight_intervals;
};

/* Struct-of-arrays lookup output; any pointer except kmer_id may be null.
   Field meaning: lookup_result, include/util.hpp:38-62. */
struct result_view {
    uint64_t* kmer_id;
    uint64_t* kmer_id_in_string;
    uint64_t* kmer_offset;
    uint64_t* string_id;
    uint64_t* string_begin;
    uint64_t* string_end;
    int8_t* kmer_orientation;
    uint8_t* minimizer_found;
};

}  // namespace sshash_amd
