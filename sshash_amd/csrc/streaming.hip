// streaming.hip -- batched streaming query (gfx950 only).
//
// Reference: streaming_query<Dict,canonical>::lookup / seed (include/streaming_query.hpp:56-109,
// 144-197) driven per read by src/query.cpp:78-108. The state machine is sequential inside a
// read and independent across reads, so reads are the parallel dimension: a lane owns a read and
// handles its EVENTS -- a seed (one point lookup, through the super-k-mer table when the replica has
// one), or the run of extensions behind a hit, measured as a longest common prefix of the read and
// the strings, 32 bases a step (streaming_run_kernel below). With the granule layout the reference's
// `remaining_string_bases` counter is implicit: a string-start mark at the next base is exactly
// remaining == 0. The negative short-cut (:150-157) becomes "the k-mers behind a miss that elect the
// same table key, and cannot be in that key's slot either, are negative too" -- the same k-mers are
// negative either way, so the counters are unchanged.
// Output: the six counters of streaming_query_report (include/util.hpp:21-36); per-k-mer results:
// the position-parallel pipeline further down.
#include <hip/hip_runtime.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <thread>

#include "engine.hpp"
#include "hooks.hpp"
#include "reads.hpp"
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

/* The counters of a workgroup (of 256) reach `report` with ONE set of atomics: the six counters share a cache line, and a set
   per wave -- 2 x 10^5 atomics on one line for 3 x 10^8 bases -- serialises at ~90 atomics/us (HISTORY.md). Called by
   every lane of the workgroup. */
__device__ __forceinline__ void block_report(uint64_t c_kmers, uint64_t c_invalid, uint64_t c_negative, uint64_t c_searches,
                                             uint64_t c_extensions, uint64_t* __restrict__ report) {
    __shared__ uint64_t partial[4][5];
    c_kmers = wave_sum(c_kmers);
    c_invalid = wave_sum(c_invalid);
    c_negative = wave_sum(c_negative);
    c_searches = wave_sum(c_searches);
    c_extensions = wave_sum(c_extensions);
    if ((threadIdx.x & 63) == 0) {
        uint64_t* mine = partial[threadIdx.x >> 6];
        mine[0] = c_kmers;
        mine[1] = c_invalid;
        mine[2] = c_negative;
        mine[3] = c_searches;
        mine[4] = c_extensions;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    uint64_t t[5];
    for (int j = 0; j < 5; ++j) t[j] = partial[0][j] + partial[1][j] + partial[2][j] + partial[3][j];
    if (t[0] == 0) return;
    atomicAdd(reinterpret_cast<unsigned long long*>(report + 0), (unsigned long long)t[0]);
    atomicAdd(reinterpret_cast<unsigned long long*>(report + 1), (unsigned long long)(t[3] + t[4]));
    atomicAdd(reinterpret_cast<unsigned long long*>(report + 2), (unsigned long long)t[2]);
    atomicAdd(reinterpret_cast<unsigned long long*>(report + 3), (unsigned long long)t[1]);
    atomicAdd(reinterpret_cast<unsigned long long*>(report + 4), (unsigned long long)t[3]);
    atomicAdd(reinterpret_cast<unsigned long long*>(report + 5), (unsigned long long)t[4]);
}

/* ==== the run-based streaming kernel (round 5) =====================================================================
   The per-base walk of rounds 1-4 (one read a lane, the lanes in step along their reads, k-mer and table key rolled base by base: HISTORY.md)
   spent its instructions where nothing is decided: 61 % of the k-mers of a high-hit read extend the
   run of the k-mer before them, and it rolled the k-mer, its reverse complement and two election candidates for every one of them,
   64 reads in step (397 vector instructions per base and wave; 48 G k-mers/s on that set, 104 on random reads, 37 on config C4's). A run is a longest common prefix: once a seed has put the read at
   offset `off` of the strings in orientation `o`, the next k-mers are extensions for as long as the read's NEXT BASE equals the
   strings' next base in that direction and no string starts there -- streaming_query.hpp:86-100 compares whole k-mers
   (`expected == kmer or expected == kmer_rc`), but given the k-mer before matched the strings' k-mer before, the two k-mers share
   k - 1 bases and differ or agree in the one new base; the other alternative of the reference's test can then only hold when the
   new bases agree as well:
       forward   F' = F[1:] + s, x' = x[1:] + r, F = x.   F' = rc(x') means y + s = comp(r) + rc(y) for y = x[1:]:  y[0] = comp(r),
                 y[i] = comp(y[k-2-(i-1)]) ..., s = comp(y[0]) = r.
       backward  F' = s + F[:-1], F = rc(x), rc(x') = comp(r) + F[:-1].   F' = x' means s + z = rc(z) + r for z = F[:-1]:
                 r = z[k-2], s = comp(z[k-2]) = comp(r).
   So the length of a run is min(longest common prefix of the read's tail and the strings -- 32 bases a step: XOR of two packed
   words and a count of trailing zeros --, bases left in the string, valid bases left in the read), and nothing per base is left.

   What remains are the SEEDS (the negative k-mers, and one search per run): 0.29 per base on the high-hit set, 0.41 on config
   C4's. The kernel is a loop over EVENTS, not over bases: in every turn every lane of a wave handles the next event of its own
   read -- a seed at its own position, or the extension that follows a hit -- so the lanes of a wave are no longer in step along
   their reads, and a lane that is done with its read takes the next one of its wave's share at once (the wave hands them out in
   order: a ballot and a prefix count, no atomics). The reads are packed to two bits a base (and a validity bit a base) by a pass
   of their own first: a seed cuts its k-mer out of two or three words, and the invalid k-mers around an `N` are counted and
   skipped with one subtraction (streaming_query.hpp:59-65: a k-mer is invalid iff one of its k characters is). */

/* pass 1: lane t packs bases [8t, 8t + 8): 16 bits of codes ((c >> 1) & 3, include/kmer.hpp:118; base i of a 32-base word in bits
   2i, 2i + 1) and 8 validity bits (A C G T a c g t, include/kmer.hpp:209-219) */
__global__ void __launch_bounds__(256)
stream_pack_kernel(const char* __restrict__ bases, const uint64_t total_bases, uint16_t* __restrict__ packed, uint8_t* __restrict__ okay) {
    const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t at = 8 * t;
    if (at >= total_bases) return;
    uint64_t v = 0;
    if (at + 8 <= total_bases && ((reinterpret_cast<uintptr_t>(bases) + at) & 7) == 0) {
        v = *reinterpret_cast<const uint64_t*>(bases + at);
    } else {
        for (uint32_t b = 0; b < 8 && at + b < total_bases; ++b) v |= uint64_t(uint8_t(bases[at + b])) << (8 * b);
    }
    uint64_t two = (v >> 1) & 0x0303030303030303ULL;
    two = (two | (two >> 6)) & 0x000F000F000F000FULL;
    two = (two | (two >> 12)) & 0x000000FF000000FFULL;
    two = two | (two >> 24);
    const uint64_t u = v & 0xDFDFDFDFDFDFDFDFULL;  // fold the case
    auto nonzero = [](uint64_t z) { return ((z & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | z; };  // bit 7 of every byte that is not zero
    uint64_t ok = ~(nonzero(u ^ 0x4141414141414141ULL) & nonzero(u ^ 0x4343434343434343ULL) & nonzero(u ^ 0x4747474747474747ULL) &
                    nonzero(u ^ 0x5454545454545454ULL)) & 0x8080808080808080ULL;
    ok >>= 7;
    ok |= ok >> 7;
    ok |= ok >> 14;
    ok |= ok >> 28;
    packed[t] = uint16_t(two);
    okay[t] = uint8_t(ok);
}

/* first invalid base in [from, end), or `end` */
__device__ __forceinline__ uint64_t first_invalid_base(const uint64_t* __restrict__ okay, uint64_t from, uint64_t end) {
    uint64_t p = from;
    while (p < end) {
        const uint64_t bad = ~okay[p >> 6] >> (p & 63u);
        if (bad) {
            p += uint64_t(__builtin_ctzll(bad));
            break;
        }
        p = (p | 63u) + 1;
    }
    return p < end ? p : end;
}

/* first valid base in [from, end), or `end` */
__device__ __forceinline__ uint64_t first_valid_base(const uint64_t* __restrict__ okay, uint64_t from, uint64_t end) {
    uint64_t p = from;
    while (p < end) {
        const uint64_t good = okay[p >> 6] >> (p & 63u);
        if (good) {
            p += uint64_t(__builtin_ctzll(good));
            break;
        }
        p = (p | 63u) + 1;
    }
    return p < end ? p : end;
}

/* the 32 bases of the packed reads starting at base p */
__device__ __forceinline__ uint64_t read_bases32(const uint64_t* __restrict__ packed, uint64_t p) {
    const uint64_t i = p >> 5;
    return funnel_shr(packed[i], packed[i + 1], 2 * (uint32_t(p) & 31u));
}

/* the 32 bases of the strings starting at base p; mark bit j (0 .. 32): a string starts at base p + j */
template <int W>
__device__ __forceinline__ void string_bases32(dict_view const& d, uint64_t p, uint64_t& bases, uint64_t& marks) {
    const uint32_t rel = uint32_t(p) & 31u;
    if constexpr (W == 1) {
        const uint4* A = reinterpret_cast<const uint4*>(d.granules) + 2 * (p >> 5);
        const uint4 q0 = A[0], q1 = A[1];
        bases = funnel_shr(uint64_t(q0.x) | (uint64_t(q0.y) << 32), uint64_t(q0.z) | (uint64_t(q0.w) << 32), 2 * rel);
        marks = (uint64_t(q1.x) | (uint64_t(q1.y) << 32)) >> rel;
    } else {
        const uint4* G = reinterpret_cast<const uint4*>(d.granules) + (p >> 5);
        const uint4 g0 = G[0], g1 = G[1];
        bases = funnel_shr(uint64_t(g0.z) | (uint64_t(g0.w) << 32), uint64_t(g1.z) | (uint64_t(g1.w) << 32), 2 * rel);
        marks = (uint64_t(g0.y) | (uint64_t(g1.y) << 32)) >> rel;
    }
}

/* Length of the run behind a hit: the read's k-mer that ends at base `b` - 1 lies at offset `off` of the strings (orientation
   `ori`); at most `room` k-mers follow it inside the read's valid bases. Returns how many of them are extensions. */
/* One step of a run's measurement: the 32 bases of the strings that come next in the run's direction and the marks that stop it,
   next to the read's 32 bases from base b + run. Forward, the t-th extension gains the strings' base off + k - 1 + t and leaves its
   string iff a string starts there. Backward it gains the base off - t, complemented, and leaves its string iff a string starts at
   off - t + 1 (streaming_query.hpp:92: remaining_string_bases = kmer_id_in_string going backward): a step takes the `have` bases below
   `top` = off - run. Loads only: what they bring is looked at by run_step_length. */
struct run_step_t {
    uint64_t s, marks, read;
    uint32_t have;  // backward: how many bases lie below `top` (32, fewer at the very start of the strings; 0: none)
};

template <int W>
__device__ __forceinline__ run_step_t run_step_load(dict_view const& d, const uint64_t* __restrict__ packed, uint64_t off, bool forward,
                                                    uint64_t b, uint64_t run) {
    run_step_t t;
    const uint64_t top = off - run;
    t.have = forward || top >= 32 ? 32u : uint32_t(top);
    string_bases32<W>(d, forward ? off + d.k + run : top - t.have, t.s, t.marks);
    t.read = read_bases32(packed, b + run);
    return t;
}

/* how many of the (at most 32) extensions of a step hold: the strings' bases along the run -- backward: the `have` bases below `top`,
   the last of them moved to place 31, reversed and complemented: place i = comp(S[top - 1 - i]) -- against the read's, up to the first
   mark that stops the run -- bit 31 - i of `gate` stops extension i; backward the mark of base top - i: marks bit (have - i) */
__device__ __forceinline__ uint32_t run_step_length(run_step_t const& t, bool forward) {
    const uint64_t along = forward ? t.s : revcomp_word(t.s << ((2 * (32 - t.have)) & 63u));  // (have = 0: a shift by 64 is not one; the step is min(., have) = 0 whatever this gives)
    const uint64_t diff = along ^ t.read;
    const uint32_t same = diff ? uint32_t(__builtin_ctzll(diff)) >> 1 : 32u;
    const uint32_t gate = forward ? uint32_t(__brev(uint32_t(t.marks))) : uint32_t((t.marks >> 1) << (32 - t.have));
    const uint32_t inside = gate ? uint32_t(__builtin_clz(gate)) : 32u;
    const uint32_t step = same < inside ? same : inside;
    return step < t.have ? step : t.have;
}

/* Length of the run behind a hit: the read's k-mer that ends at base `b` - 1 lies at offset `off` of the strings (orientation `ori`);
   at most `room` k-mers follow it inside the read's valid bases; `first`: the first step's loads, asked for by the caller early in
   its turn so that they travel while the turn's bucket lines do. Returns how many of the following k-mers are extensions. */
template <int W>
__device__ __forceinline__ uint64_t extend_run(dict_view const& d, const uint64_t* __restrict__ packed, uint64_t off, int ori, uint64_t b,
                                               uint64_t room, run_step_t t) {
    const bool forward = ori > 0;
    uint64_t run = 0;
    for (;;) {
        uint64_t step = run_step_length(t, forward);
        if (step > room - run) step = room - run;
        run += step;
        if (step < 32 || run >= room) break;
        t = run_step_load<W>(d, packed, off, forward, b, run);
    }
    return run;
}

/* (five waves a SIMD: 96 registers. Round 5's steps there, same-box: k <= 31 four -> five waves 110 -> 126 G k-mers/s; k <= 63, once 32-bit
   counters and a run measurement without early loads had made room, 138 -> 147 with 20 bytes of scratch. Compiled for six: 14 / 31 % slower,
   profiles/r05/streaming_run_kernel_waves_per_simd_ab.txt.) */
#ifndef SSHASH_STREAM_WAVES
#define SSHASH_STREAM_WAVES 5
#endif
template <int W, bool CANON, bool SK>
__global__ void __launch_bounds__(256, SSHASH_STREAM_WAVES)
streaming_run_kernel(const dict_view d, const skew_part_dev* __restrict__ skew, const uint64_t* __restrict__ packed,
                     const uint64_t* __restrict__ okay, const uint64_t* __restrict__ offsets, const uint64_t n_reads,
                     const uint64_t reads_per_wave, const uint32_t move_out_every, uint64_t* __restrict__ report) {
    __shared__ uint4 stage[SK ? 4 * 256 : 1];  // a wave's 64 bucket lines on their way from the quads that fetch them to the lanes that own them
    uint4* const wave_stage = stage + (SK ? (threadIdx.x >> 6) * 256 : 0);
    /* the counters are 32 bits wide in the lanes (six registers fewer than five 64-bit ones: with them the k <= 63 kernel fits five waves a
       SIMD); every 2^16 turns -- long before one could wrap -- the wave moves them into 64-bit totals of its own in LDS */
    __shared__ unsigned long long moved_out[4][5];
    uint32_t c_invalid = 0, c_negative = 0, c_searches = 0, c_extensions = 0;  // (the k-mers of a read go straight to the wave's total)
    const uint32_t k = d.k;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long* const wave_moved_out = moved_out[threadIdx.x >> 6];
    if (lane < 5) wave_moved_out[lane] = 0;
    /* this wave's share of the reads, handed out in order to whichever lane is done with its read */
    const uint64_t wave = uint64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint64_t next = wave * reads_per_wave, last = next + reads_per_wave;
    if (next > n_reads) next = n_reads;
    if (last > n_reads) last = n_reads;
    uint64_t cur = 0, rd_end = 0, inv = 0;  // the k-mer to settle starts at base cur of the read that ends at rd_end; inv: first invalid base >= cur
    bool pending = false;                   // the k-mer before cur was found at `off`, orientation `ori`: the run behind it is to be measured
    uint64_t off = 0;
    int ori = 1;
    bool neg_unknown_mini = false;          // (no table) streaming_query.hpp:150-157
    uint64_t prev_f = 0, prev_r = 0;
    /* a seed whose probe has to go on past its key's first bucket is WALKING: its k-mer, its key and where the walk stands are kept
       from turn to turn. `where`: the choice of the sequence it is on | flags | the choice of the key's sequence to come back to */
    constexpr uint32_t WALK_CHOICE = 7u, WALK_COMPACT = 8u, WALK_VISITED = 16u, WALK_BACK_SHIFT = 8u, WALK_NO_RETURN = 7u;
    constexpr uint32_t WALK_LASTS_SHIFT = 16u, WALK_LASTS = 63u;  // what the slots met so far allow (min over the walk; 63: anything)
    /* A HEAVY key remembered (round 6). The k-mers of a heavy key are entered under keys of their own, and a seed learns that from the key's
       marker: one bucket for the marker, another -- a turn later -- on the k-mer's own sequence. Over a substitution inside a repeat every one
       of the k (negative) k-mers paid both, and none stands for its neighbours: on the high-hit set a fifth of all lane-turns were such
       second turns (profiles/r06/streaming_lane_occupancy_per_turn.txt). The k-mers that FOLLOW along the read and elect the same key
       occurrence (sk_key_persists) would walk the same buckets of the key's sequence up to the same marker; if none of those buckets holds
       an inline slot with the key's fingerprint (WALK_INLINE: then nothing in them can answer any k-mer of this key) they may start where
       that walk ended: on their own sequence, with the same choice of the key's sequence to come back to. Fields of `where` that outlive
       the seed: how many following k-mers may still do so, and that choice. */
    constexpr uint32_t WALK_INLINE = 32u, WALK_HEAVY_BACK_SHIFT = 11u, WALK_HEAVY_SHIFT = 24u, WALK_HEAVY = 63u, WALK_HEAVY_PENDING = 1u << 30;
    constexpr uint32_t WALK_HEAVY_FIELDS = (7u << WALK_HEAVY_BACK_SHIFT) | (WALK_HEAVY << WALK_HEAVY_SHIFT) | WALK_HEAVY_PENDING;
    bool walking = false;
    kmer_w<W> x = kmer_zero<W>(), x_rc = kmer_zero<W>();
    sk_key_t kk{};
    uint64_t on = 0, on_a = 0;
    uint32_t where = 0;
    uint32_t turns = 0;
#ifdef SSHASH_STREAM_STATS
    /* (a debug build, tools/jobs/r06_stream_stats.sh: what the lanes of a wave do per turn -- wave-level sums in scalar registers, out
       through report[6 ..]; the caller's report has 16 entries then) */
    unsigned long long st_turns = 0, st_fresh = 0, st_walk = 0, st_ext = 0, st_slot1 = 0, st_inv = 0, st_idle = 0, st_full = 0, st_short = 0;
    uint64_t st_kept_lane = 0;
#define STAT(var, pred) var += (unsigned long long)__popcll(__ballot(pred))
#else
#define STAT(var, pred) (void)0
#endif
    for (;;) {
#ifdef SSHASH_STREAM_STATS
        ++st_turns;
#endif
        if (++turns >= move_out_every) {  // (scalar; a turn adds less than 2^15 to a lane's counter -- a longer run of extensions or of invalid k-mers goes to the wave's 64-bit totals at once --, so 2^16 turns stay below 2^31)
            turns = 0;
            const uint64_t i = wave_sum(c_invalid), n = wave_sum(c_negative), f = wave_sum(c_searches), e = wave_sum(c_extensions);
            if (lane == 0) {
                wave_moved_out[0] += i;
                wave_moved_out[1] += n;
                wave_moved_out[2] += f;
                wave_moved_out[3] += e;
            }
            c_invalid = c_negative = c_searches = c_extensions = 0;
        }
        /* -- the run behind the hit of the turn before, THEN this turn's seed (round 6): the lane measures its run and goes straight on to
              whatever lies behind it in the same turn -- the negative over the substitution that ended the run, as a rule; the NEXT READ when
              the run reached the end of this one (which is why the run comes before the reads are handed out). Until round 5 a lane did one
              or the other in a turn (the run's first loads travelled beside the seeds' bucket lines: one wait for both), and a hit -- run --
              miss cycle took a turn more than it has events (that order, as a build of this file, is the "run after seed" column of
              profiles/r06/streaming_*_ab.txt). (`pending` is only set when the k-mer behind the hit lies inside the read's
              valid bases: no test of its own here.) -- */
        STAT(st_ext, pending);
        if (pending) {
            const uint64_t b = cur + k - 1, valid_end = inv < rd_end ? inv : rd_end;
            const uint64_t run = extend_run<W>(d, packed, off, ori, b, valid_end - b, run_step_load<W>(d, packed, off, ori > 0, b, 0));
            if (run >> 15) atomicAdd(wave_moved_out + 3, (unsigned long long)run);  // (past what the lane's 32-bit counter may take in one turn: 2^16 turns lie between two move-outs)
            else c_extensions += uint32_t(run);
            cur += run;
            if (run) where &= ~WALK_HEAVY_FIELDS;  // (the k-mer behind a run elects a key of its own)
        }
        pending = false;
        /* -- the reads: whoever has none left takes the next of the wave's share -- */
        const bool want = !walking && cur + k > rd_end;
        const uint64_t wants = __ballot(want);
        if (wants != 0 && next < last) {
            const uint64_t rank = uint64_t(__popcll(wants & ((uint64_t(1) << lane) - 1)));
            if (want && rank < last - next) {
                const uint64_t r = next + rank;
                cur = offsets[r];
                rd_end = offsets[r + 1];
                if (rd_end - cur >= k) atomicAdd(wave_moved_out + 4, (unsigned long long)(rd_end - cur - k + 1));
                inv = first_invalid_base(okay, cur, rd_end);
                neg_unknown_mini = false;
                where = 0;
            }
            const uint64_t taken = uint64_t(__popcll(wants));
            next = taken < last - next ? next + taken : last;
        }
        bool live = cur + k <= rd_end;
        if (next >= last && __ballot(live) == 0) break;
        /* -- the k-mers over an invalid base: all invalid (streaming_query.hpp:59-65), counted and skipped -- a whole RUN of invalid
              bases a step: every k-mer that starts at or before the run's last base holds one of them (those from `cur` on reach
              `inv`: cur + k > inv). (Round 5 advanced one invalid base per iteration, each with a dependent load, the other 63 lanes
              waiting: a read with a long run of N was a cliff) -- */
        uint32_t c_invalid_turn = 0;
        STAT(st_inv, live && cur + k > inv);
        while (live && cur + k > inv) {
            const uint64_t nv = first_valid_base(okay, inv + 1, rd_end);  // the run of invalid bases is [inv, nv)
            const uint64_t last_over_it = nv - 1 < rd_end - k ? nv - 1 : rd_end - k;
            const uint64_t over = last_over_it - cur + 1;
            /* (a turn may add 2^15 - 1 to a lane's 32-bit counter and no more -- 2^16 turns lie between two move-outs --; this loop
               can cross a whole read in one turn, so what it adds is bounded by the loop's own sum, kept here) */
            if ((over + c_invalid_turn) >> 15) atomicAdd(wave_moved_out + 0, (unsigned long long)over);
            else c_invalid_turn += uint32_t(over);
            cur = nv;
            inv = first_invalid_base(okay, nv, rd_end);
            neg_unknown_mini = false;
            where = 0;
            live = cur + k <= rd_end;
        }
        c_invalid += c_invalid_turn;
        const uint64_t valid_end = inv < rd_end ? inv : rd_end;
        /* -- seed() at cur (streaming_query.hpp:144-197): the k-mer, its key, its key's first bucket; a lane in the middle of a walk
              (its key's first bucket was not the end of it) keeps its k-mer and comes with the walk's next bucket instead -- */
        uint64_t ahead_f = 0, ahead_r = 0;
        bool table = false;
        STAT(st_fresh, live && !walking);
        STAT(st_walk, live && walking);
        STAT(st_idle, !live);
        if (live && !walking) {
            const uint64_t i = cur >> 5;
            const uint32_t sh = 2 * (uint32_t(cur) & 31u);
            const uint64_t w0 = packed[i], w1 = packed[i + 1];
            x.w[0] = funnel_shr(w0, w1, sh);
            if constexpr (W == 2) x.w[1] = funnel_shr(w1, packed[i + 2], sh);
            x = kmer_take_chars<W>(x, k);
            x_rc = kmer_revcomp<W>(x, k);
            if constexpr (SK) {
                /* what sk_key_persists looks at, asked for together with the k-mer's own words: one wait for all of the read this turn needs */
                const uint32_t sm = d.sk.m, hashed = sm < 12 ? sm : 12;
                ahead_f = read_bases32(packed, cur + k - sm + 1);
                ahead_r = read_bases32(packed, cur + k + 1 - hashed);
                const uint32_t heavy_left = (where >> WALK_HEAVY_SHIFT) & WALK_HEAVY;
                STAT(st_short, heavy_left != 0);
                if (heavy_left) {
                    /* the k-mer before this one met its key's marker, and this one elects the same occurrence -- one base further along its
                       strand --: straight to its own sequence */
                    kk.pos = kk.rc ? kk.pos + 1 : kk.pos - 1;
                    table = true;
                    on = sk_kmer_key<W>(x, x_rc);
                    const uint32_t back = (where >> WALK_HEAVY_BACK_SHIFT) & 7u;
                    where = WALK_VISITED | WALK_COMPACT | (back << WALK_BACK_SHIFT) | (back << WALK_HEAVY_BACK_SHIFT) | ((heavy_left - 1) << WALK_HEAVY_SHIFT);
                } else {
                    kk = sk_key<W>(x, x_rc, k, sm);
                    table = sk_usable(d, kk);
                    on = kk.key;
                    where = (WALK_NO_RETURN << WALK_BACK_SHIFT) | (WALK_LASTS << WALK_LASTS_SHIFT);  // choice 0 of the key's own sequence, nothing to come back to
                }
                on_a = sk_hash_a(on);
            }
        }
        if constexpr (SK) {
            if (live && walking && (!(where & WALK_VISITED) || (where & WALK_HEAVY_PENDING))) {  // (a walk along the key's own sequence may end in a miss that stands for more than itself; one behind a heavy key's marker ends with a look at how long the key lasts)
                const uint32_t sm = d.sk.m, hashed = sm < 12 ? sm : 12;
                ahead_f = read_bases32(packed, cur + k - sm + 1);
                ahead_r = read_bases32(packed, cur + k + 1 - hashed);
            }
        }
        bool settled = false, found = false;
        if constexpr (SK) {
            /* ONE bucket a turn for every lane that is at a seed: the key's first, or the next of a walk -- fetched by the wave together
               (sk_stage_lines: the four lanes of a quad read one line with one instruction, ONE address translation a line; a lane
               reading its own line in four pieces asks for four, and at 70 G translation misses a second chip-wide that bounded the
               first version of this kernel) and examined out of LDS. A walk that goes on does so in the lane's next turn: no lane
               waits inside a turn for another lane's second bucket (what the walk cost as a loop inside the turn: 27 % of the kernel,
               profiles/r05/streaming_ablation_what_each_part_costs.txt). */
            const bool need = table || walking;
            const uint32_t c = where & WALK_CHOICE;
            const bool compact = (where & WALK_COMPACT) != 0;  // on the k-mer's own sequence: the k-mers' region, compact entries
            uint32_t bucket = 0;
            if (need) bucket = compact ? d.sk.num_buckets + sk_choice_of(on, on_a, c, d.sk.kmer_buckets) : sk_choice_of(on, on_a, c, d.sk.num_buckets);
            sk_stage_lines<W, true>(d, bucket, 0u, need, wave_stage, compact);  // (all 64 lanes, whatever their event)
            if (need) {
                const uint4* mine = wave_stage + (lane & 3u) * 64 + (lane >> 2) * 4;
                sk_query_t<W> Q = sk_make_query<W>(x, x_rc, kk, uint32_t(on_a) & 0xFFFFFFu);
                fast_t r = fast_unsettled(false);
                sk_bucket_flags flags;
                flags.go_on = 0;
                flags.second_used = false;
                bool marker = false, seen = false, inline_seen = false;
                uint32_t lasts = 0xFFFFu;
                if (!compact) {
                    sk_examine_slot_tracking<W, true, true>(d, Q, c, [mine](uint32_t i) { return mine[i]; }, r, seen, marker, flags, lasts, inline_seen);
                    STAT(st_slot1, r.outcome == FAST_MISS && flags.second_used);
                    if (r.outcome == FAST_MISS && flags.second_used) {
                        if constexpr (W == 1) {
                            sk_examine_slot_tracking<W, false, true>(d, Q, c, [mine](uint32_t i) { return mine[2 + i]; }, r, seen, marker, flags, lasts, inline_seen);
                        } else {  // slot 1 lives in the bucket's second line, and only the lanes whose key's fingerprint is there get here
                            const uint4* B1 = reinterpret_cast<const uint4*>(d.sk.slots) + (SK_BUCKET_SLOTS * 2 * W) * uint64_t(bucket) + 2 * W;
                            sk_examine_slot_tracking<W, false, true>(d, Q, c, [B1](uint32_t i) { return B1[i]; }, r, seen, marker, flags, lasts, inline_seen);
                        }
                    }
                } else if constexpr (W == 1) {
                    const uint32_t* words = reinterpret_cast<const uint32_t*>(mine);
                    sk_examine_kmer_line(Q, c, [words](uint32_t i) { return words[i]; }, r, flags);
                } else {
                    sk_examine_kmer_entry<true>(Q, c, [mine](uint32_t i) { return mine[i]; }, r, flags);
                    sk_examine_kmer_entry<false>(Q, c, [mine](uint32_t i) { return mine[2 + i]; }, r, flags);
                }
                /* what the slots of the key's sequence have allowed so far, this bucket's included (the buckets of the KEY's sequence are
                   the key's alone: which of them a walk sees does not depend on the k-mer) */
                {
                    const uint32_t before = (where >> WALK_LASTS_SHIFT) & WALK_LASTS;
                    lasts = lasts < before ? lasts : before;
                    where = (where & ~(WALK_LASTS << WALK_LASTS_SHIFT)) | (lasts << WALK_LASTS_SHIFT) | (inline_seen ? WALK_INLINE : 0u);
                }
                /* where the walk goes from here (lookup_device.hpp: sk_walk_step) */
                const bool go_on = flags.go_on != 0;
                bool more = false, defer = false;
                if (r.outcome == FAST_MISS) {
                    if (marker && !(where & WALK_VISITED)) {  // the key is heavy: its k-mers are entered under keys of their own
                        const uint32_t back = go_on ? c + 1 : WALK_NO_RETURN;
                        where = WALK_VISITED | WALK_COMPACT | (back << WALK_BACK_SHIFT) |
                                ((where & WALK_INLINE) ? 0u : (WALK_HEAVY_PENDING | (back << WALK_HEAVY_BACK_SHIFT)));  // (nothing inline on the way: the k-mers behind may skip it)
                        on = sk_kmer_key<W>(x, x_rc);
                        on_a = sk_hash_a(on);
                        more = true;
                    } else if (go_on) {
                        defer = c + 1 >= SK_CHOICES;  // a key (or k-mer) that found no slot: the complete path
                        where = (where & ~WALK_CHOICE) | (c + 1);
                        more = !defer;
                    } else if (compact && ((where >> WALK_BACK_SHIFT) & WALK_CHOICE) != WALK_NO_RETURN) {
                        /* the k-mer is not under its own key (the marker may have been another key's with an equal fingerprint): what is
                           left is the rest of the key's sequence */
                        const uint32_t back = (where >> WALK_BACK_SHIFT) & WALK_CHOICE;
                        defer = back >= SK_CHOICES;
                        where = WALK_VISITED | back | (WALK_NO_RETURN << WALK_BACK_SHIFT) | (where & WALK_HEAVY_FIELDS);
                        on = kk.key;
                        on_a = sk_hash_a(on);
                        more = !defer;
                    }
                }
                walking = more;
                if (!more && !defer) {
                    settled = true;
                    found = r.outcome == FAST_HIT;
                    off = r.kmer_offset;
                    ori = r.orientation;
                    const bool stands = !found && !(where & WALK_VISITED);
                    if (stands || (where & WALK_HEAVY_PENDING)) {
                        /* a miss that stands for the k-mers behind this one: those that elect the same key occurrence (sk_key_persists)
                           and still hold the base that keeps the read and the key's slots apart (`lasts`, over every bucket of the key's
                           sequence the walk has seen; no slot with the key: all of them) are negative as well -- counted, not looked at.
                           (A walk that met its key's marker went on along the K-MER's own sequence: its miss stands for itself.) */
                        uint64_t keep = sk_key_persists<W>(kk, k, d.sk.m, ahead_f, ahead_r);
                        keep = keep < valid_end - (cur + k) ? keep : valid_end - (cur + k);
                        if (stands) {
                            keep = keep < lasts ? keep : lasts;
                            c_negative += keep;
                            cur += keep;
#ifdef SSHASH_STREAM_STATS
                            st_kept_lane += keep;
#endif
                        } else {  // (a heavy key's first k-mer, settled on its own sequence -- found or not: that many k-mers behind it share the key)
                            where = (where & ~(WALK_HEAVY_PENDING | (WALK_HEAVY << WALK_HEAVY_SHIFT))) | (uint32_t(keep) << WALK_HEAVY_SHIFT);
                        }
                    }
                }
            }
        }
        const bool finishing = live && !walking;  // (a walking lane's seed is settled in a later turn)
        STAT(st_full, finishing && !settled);
        if (finishing && !settled) {
            /* no table, or a tie / an unplaced key / another shard's key: the complete seed() */
            const minimizer_t mf = compute_minimizer<W>(x, k, d.m, d.hash_magic);
            const minimizer_t mr = compute_minimizer<W>(x_rc, k, d.m, d.hash_magic);
            if (!SK && neg_unknown_mini && mf.value == prev_f && mr.value == prev_r) {  // :150-157
                found = false;
            } else {
                prev_f = mf.value;
                prev_r = mr.value;
                const hit_t h = seed_lookup<W, CANON>(d, skew, x, x_rc, mf, mr);
                found = h.found;
                off = h.kmer_offset;
                ori = h.orientation;
                neg_unknown_mini = !SK && !h.found && !h.minimizer_found;
            }
        }
        if (finishing) {
            if (found) {
                ++c_searches;
                neg_unknown_mini = false;
                pending = cur + 1 + k <= valid_end;  // (otherwise nothing can extend it)
            } else {
                ++c_negative;
            }
            ++cur;
        }
    }
#ifdef SSHASH_STREAM_STATS
    const unsigned long long st_neg_kept = wave_sum(st_kept_lane);
    if (lane == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(report) + 6;
        atomicAdd(st + 0, st_turns); atomicAdd(st + 1, st_fresh); atomicAdd(st + 2, st_walk); atomicAdd(st + 3, st_ext); atomicAdd(st + 4, st_slot1);
        atomicAdd(st + 5, st_inv); atomicAdd(st + 6, st_idle); atomicAdd(st + 7, st_full); atomicAdd(st + 8, st_neg_kept); atomicAdd(st + 9, st_short);
    }
#endif
    const bool first = lane == 0;  // (what the wave moved out is added once)
    block_report(first ? wave_moved_out[4] : 0, uint64_t(c_invalid) + (first ? wave_moved_out[0] : 0), uint64_t(c_negative) + (first ? wave_moved_out[1] : 0),
                 uint64_t(c_searches) + (first ? wave_moved_out[2] : 0), uint64_t(c_extensions) + (first ? wave_moved_out[3] : 0), report);
}

template <int W, bool CANON>
void launch_streaming_runs(device_replica const* rep, dict_view const& d, char const* bases, uint64_t const* offsets, uint64_t n_reads,
                           uint64_t total_bases, uint64_t* report, hipStream_t s) {
    /* two bits and a validity bit a base, in words of 32 and 64 bases; three words of slack behind the last base (a seed and a
       run read up to two words past their first) */
    const uint64_t packed_bytes = ((total_bases + 31) / 32 + 3) * 8, okay_bytes = ((total_bases + 63) / 64 + 2) * 8;
    /* (scratch the replica keeps for this stream, replica.hpp; two host threads that share a stream must not interleave their
       launch sequences, which share it) */
    std::lock_guard<std::mutex> sequence(rep->launch_mutex);
    uint64_t* packed = static_cast<uint64_t*>(rep->read_scratch_for(s, packed_bytes + okay_bytes));
    uint64_t* okay = packed + packed_bytes / 8;
    if (total_bases) {
        const uint64_t lanes = (total_bases + 7) / 8;
        hipLaunchKernelGGL(stream_pack_kernel, dim3(uint32_t((lanes + 255) / 256)), dim3(256), 0, s, bases, total_bases,
                           reinterpret_cast<uint16_t*>(packed), reinterpret_cast<uint8_t*>(okay));
    }
    /* waves: as many as the chip holds at once -- a lane that finishes its read takes the next of its wave's share, and the longer the
       share, the better the lanes of a wave even out --, fewer for a small call (a piece of a query file: some 10^4 reads, many calls
       side by side on their own streams), down to two reads a lane */
    const uint64_t max_waves = uint64_t(256) * 4 * SSHASH_STREAM_WAVES;  // (what the chip holds of this kernel: five waves a SIMD, 96 registers)  // (what the chip holds of this kernel: 96 registers at k <= 31, 106 at k <= 63)
    uint64_t waves = std::min<uint64_t>(max_waves, std::max<uint64_t>(1, n_reads / (64 * 2)));
    waves = (waves + 3) / 4 * 4;
    const uint64_t reads_per_wave = (n_reads + waves - 1) / waves;
    const dim3 grid(uint32_t(waves / 4)), block(256);
    const uint32_t move_out_every = uint32_t(test_hook_u64("stream_move_out_every", uint64_t(1) << 16, 1, uint64_t(1) << 16));
    if (d.sk.enabled) hipLaunchKernelGGL((streaming_run_kernel<W, CANON, true>), grid, block, 0, s, d, rep->d_skew, packed, okay, offsets, n_reads, reads_per_wave, move_out_every, report);
    else hipLaunchKernelGGL((streaming_run_kernel<W, CANON, false>), grid, block, 0, s, d, rep->d_skew, packed, okay, offsets, n_reads, reads_per_wave, move_out_every, report);
    HIP_CHECK(hipGetLastError());
}

}  // namespace

void engine::streaming_query_device(int device, char const* d_bases, uint64_t const* d_read_offsets, uint64_t n_reads,
                                    uint64_t total_bases, uint64_t* d_report, void* stream) const {
    device_replica const* rep = replica(device);
    if (n_reads == 0) return;
    device_guard guard(device);
    dict_view const& d = rep->view;
    hipStream_t s = hipStream_t(stream);
    const bool wide = d.k > 31;
    if (total_bases == 0) {
        /* the packing pass covers bases [0, read_offsets[n_reads]): a caller that does not say how many that is (the C ABI's device
           entry point) costs the launch one 8-byte read-back on its stream */
        HIP_CHECK(hipMemcpyAsync(&total_bases, d_read_offsets + n_reads, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (total_bases == 0) return;
    }
    if (!wide && !d.canonical) launch_streaming_runs<1, false>(rep, d, d_bases, d_read_offsets, n_reads, total_bases, d_report, s);
    else if (!wide && d.canonical) launch_streaming_runs<1, true>(rep, d, d_bases, d_read_offsets, n_reads, total_bases, d_report, s);
    else if (wide && !d.canonical) launch_streaming_runs<2, false>(rep, d, d_bases, d_read_offsets, n_reads, total_bases, d_report, s);
    else launch_streaming_runs<2, true>(rep, d, d_bases, d_read_offsets, n_reads, total_bases, d_report, s);
}

/* ---- per-k-mer results: the streaming query as a position-parallel pipeline -----------------------------------
   The reference asserts, for every k-mer, that its streaming result equals the point lookup
   (include/streaming_query.hpp:107); what the state machine adds is bookkeeping: a k-mer is invalid iff one of its k
   characters is (:59-65: an invalid one resets to full validation), a positive k-mer is an *extension* iff the
   previous k-mer of the read was positive, in the same string, and the id moved by that k-mer's orientation (:86-100;
   `remaining_string_bases > 0` is "same string"), otherwise a *search*; everything else is negative. So:
     1. encode    one lane per base of the reads: which read it lies in (a tile of 256 positions resolves its reads
                  through LDS), whether a k-mer starts there, its validity, the packed k-mer (characters staged in
                  LDS, four packed at a time);
     2. lookup    the batched lookup over all places holding a valid k-mer (engine.hip: first / resume / deferred
                  passes) -- every k-mer at full lookup speed, no serial chain along a read, long reads cost nothing
                  special;
     3. classify  one lane per k-mer: counters, and the default result for invalid k-mers. */
constexpr uint8_t SQ_VALID = 1, SQ_INVALID = 2, SQ_FIRST = 4;  // flags of a place; 0 = no k-mer starts here

__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t v) {  // 0x80 in every byte of v that is not zero
    return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
}

/* The read holding the first base of every tile of 256 places: one lane per TILE. (Looked up by lane 0 of the tile's own
   workgroup, these ~20 dependent loads were the lifetime of the workgroup: the encode pass was slower than the lookups.) */
__global__ void __launch_bounds__(256)
stream_tile_reads_kernel(const uint64_t* __restrict__ offsets, const uint64_t n_reads, const uint64_t first, const uint64_t tiles,
                         uint64_t* __restrict__ tile_read) {
    const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    const uint64_t g0 = first + t * 256;
    uint64_t lo = 0, hi = n_reads - 1;  // largest r with offsets[r] <= g0
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (offsets[mid] <= g0) lo = mid;
        else hi = mid - 1;
    }
    tile_read[t] = lo;
}

template <int W>
__global__ void __launch_bounds__(256)
stream_encode_kernel(const char* __restrict__ bases, const uint64_t* __restrict__ offsets, const uint64_t* __restrict__ tile_read, const uint64_t n_reads,
                     const uint64_t total_bases, const uint64_t first, const uint64_t count, const uint32_t k,
                     uint64_t* __restrict__ kmers /* relative to `first` */, uint8_t* __restrict__ flags /* absolute */) {
    constexpr uint32_t TILE = 256, SPAN = TILE + 64;
    __shared__ uint32_t chars[SPAN / 4 + 2];
    /* the same characters as 2-bit codes back to back, and one "is A, C, G or T" bit per character: every lane then cuts
       its k-mer and its k validity bits out of these with funnel shifts (packing k characters per lane from `chars`
       cost ~100 VALU instructions per position: the encode pass took longer than the lookups it feeds) */
    __shared__ uint32_t codes[SPAN / 16 + 4];
    __shared__ uint32_t okay[SPAN / 32 + 4];
    __shared__ uint64_t ends[TILE + 2];  // offsets[r0 + 1 ...]: the read boundaries that may fall into the tile
    const uint64_t g0 = first + uint64_t(blockIdx.x) * TILE;
    const uint32_t tid = threadIdx.x;
    const uint64_t r0 = tile_read[blockIdx.x];  // the read holding the tile's first base
    /* the characters of the tile and the k - 1 after it */
    for (uint32_t c = tid; c < SPAN / 4 + 2; c += TILE) {
        const uint64_t at = g0 + 4 * uint64_t(c);
        uint32_t v = 0;
        if (at + 4 <= total_bases && ((reinterpret_cast<uintptr_t>(bases) + at) & 3) == 0) v = *reinterpret_cast<const uint32_t*>(bases + at);
        else
            for (uint32_t b = 0; b < 4; ++b)
                if (at + b < total_bases) v |= uint32_t(uint8_t(bases[at + b])) << (8 * b);
        chars[c] = v;
    }
    __syncthreads();
    for (uint32_t c = tid; c < SPAN / 4 + 2; c += TILE) {
        const uint32_t four = chars[c];
        uint32_t two = (four >> 1) & 0x03030303u;  // (c >> 1) & 3 (include/kmer.hpp:118) for four characters at once
        two = (two | (two >> 6) | (two >> 12) | (two >> 18)) & 0xFFu;
        reinterpret_cast<uint8_t*>(codes)[c] = uint8_t(two);
    }
    for (uint32_t q = tid; q < SPAN + 64; q += TILE) {  // uniform over each wave: whole waves take part in the ballot
        /* A C G T a c g t only (include/kmer.hpp:209-219): fold the case */
        const uint32_t u = (q < 4 * (SPAN / 4 + 2) ? reinterpret_cast<const uint8_t*>(chars)[q] : 0u) & 0xDFu;
        const uint64_t mask = __ballot(u == 0x41u || u == 0x43u || u == 0x47u || u == 0x54u);
        if ((q & 63u) == 0) {
            okay[q / 32] = uint32_t(mask);
            okay[q / 32 + 1] = uint32_t(mask >> 32);
        }
    }
    for (uint32_t c = tid; c < TILE + 2; c += TILE) ends[c] = r0 + 1 + c <= n_reads ? offsets[r0 + 1 + c] : ~uint64_t(0);
    __syncthreads();
    const uint64_t p = g0 + tid;
    if (tid >= count - uint64_t(blockIdx.x) * TILE || p >= total_bases) return;
    /* my read: r0 + the number of boundaries ends[.] <= p */
    uint32_t lo = 0, hi = TILE + 1;  // first index with ends[index] > p
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (ends[mid] <= p) lo = mid + 1;
        else hi = mid;
    }
    uint64_t begin, end;
    if (lo <= TILE) {
        end = ends[lo];
        begin = lo ? ends[lo - 1] : offsets[r0];
    } else {  // more than TILE reads begin inside the tile (empty reads): resolve this place on its own
        uint64_t a = r0, b = n_reads - 1;
        while (a < b) {
            const uint64_t mid = a + (b - a + 1) / 2;
            if (offsets[mid] <= p) a = mid;
            else b = mid - 1;
        }
        begin = offsets[a];
        end = offsets[a + 1];
    }
    uint8_t f = 0;
    kmer_w<W> x = kmer_zero<W>();
    /* (a place behind the last read's end -- the caller's total_bases exceeds read_offsets[num_reads] -- belongs to no read: no
       k-mer starts there; ADVICE r2: it used to be looked up as part of one endless read) */
    if (end != ~uint64_t(0) && p + k <= end) {
        const uint32_t cw = tid >> 4, cs = 2 * (tid & 15u);
        uint32_t word[2 * W];
        for (int j = 0; j < 2 * W; ++j) word[j] = __builtin_amdgcn_alignbit(codes[cw + j + 1], codes[cw + j], cs);
        for (int j = 0; j < W; ++j) x.w[j] = uint64_t(word[2 * j]) | (uint64_t(word[2 * j + 1]) << 32);
        x = kmer_take_chars<W>(x, k);
        const uint32_t vw = tid >> 5, vs = tid & 31u;
        const uint64_t ok = uint64_t(__builtin_amdgcn_alignbit(okay[vw + 1], okay[vw], vs)) |
                            (uint64_t(__builtin_amdgcn_alignbit(okay[vw + 2], okay[vw + 1], vs)) << 32);
        const bool valid = (~ok & low_mask(k)) == 0;  // k <= 63
        f = (valid ? SQ_VALID : SQ_INVALID) | (p == begin ? SQ_FIRST : 0);
    }
    flags[p] = f;
    for (int j = 0; j < W; ++j) kmers[(p - first) * W + j] = x.w[j];
}

__global__ void __launch_bounds__(256)
stream_classify_kernel(const uint8_t* __restrict__ flags, const uint64_t total_bases, const result_view out,
                       const uint64_t* string_id, const int8_t* orientation /* may be out's own arrays */, uint64_t* __restrict__ report,
                       const bool has_predecessor /* the arrays continue below index 0: a later piece of one call */) {
    /* grid-stride: the six counters are accumulated in registers and reach `report` once per wave -- one hot set of
       atomics per wave of 64 k-mers would serialise at ~90 atomics/us (HISTORY.md) */
    uint64_t c_kmer = 0, c_invalid = 0, c_negative = 0, c_search = 0, c_extension = 0;
    /* Four consecutive places per lane and turn, everything they may need requested at once (what a branch does not use
       is ignored): one load after the other, each behind its test, made a chain of round trips per place, and the pass
       ran at a quarter of the memory rate. `flags` is this file's own array (4-byte loads); the others may be the caller's. */
    constexpr uint32_t PLACES = 4;
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x * PLACES;
    for (uint64_t p0 = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * PLACES; p0 < total_bases; p0 += stride) {
        const int64_t q = (p0 || has_predecessor) ? int64_t(p0) - 1 : 0;  // only the call's very first place has no predecessor
        uint8_t f[PLACES + 1];
        uint64_t id[PLACES + 1], sid[PLACES + 1];
        int8_t ori[PLACES + 1];
        f[0] = flags[q];
        id[0] = out.kmer_id[q];
        sid[0] = string_id[q];
        ori[0] = orientation[q];
        const bool whole = p0 + PLACES <= total_bases && ((reinterpret_cast<uintptr_t>(flags) + p0) & 3) == 0;
        const uint32_t four = whole ? *reinterpret_cast<const uint32_t*>(flags + p0) : 0u;
#pragma unroll
        for (uint32_t j = 0; j < PLACES; ++j) {
            const uint64_t p = p0 + j < total_bases ? p0 + j : total_bases - 1;
            f[j + 1] = whole ? uint8_t(four >> (8 * j)) : flags[p];
            id[j + 1] = out.kmer_id[p];
            sid[j + 1] = string_id[p];
            ori[j + 1] = orientation[p];
        }
#pragma unroll
        for (uint32_t j = 0; j < PLACES; ++j) {
            const uint64_t p = p0 + j;
            if (p >= total_bases) break;
            if (f[j + 1] & SQ_INVALID) {
                /* what streaming_query::lookup returns after its reset(): a default lookup_result (include/util.hpp:38-62) */
                ++c_kmer;
                ++c_invalid;
                out.kmer_id[p] = INVALID_U64;
                if (out.kmer_id_in_string) out.kmer_id_in_string[p] = INVALID_U64;
                if (out.kmer_offset) out.kmer_offset[p] = INVALID_U64;
                if (out.string_id) out.string_id[p] = INVALID_U64;
                if (out.string_begin) out.string_begin[p] = INVALID_U64;
                if (out.string_end) out.string_end[p] = INVALID_U64;
                if (out.kmer_orientation) out.kmer_orientation[p] = 1;
            } else if (f[j + 1] & SQ_VALID) {
                ++c_kmer;
                if (id[j + 1] == INVALID_U64) {
                    ++c_negative;
                } else {
                    const bool extension = !(f[j + 1] & SQ_FIRST) && (f[j] & SQ_VALID) && id[j] != INVALID_U64 && sid[j] == sid[j + 1] &&
                                           id[j + 1] == id[j] + uint64_t(int64_t(ori[j]));
                    c_extension += extension;
                    c_search += !extension;
                }
            }
        }
    }
    if (!report) return;  // uniform
    block_report(c_kmer, c_invalid, c_negative, c_search, c_extension, report);
}

void engine::streaming_lookup_device(int device, char const* d_bases, uint64_t const* d_read_offsets, uint64_t n_reads,
                                     uint64_t total_bases, result_view const& d_out, uint64_t* d_report, void* stream) const {
    device_replica const* rep = replica(device);
    if (!d_out.kmer_id && !d_report) throw error(error_kind::argument, "neither a kmer_id array nor a report to fill");
    if (d_out.minimizer_found) throw error(error_kind::argument, "the streaming lookup does not report minimizer_found");
    if (n_reads == 0 || total_bases == 0) return;
    device_guard guard(device);
    hipStream_t s = hipStream_t(stream);
    dict_view const& d = rep->view;
    const uint32_t W = d.k <= 31 ? 1 : 2;
    const uint64_t chunk = std::min<uint64_t>(total_bases, uint64_t(1) << 27);
    /* stream-ordered temporaries, given back on every exit path (an exception between allocation and release must not
       leak them: ADVICE r2) */
    struct temporaries {
        hipStream_t s;
        device_replica const* rep;
        std::vector<void*> owned;
        void* get(uint64_t bytes) {
            void* p = rep->stream_alloc(std::max<uint64_t>(bytes, 8), s);
            owned.push_back(p);
            return p;
        }
        ~temporaries() {
            for (void* p : owned) (void)hipFreeAsync(p, s);
        }
    } tmp{s, rep, {}};
    uint64_t* sid = d_out.string_id;
    int8_t* ori = d_out.kmer_orientation;
    uint64_t* ids = d_out.kmer_id;  // null: counters only (streaming_query_host over reads too long for one lane each)
    if (!ids) ids = static_cast<uint64_t*>(tmp.get(total_bases * sizeof(uint64_t)));
    uint8_t* flags = static_cast<uint8_t*>(tmp.get(total_bases));
    uint64_t* kmers = static_cast<uint64_t*>(tmp.get(chunk * W * sizeof(uint64_t)));
    uint64_t* tile_read = static_cast<uint64_t*>(tmp.get(((chunk + 255) / 256) * sizeof(uint64_t)));
    if (!sid) sid = static_cast<uint64_t*>(tmp.get(total_bases * sizeof(uint64_t)));
    if (!ori) ori = static_cast<int8_t*>(tmp.get(total_bases));
    result_view all = d_out;
    all.kmer_id = ids;
    all.string_id = sid;
    all.kmer_orientation = ori;
    for (uint64_t first = 0; first < total_bases; first += chunk) {
        const uint64_t count = std::min(chunk, total_bases - first);
        const dim3 grid(uint32_t((count + 255) / 256)), block(256);
        hipLaunchKernelGGL(stream_tile_reads_kernel, dim3((grid.x + 255) / 256), block, 0, s, d_read_offsets, n_reads, first, uint64_t(grid.x), tile_read);
        if (W == 1) hipLaunchKernelGGL(stream_encode_kernel<1>, grid, block, 0, s, d_bases, d_read_offsets, tile_read, n_reads, total_bases, first, count, d.k, kmers, flags);
        else hipLaunchKernelGGL(stream_encode_kernel<2>, grid, block, 0, s, d_bases, d_read_offsets, tile_read, n_reads, total_bases, first, count, d.k, kmers, flags);
        HIP_CHECK(hipGetLastError());
        result_view part = all;
        part.kmer_id += first;
        if (part.kmer_id_in_string) part.kmer_id_in_string += first;
        if (part.kmer_offset) part.kmer_offset += first;
        part.string_id += first;
        if (part.string_begin) part.string_begin += first;
        if (part.string_end) part.string_end += first;
        part.kmer_orientation += first;
        lookup_packed_masked_device(device, kmers, flags + first, count, true, out_mode::full, part, s);
    }
    for (uint64_t first = 0; first < total_bases; first += uint64_t(1) << 30) {
        const uint64_t count = std::min<uint64_t>(uint64_t(1) << 30, total_bases - first);
        result_view part = d_out;
        part.kmer_id = ids + first;
        if (part.kmer_id_in_string) part.kmer_id_in_string += first;
        if (part.kmer_offset) part.kmer_offset += first;
        if (part.string_id) part.string_id += first;
        if (part.string_begin) part.string_begin += first;
        if (part.string_end) part.string_end += first;
        if (part.kmer_orientation) part.kmer_orientation += first;
        hipLaunchKernelGGL(stream_classify_kernel, dim3(uint32_t(std::min<uint64_t>((count + 1023) / 1024, 2048))), dim3(256), 0, s, flags + first, count, part,
                           sid + first, ori + first, d_report, first != 0);
        HIP_CHECK(hipGetLastError());
    }
}

/* Host buffers: pieces of whole reads (at most ~64 MiB of bases each) go through one stream: H2D, the device pipeline
   above, D2H of the arrays the caller asked for. */
streaming_report engine::streaming_lookup_host(char const* bases, uint64_t const* read_offsets, uint64_t n_reads,
                                               result_view const& h_out) const {
    streaming_report total;
    if (n_reads == 0) return total;
    if (!h_out.kmer_id) throw error(error_kind::argument, "kmer_id output pointer is null");
    const std::vector<int> devs = devices();
    if (devs.empty()) throw error(error_kind::no_device, "dictionary is not resident on any device (call sshash_to_device first)");
    const int device = devs[0];
    device_guard guard(device);
    const uint64_t piece_bases = uint64_t(64) << 20;
    hipStream_t s = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct buffers {
        std::vector<void*> dev;
        hipStream_t s;
        ~buffers() {
            for (void* p : dev) (void)hipFree(p);
            (void)hipStreamDestroy(s);
        }
    } own{{}, s};
    auto dmalloc = [&](uint64_t bytes) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<uint64_t>(bytes, 8)));
        own.dev.push_back(p);
        return p;
    };
    uint64_t max_bases = 0, max_reads = 0;
    std::vector<uint64_t> cuts{0};
    for (uint64_t at = 0; at < n_reads;) {
        uint64_t end = at + 1;
        while (end < n_reads && read_offsets[end + 1] - read_offsets[at] <= piece_bases) ++end;
        max_bases = std::max(max_bases, read_offsets[end] - read_offsets[at]);
        max_reads = std::max(max_reads, end - at);
        cuts.push_back(end);
        at = end;
    }
    char* d_bases = static_cast<char*>(dmalloc(max_bases + 8));
    uint64_t* d_offsets = static_cast<uint64_t*>(dmalloc((max_reads + 1) * 8));
    uint64_t* d_report = static_cast<uint64_t*>(dmalloc(6 * 8));
    HIP_CHECK(hipMemsetAsync(d_report, 0, 48, s));
    result_view d_out{};
    d_out.kmer_id = static_cast<uint64_t*>(dmalloc(max_bases * 8));
    if (h_out.kmer_id_in_string) d_out.kmer_id_in_string = static_cast<uint64_t*>(dmalloc(max_bases * 8));
    if (h_out.kmer_offset) d_out.kmer_offset = static_cast<uint64_t*>(dmalloc(max_bases * 8));
    if (h_out.string_id) d_out.string_id = static_cast<uint64_t*>(dmalloc(max_bases * 8));
    if (h_out.string_begin) d_out.string_begin = static_cast<uint64_t*>(dmalloc(max_bases * 8));
    if (h_out.string_end) d_out.string_end = static_cast<uint64_t*>(dmalloc(max_bases * 8));
    if (h_out.kmer_orientation) d_out.kmer_orientation = static_cast<int8_t*>(dmalloc(max_bases));
    if (h_out.minimizer_found) throw error(error_kind::argument, "the streaming lookup does not report minimizer_found");
    std::vector<uint64_t> rel(max_reads + 1);
    for (size_t piece = 0; piece + 1 < cuts.size(); ++piece) {
        const uint64_t first = cuts[piece], last = cuts[piece + 1];
        const uint64_t b0 = read_offsets[first], nb = read_offsets[last] - b0;
        if (nb == 0) continue;
        for (uint64_t i = first; i <= last; ++i) rel[i - first] = read_offsets[i] - b0;
        HIP_CHECK(hipMemcpyAsync(d_offsets, rel.data(), (last - first + 1) * 8, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d_bases, bases + b0, nb, hipMemcpyHostToDevice, s));
        /* places without a k-mer keep what the caller's arrays hold (sshash_amd.h: "left untouched"): EVERY requested
           device array is seeded with the caller's (only kmer_id was, the others came back as stale device memory) */
        auto seed = [&](auto* h, auto* dptr, uint64_t width) {
            if (h) HIP_CHECK(hipMemcpyAsync(dptr, h + b0, nb * width, hipMemcpyHostToDevice, s));
        };
        seed(h_out.kmer_id, d_out.kmer_id, 8);
        seed(h_out.kmer_id_in_string, d_out.kmer_id_in_string, 8);
        seed(h_out.kmer_offset, d_out.kmer_offset, 8);
        seed(h_out.string_id, d_out.string_id, 8);
        seed(h_out.string_begin, d_out.string_begin, 8);
        seed(h_out.string_end, d_out.string_end, 8);
        seed(h_out.kmer_orientation, d_out.kmer_orientation, 1);
        streaming_lookup_device(device, d_bases, d_offsets, last - first, nb, d_out, d_report, s);
        HIP_CHECK(hipMemcpyAsync(h_out.kmer_id + b0, d_out.kmer_id, nb * 8, hipMemcpyDeviceToHost, s));
        auto back = [&](auto* h, auto* dptr, uint64_t width) {
            if (h) HIP_CHECK(hipMemcpyAsync(h + b0, dptr, nb * width, hipMemcpyDeviceToHost, s));
        };
        back(h_out.kmer_id_in_string, d_out.kmer_id_in_string, 8);
        back(h_out.kmer_offset, d_out.kmer_offset, 8);
        back(h_out.string_id, d_out.string_id, 8);
        back(h_out.string_begin, d_out.string_begin, 8);
        back(h_out.string_end, d_out.string_end, 8);
        back(h_out.kmer_orientation, d_out.kmer_orientation, 1);
        HIP_CHECK(hipStreamSynchronize(s));
    }
    uint64_t h[6];
    HIP_CHECK(hipMemcpyAsync(h, d_report, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    total.num_kmers = h[0];
    total.num_positive_kmers = h[1];
    total.num_negative_kmers = h[2];
    total.num_invalid_kmers = h[3];
    total.num_searches = h[4];
    total.num_extensions = h[5];
    return total;
}

/* Host buffers: the reads are cut into pieces of at most ~32 MiB of bases; per replica up to eight lanes (the
   pooled pinned pipelines of the lookup host path, replica.hpp) pull pieces from a shared counter and run
   copy-in -> H2D -> kernel, accumulating the six counters in device memory; one read-back per lane. */
constexpr uint64_t LONG_READ_BASES = uint64_t(1) << 16;

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
                /* one lane walks one read: a read of megabases (a contig, a multiline FASTA record) would keep a single
                   lane busy for minutes; such pieces go through the position-parallel pipeline, which gives the same counters */
                bool long_read = false;
                for (uint64_t i = first; i < last && !long_read; ++i) long_read = read_offsets[i + 1] - read_offsets[i] > LONG_READ_BASES;
                if (long_read) streaming_lookup_device(device, dp + bases_at, reinterpret_cast<uint64_t const*>(dp), last - first, nb, result_view{}, d_report, s);
                else streaming_query_device(device, dp + bases_at, reinterpret_cast<uint64_t const*>(dp), last - first, nb, d_report, s);
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

/* The file query of an uncompressed FASTQ (engine.hpp). One reader thread feeding batches through streaming_query_host splits
   9 GB/s of file and the devices wait for it nine tenths of the time (HISTORY.md); here parsing is the lanes' own
   work: lane = host thread + stream + pinned block + device block, as many lanes as usable CPUs, dealt round-robin to the
   resident replicas. */
bool engine::streaming_query_fastq_pieces(std::string const& filename, streaming_report& total) const {
    const std::vector<int> devs = devices();
    if (devs.empty()) throw error(error_kind::no_device, "dictionary is not resident on any device (call sshash_to_device first)");
    const uint64_t G = devs.size();
    uint64_t want_lanes = std::max<uint64_t>(G, std::min<uint64_t>(32, usable_cpus()));
    if (char const* e = std::getenv("SSHASH_AMD_READER_THREADS")) want_lanes = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));
    /* pieces of 32 MiB of file; a smaller file is cut finer, so that every lane still gets several (4 MiB at least) */
    uint64_t piece_bytes = uint64_t(32) << 20;
    {
        struct stat st;
        if (stat(filename.c_str(), &st) == 0)
            piece_bytes = std::min<uint64_t>(piece_bytes, std::max<uint64_t>(uint64_t(4) << 20, uint64_t(st.st_size) / (want_lanes * 8)));
    }
    piece_bytes = test_hook_u64("fastq_piece_bytes", piece_bytes, 4096, ~uint64_t(0));  // (tests: many pieces of a small file)
    const fastq_pieces file(filename, piece_bytes);
    const uint64_t num_pieces = file.num_pieces();
    if (num_pieces == 0) return true;  // an empty file: an empty report
    const uint32_t k = m_idx->k;
    const uint64_t off_capacity = file.offsets_capacity(k), bases_capacity = file.bases_capacity();
    const uint64_t bases_at = (off_capacity * sizeof(uint64_t) + 255) & ~uint64_t(255);
    const uint64_t report_at = (bases_at + bases_capacity + 255) & ~uint64_t(255);
    const uint64_t lane_bytes = report_at + 6 * sizeof(uint64_t);
    const uint64_t num_lanes = std::min(num_pieces, want_lanes);

    std::atomic<uint64_t> next{0};
    std::atomic<bool> give_up{false};
    std::vector<fastq_pieces::parsed> seen(num_pieces);
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
            std::vector<char> raw;
            for (;;) {
                const uint64_t piece = next.fetch_add(1);
                if (piece >= num_pieces || give_up) break;
                uint64_t* offsets = reinterpret_cast<uint64_t*>(hp);
                const fastq_pieces::parsed got = file.parse(piece, k, hp + bases_at, bases_capacity, offsets, off_capacity, raw);
                seen[piece] = got;
                if (got.overflow) {
                    give_up = true;
                    break;
                }
                if (got.num_reads == 0) continue;
                HIP_CHECK(hipMemcpyAsync(dp, hp, (got.num_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
                HIP_CHECK(hipMemcpyAsync(dp + bases_at, hp + bases_at, got.num_bases, hipMemcpyHostToDevice, s));
                bool long_read = false;  // (as streaming_query_host: a read of megabases must not sit on one lane of a wave)
                for (uint64_t i = 0; i < got.num_reads && !long_read; ++i) long_read = offsets[i + 1] - offsets[i] > LONG_READ_BASES;
                if (long_read) streaming_lookup_device(device, dp + bases_at, reinterpret_cast<uint64_t const*>(dp), got.num_reads, got.num_bases, result_view{}, d_report, s);
                else streaming_query_device(device, dp + bases_at, reinterpret_cast<uint64_t const*>(dp), got.num_reads, got.num_bases, d_report, s);
                HIP_CHECK(hipStreamSynchronize(s));  // the pinned block is parsed into again
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
        } catch (...) {
            errors[li] = std::current_exception();
            give_up = true;
        }
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
    if (give_up) return false;
    /* every piece began where its predecessor stopped, the first at 0, the last stopped at the end of the file: the pieces
       together are the sequential reader's records (reads.hpp) */
    uint64_t expect = 0;
    for (uint64_t i = 0; i < num_pieces; ++i) {
        if (seen[i].first_record != expect) return false;
        expect = seen[i].next_record;
    }
    if (expect != file.file_bytes()) return false;
    for (auto const& p : partial) {
        total.num_kmers += p.num_kmers;
        total.num_positive_kmers += p.num_positive_kmers;
        total.num_negative_kmers += p.num_negative_kmers;
        total.num_invalid_kmers += p.num_invalid_kmers;
        total.num_searches += p.num_searches;
        total.num_extensions += p.num_extensions;
    }
    return true;
}

}  // namespace sshash_amd
