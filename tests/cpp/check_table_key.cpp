// check_table_key.cpp -- host-side properties of the super-k-mer table's key function (csrc/device_layout.hpp):
// a k-mer and its reverse complement must elect the same m-mer occurrence, or the table could not serve both
// strands with one slot. Plain g++, no GPU. Prints "OK <checked> <ties>" or the first counter-example.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../sshash_amd/csrc/device_layout.hpp"

using namespace sshash_amd;

template <int W>
static int check(uint32_t k, uint32_t m, uint64_t trials, std::mt19937_64& rng, uint64_t& ties) {
    const uint64_t mask = low_mask(2 * m);
    for (uint64_t t = 0; t < trials; ++t) {
        kmer_w<W> x;
        for (int j = 0; j < W; ++j) x.w[j] = rng();
        if (t % 7 == 0) x.w[0] &= 0x3333333333333333ULL;  // low-complexity k-mers: repeated m-mers inside one window
        x = kmer_take_chars<W>(x, k);
        const kmer_w<W> y = kmer_revcomp<W>(x, k);
        const sk_key_t a = sk_key<W>(x, y, k, m), b = sk_key<W>(y, x, k, m);
        if (a.tie != b.tie) return printf("tie flag differs between strands (k=%u m=%u)\n", k, m), 1;
        if (a.tie) {
            ++ties;
            continue;
        }
        if (a.key != b.key) return printf("different keys for the two strands (k=%u m=%u)\n", k, m), 1;
        if (a.rc == b.rc) return printf("both strands claim the same orientation (k=%u m=%u)\n", k, m), 1;
        if (a.pos != b.pos) return printf("different positions on the winning strand (k=%u m=%u)\n", k, m), 1;
        if (a.pos > k - m) return printf("position outside the k-mer (k=%u m=%u)\n", k, m), 1;
        const kmer_w<W>& winner = a.rc ? y : x;
        if ((kmer_shr_chars<W>(winner, a.pos).w[0] & mask) != a.key) return printf("key is not the m-mer at its position\n"), 1;
        /* leftmost among equal hashes on the winning strand: the election hashes the first min(m, 12) bases of an occurrence
           (device_layout.hpp: sk_select_hash; m < 12: the window shifted left by 24 - 2m bits) and compares the 26-bit hashes */
        auto hash_at = [&](kmer_w<W> const& strand, uint32_t i) {
            uint32_t word = uint32_t(kmer_shr_chars<W>(strand, i).w[0]) ^ sk_select_salt<W>();
            if (m < 12) word <<= 24 - 2 * m;
            return sk_select_hash(word, 0u, sk_select_mul()) >> SK_POS_BITS;
        };
        const uint32_t won = hash_at(winner, a.pos);
        for (uint32_t i = 0; i + m <= k; ++i) {
            const uint32_t h = hash_at(winner, i);
            if (i < a.pos ? h <= won : h < won)
                return printf("another m-mer of the winning strand should have been elected (k=%u m=%u)\n", k, m), 1;
        }
        const kmer_w<W>& loser = a.rc ? x : y;
        for (uint32_t i = 0; i + m <= k; ++i)
            if (hash_at(loser, i) <= won)
                return printf("an m-mer of the other strand should have been elected (k=%u m=%u)\n", k, m), 1;
    }
    return 0;
}

/* sk_key_persists along reads: for every k-mer whose key is not a tie, the t k-mers that follow must elect the very same occurrence
   (same key, same strand, the position moved by t); how far it looks is printed next to how far the key really lasts */
template <int W>
static int check_persists(uint32_t k, uint32_t m, uint64_t reads, std::mt19937_64& rng, uint64_t& kmers, uint64_t& claimed, uint64_t& lasted) {
    const uint32_t L = m < 12 ? m : 12;
    for (uint64_t t = 0; t < reads; ++t) {
        const uint32_t len = k + 40 + uint32_t(rng() % 160);
        const uint32_t alphabet = t % 5 == 0 ? 2 : 4;
        std::vector<uint8_t> read(len + 80);
        for (uint32_t j = 0; j < read.size(); ++j) read[j] = uint8_t(t % 11 == 0 && (j / 40) % 2 ? 0 : rng() % alphabet);
        auto bases32 = [&](uint32_t from) {  // (what read_bases32 of streaming.hip hands over: whatever lies there, also behind the read's end)
            uint64_t w = 0;
            for (uint32_t i = 0; i < 32; ++i) w |= uint64_t(from + i < read.size() ? read[from + i] : 0) << (2 * i);
            return w;
        };
        auto kmer_at = [&](uint32_t from) {
            kmer_w<W> x = kmer_zero<W>();
            for (uint32_t i = 0; i < k; ++i) x = kmer_roll<W>(x, read[from + i], k);
            return x;
        };
        for (uint32_t cur = 0; cur + k <= len; ++cur) {
            const kmer_w<W> x = kmer_at(cur);
            const sk_key_t a = sk_key<W>(x, kmer_revcomp<W>(x, k), k, m);
            if (a.tie) continue;
            ++kmers;
            const uint32_t n = sk_key_persists<W>(a, k, m, bases32(cur + k - m + 1), bases32(cur + k + 1 - L));
            if (n > k - m) return printf("sk_key_persists claims more k-mers than a key can last (k=%u m=%u)\n", k, m), 1;
            claimed += n;
            for (uint32_t s = 1; s <= k - m && cur + s + k <= len; ++s) {
                const kmer_w<W> z = kmer_at(cur + s);
                const sk_key_t b = sk_key<W>(z, kmer_revcomp<W>(z, k), k, m);
                const bool same = !b.tie && b.rc == a.rc && b.key == a.key && b.pos == (a.rc ? a.pos + s : a.pos - s) && b.hash == a.hash;
                if (!same) {
                    if (s <= n) return printf("sk_key_persists overstates (k=%u m=%u read %llu k-mer %u: claims %u, the key changes after %u)\n", k, m,
                                              (unsigned long long)t, cur, n, s - 1), 1;
                    break;
                }
                ++lasted;
                if (a.rc ? a.pos + s == k - m : a.pos == s) break;  // the occurrence leaves the k-mer next
            }
        }
    }
    return 0;
}

int main() {
    std::mt19937_64 rng(12345);
    uint64_t ties = 0, checked = 0;
    const uint32_t cases[][2] = {{31, 21}, {31, 13}, {15, 7}, {21, 21}, {31, 1}, {63, 25}, {63, 31}, {47, 20}, {33, 5}};
    for (auto const& c : cases) {
        const uint64_t trials = 200000;
        const uint64_t before = ties;
        const int bad = c[0] <= 31 ? check<1>(c[0], c[1], trials, rng, ties) : check<2>(c[0], c[1], trials, rng, ties);
        if (bad) return 1;
        checked += trials;
        fprintf(stderr, "k=%u m=%u: %llu ties in %llu k-mers\n", c[0], c[1], (unsigned long long)(ties - before), (unsigned long long)trials);
    }
    for (auto const& c : cases) {
        uint64_t seen = 0, claimed = 0, lasted = 0;
        if (c[0] <= 31 ? check_persists<1>(c[0], c[1], 300, rng, seen, claimed, lasted) : check_persists<2>(c[0], c[1], 300, rng, seen, claimed, lasted)) return 1;
        fprintf(stderr, "sk_key_persists k=%u m=%u: %llu k-mers, %.2f following k-mers claimed per k-mer of %.2f the key lasts (read ends included)\n", c[0], c[1],
                (unsigned long long)seen, double(claimed) / double(seen ? seen : 1), double(lasted) / double(seen ? seen : 1));
    }
    /* bucket hashing: every choice inside the table */
    for (uint64_t key = 1; key < 100000; key += 7) {
        const sk_hash_t h = sk_hash(key * 0x9E3779B97F4A7C15ULL >> 22, 1000003u);
        for (uint32_t c = 0; c < SK_CHOICES; ++c)
            if (h.bucket[c] >= 1000003u) return printf("bucket out of range\n"), 1;
        if (h.fingerprint >> 24) return printf("fingerprint wider than 24 bits\n"), 1;
        /* the choices computed one by one (sk_choice_of: the streaming query's walk) are sk_hash's */
        const uint64_t k64 = key * 0x9E3779B97F4A7C15ULL >> 22, a = sk_hash_a(k64);
        if ((uint32_t(a) & 0xFFFFFFu) != h.fingerprint) return printf("sk_hash_a's fingerprint differs from sk_hash's\n"), 1;
        for (uint32_t c = 0; c < SK_CHOICES; ++c)
            if (sk_choice_of(k64, a, c, 1000003u) != h.bucket[c]) return printf("sk_choice_of differs from sk_hash (choice %u)\n", c), 1;
    }
    printf("OK %llu %llu\n", (unsigned long long)checked, (unsigned long long)ties);
    return 0;
}
