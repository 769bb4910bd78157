// reads.cpp -- see reads.hpp.
#include "reads.hpp"

#include "errors.hpp"

#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <stdexcept>
#include <thread>

namespace sshash_amd {

namespace {

bool ends_with(std::string const& s, char const* suffix) {
    const size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

/* ---- BGZF (bgzip, htslib) -------------------------------------------------------------------------------------
   A plain .gz is one deflate stream: it inflates on one thread, at zlib's 0.28 G bases/s, and that is what the file query
   then runs at (HISTORY.md). A BGZF file -- what bgzip writes, a series of gzip members of at most 64 KiB each
   whose header says how long the member is (extra subfield 'B','C': BSIZE) -- needs no decoding to find its members, so
   they are inflated side by side: a group of members is read, their sizes are in their last four bytes (ISIZE), every
   worker inflates a share of them into its place of the group's output, CRC-checked; the next group is decoded while the
   lines of this one are being split. Recognised by its first header; anything else goes through gzread as before. */
struct bgzf_source {
    FILE* f = nullptr;
    unsigned threads = 1;
    struct group {
        std::vector<unsigned char> packed;  // the members, back to back
        std::unique_ptr<char[]> out;        // (not a vector: no zero-fill of what inflate is about to write)
        size_t out_size = 0, out_capacity = 0;
        size_t taken = 0;
        bool last = false;
    };
    std::future<std::unique_ptr<group>> ahead;
    std::unique_ptr<group> cur;

    static constexpr size_t GROUP_PACKED = size_t(8) << 20;  // compressed bytes per group (a few hundred members)

    static bool is_bgzf_header(unsigned char const* h) {
        /* FLG = FEXTRA alone, XLEN = 6: what bgzip writes (a name or comment field would move the deflate data; not BGZF as specified) */
        return h[0] == 31 && h[1] == 139 && h[2] == 8 && h[3] == 4 && h[10] == 6 && h[11] == 0 && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
    }
    /* does the file start with a BGZF member? */
    static bool probe(std::string const& filename) {
        FILE* g = fopen(filename.c_str(), "rb");
        if (!g) return false;
        unsigned char h[18];
        const bool yes = fread(h, 1, 18, g) == 18 && is_bgzf_header(h);
        fclose(g);
        return yes;
    }
    explicit bgzf_source(std::string const& filename) {
        f = fopen(filename.c_str(), "rb");
        if (!f) throw error(error_kind::io, "error in opening the file '" + filename + "'");
        threads = std::max(1u, std::min(16u, usable_cpus()));
        if (char const* e = std::getenv("SSHASH_AMD_READER_THREADS")) threads = unsigned(std::max(1l, std::min(64l, std::atol(e))));
        ahead = std::async(std::launch::async, [this] { return decode_group(nullptr); });
    }
    ~bgzf_source() {
        if (ahead.valid()) {
            try { ahead.get(); } catch (...) {}
        }
        if (f) fclose(f);
    }
    bgzf_source(bgzf_source const&) = delete;
    bgzf_source& operator=(bgzf_source const&) = delete;

    /* `g`: a group whose bytes have all been handed out, to be filled again (its buffers are kept: a fresh 40 MB block per group
       is 10 000 page faults per group) */
    std::unique_ptr<group> decode_group(std::unique_ptr<group> g) {
        if (!g) g = std::make_unique<group>();
        g->packed.clear();
        g->taken = 0;
        g->last = false;
        struct member { size_t at, size, out_at, out_size; };
        std::vector<member> members;
        size_t out_total = 0;
        while (g->packed.size() < GROUP_PACKED) {
            unsigned char h[18];
            const size_t got = fread(h, 1, 18, f);
            if (got == 0) { g->last = true; break; }
            if (got != 18 || !is_bgzf_header(h)) throw error(error_kind::io, "error while reading the query file: not a BGZF member where one was expected");
            const size_t size = size_t(h[16]) + (size_t(h[17]) << 8) + 1;  // whole member, header and trailer included
            if (size < 18 + 8) throw error(error_kind::io, "error while reading the query file: BGZF member too short");
            const size_t at = g->packed.size();
            g->packed.resize(at + size);
            memcpy(g->packed.data() + at, h, 18);
            if (fread(g->packed.data() + at + 18, 1, size - 18, f) != size - 18) throw error(error_kind::io, "error while reading the query file: truncated BGZF member");
            unsigned char const* t = g->packed.data() + at + size - 4;
            const size_t isize = size_t(t[0]) | (size_t(t[1]) << 8) | (size_t(t[2]) << 16) | (size_t(t[3]) << 24);
            /* a BGZF member holds at most 64 KiB of input (SAM specification 4.1); the trailer is untrusted: a larger claim is a
               corrupt file, not a reason to allocate gigabytes */
            if (isize > (size_t(1) << 16)) throw error(error_kind::io, "error while reading the query file: corrupt BGZF member");
            members.push_back({at, size, out_total, isize});
            out_total += isize;
        }
        /* (also when nothing is to be written: a group of empty members only -- the 28-byte end-of-file marker alone in a fresh
           group -- must hand inflate a valid pointer) */
        if (!g->out || out_total > g->out_capacity) {
            g->out_capacity = out_total + out_total / 4 + 1;
            g->out.reset(new char[g->out_capacity]);
        }
        g->out_size = out_total;
        std::atomic<size_t> next{0};
        std::atomic<bool> bad{false};
        auto work = [&] {
            z_stream z;
            memset(&z, 0, sizeof(z));
            if (inflateInit2(&z, -15) != Z_OK) { bad = true; return; }  // raw deflate: the member's header and trailer are handled here
            for (;;) {
                const size_t first = next.fetch_add(16);  // 16 members = 1 MB of output at a time
                if (first >= members.size() || bad) break;
                for (size_t i = first; i < std::min(first + 16, members.size()); ++i) {
                    member const& m = members[i];
                    unsigned char const* p = g->packed.data() + m.at;
                    inflateReset(&z);
                    z.next_in = const_cast<unsigned char*>(p + 18);
                    z.avail_in = unsigned(m.size - 18 - 8);
                    z.next_out = reinterpret_cast<unsigned char*>(g->out.get() + m.out_at);
                    z.avail_out = unsigned(m.out_size);
                    const int rc = inflate(&z, Z_FINISH);
                    unsigned char const* t = p + m.size - 8;
                    const uint32_t crc = uint32_t(t[0]) | (uint32_t(t[1]) << 8) | (uint32_t(t[2]) << 16) | (uint32_t(t[3]) << 24);
                    if (rc != Z_STREAM_END || z.avail_out != 0 ||
                        uint32_t(crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<unsigned char const*>(g->out.get() + m.out_at), unsigned(m.out_size))) != crc)
                        bad = true;
                }
            }
            inflateEnd(&z);
        };
        const unsigned n = unsigned(std::min<size_t>(threads, (members.size() + 15) / 16));
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n; ++t) pool.emplace_back(work);
        if (!members.empty()) work();
        for (auto& t : pool) t.join();
        if (bad) throw error(error_kind::io, "error while reading the query file: corrupt BGZF member");
        return g;
    }
    /* up to `room` inflated bytes; 0 = end of file */
    size_t read(char* dst, size_t room) {
        for (;;) {
            if (cur && cur->taken < cur->out_size) {
                const size_t n = std::min(room, cur->out_size - cur->taken);
                memcpy(dst, cur->out.get() + cur->taken, n);
                cur->taken += n;
                return n;
            }
            if (cur && cur->last) return 0;
            std::unique_ptr<group> done = std::move(cur);
            cur = ahead.get();  // rethrows a worker's error
            if (!cur->last) {
                group* recycled = done.release();
                ahead = std::async(std::launch::async, [this, recycled] { return decode_group(std::unique_ptr<group>(recycled)); });
            }
        }
    }
};

/* Lines of a (possibly gzip-compressed) file: the file is inflated a block of megabytes at a time (gzread; BGZF: above) and
   split with memchr; a line is handed out as a view into the block -- no per-line gzgets, no per-line std::string. (Round 2
   assembled every line with gzgets into a std::string: 0.2 GB/s on uncompressed FASTQ, below zlib's own inflate rate; this
   reader splits at several GB/s, so an uncompressed file is read at the page cache's pace and a .gz at zlib's.) */
struct line_reader {
    gzFile f = nullptr;
    std::unique_ptr<bgzf_source> bgzf;
    std::vector<char> buf;
    size_t begin = 0, end = 0;
    bool eof = false;
    explicit line_reader(std::string const& filename) : buf(size_t(8) << 20) {
        if (bgzf_source::probe(filename)) {
            bgzf = std::make_unique<bgzf_source>(filename);
            return;
        }
        f = gzopen(filename.c_str(), "rb");
        if (!f) throw error(error_kind::io, "error in opening the file '" + filename + "'");
        gzbuffer(f, 1 << 20);
    }
    ~line_reader() {
        if (f) gzclose(f);
    }
    line_reader(line_reader const&) = delete;
    line_reader& operator=(line_reader const&) = delete;
    /* std::getline semantics: false once nothing at all could be read; the view (without the '\n') is valid until the next call */
    bool next(char const*& p, size_t& n) {
        for (;;) {
            if (begin < end) {
                if (void const* nl = memchr(buf.data() + begin, '\n', end - begin)) {
                    p = buf.data() + begin;
                    n = size_t(static_cast<char const*>(nl) - p);
                    begin += n + 1;
                    return true;
                }
            }
            if (eof) {
                if (begin == end) return false;
                p = buf.data() + begin;  // a last line without '\n'
                n = end - begin;
                begin = end;
                return true;
            }
            if (begin > 0) {  // keep the unfinished line, refill behind it
                memmove(buf.data(), buf.data() + begin, end - begin);
                end -= begin;
                begin = 0;
            }
            if (end == buf.size()) buf.resize(buf.size() * 2);  // a line longer than the block (a chromosome on one line)
            const size_t room = std::min<size_t>(buf.size() - end, size_t(1) << 30);
            if (bgzf) {
                const size_t got = bgzf->read(buf.data() + end, room);
                if (got == 0) eof = true;
                end += got;
                continue;
            }
            const int got = gzread(f, buf.data() + end, unsigned(room));
            if (got < 0) throw error(error_kind::io, "error while reading the query file");
            if (got == 0) eof = true;
            end += size_t(got);
        }
    }
    bool skip() {
        char const* p;
        size_t n;
        return next(p, n);
    }
};

void push_read(read_batch& out, char const* s, size_t n, uint32_t k) {
    if (n < k) return;
    out.bases.insert(out.bases.end(), s, s + n);
    out.offsets.push_back(out.bases.size());
}

}  // namespace

struct read_stream::impl {
    line_reader in;
    enum { FASTQ, FASTA, FASTA_MULTILINE } format;
    uint32_t k;
    bool done = false;
    std::string segment;
    impl(std::string const& filename, int fmt, uint32_t k_) : in(filename), format(decltype(format)(fmt)), k(k_) {}
};

read_stream::read_stream(std::string const& filename, bool multiline, uint32_t k) {
    const bool fasta = ends_with(filename, ".fa") || ends_with(filename, ".fasta") || ends_with(filename, ".fa.gz") ||
                       ends_with(filename, ".fasta.gz");
    const bool fastq = ends_with(filename, ".fq") || ends_with(filename, ".fastq") || ends_with(filename, ".fq.gz") ||
                       ends_with(filename, ".fastq.gz");
    if (!fasta && !fastq) {
        /* the reference opens the file before looking at the extension (src/query.cpp:127-128) */
        line_reader probe(filename);
        return;
    }
    m = std::make_unique<impl>(filename, fastq ? impl::FASTQ : multiline ? impl::FASTA_MULTILINE : impl::FASTA, k);
}

read_stream::~read_stream() = default;

bool read_stream::next(read_batch& out, uint64_t max_bases) {
    out.bases.clear();
    out.offsets.assign(1, 0);
    if (!m || m->done) return false;
    impl& r = *m;
    if (max_bases < (uint64_t(1) << 32)) out.bases.reserve(size_t(max_bases) + (size_t(1) << 16));
    char const* p = nullptr;
    size_t n = 0;
    while (out.bases.size() < max_bases) {
        if (r.format == impl::FASTQ) {
            if (!r.in.skip() || !r.in.next(p, n)) { r.done = true; break; }  // header, bases
            push_read(out, p, n, r.k);
            r.in.skip();  // '+'
            r.in.skip();  // qualities
        } else if (r.format == impl::FASTA) {
            if (!r.in.skip() || !r.in.next(p, n)) { r.done = true; break; }
            push_read(out, p, n, r.k);
        } else {
            if (!r.in.next(p, n)) {
                push_read(out, r.segment.data(), r.segment.size(), r.k);
                r.segment.clear();
                r.done = true;
                break;
            }
            if (n == 0) {
                push_read(out, r.segment.data(), r.segment.size(), r.k);
                r.segment.clear();
            } else {
                r.segment.append(p, n);
            }
        }
    }
    return out.num_reads() != 0 || !r.done;
}

unsigned usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::max(1, CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        unsigned long long period = 0;
        if (fscanf(f, "%31s %llu", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const unsigned long long q = strtoull(quota, nullptr, 10);
            n = unsigned(std::max<unsigned long long>(1, std::min<unsigned long long>(n, q / period)));
        }
        fclose(f);
    }
    return n;
}

/* ---- plain FASTQ in pieces (reads.hpp) ------------------------------------------------------------------------ */

bool fastq_pieces::applicable(std::string const& filename) {
    if (!(ends_with(filename, ".fq") || ends_with(filename, ".fastq"))) return false;
    FILE* f = fopen(filename.c_str(), "rb");
    if (!f) return false;
    unsigned char h[2] = {0, 0};
    const size_t got = fread(h, 1, 2, f);
    fclose(f);
    return !(got == 2 && h[0] == 31 && h[1] == 139);
}

fastq_pieces::fastq_pieces(std::string const& filename, uint64_t piece_bytes) : piece_(std::max<uint64_t>(piece_bytes, 4096)) {
    fd_ = ::open(filename.c_str(), O_RDONLY);
    if (fd_ < 0) throw error(error_kind::io, "error in opening the file '" + filename + "'");
    struct stat st;
    if (fstat(fd_, &st) != 0) {
        ::close(fd_);
        throw error(error_kind::io, "error in opening the file '" + filename + "'");
    }
    size_ = uint64_t(st.st_size);
    pieces_ = (size_ + piece_ - 1) / piece_;
}

fastq_pieces::~fastq_pieces() {
    if (fd_ >= 0) ::close(fd_);
}

fastq_pieces::parsed fastq_pieces::parse(uint64_t i, uint32_t k, char* bases, uint64_t bases_capacity, uint64_t* offsets,
                                         uint64_t offsets_capacity, std::vector<char>& raw) const {
    const uint64_t cut = i * piece_, next_cut = std::min(size_, cut + piece_);
    parsed out;
    offsets[0] = 0;
    for (uint64_t slack = slack_;; slack *= 4) {
        if (slack > slack_ && next_cut - cut + slack > bases_capacity) {  // a record longer than the output: not for this reader
            out.overflow = true;
            return out;
        }
        /* file bytes [from, to): from one byte before the cut (does the cut start a line?) to the next cut plus slack */
        const uint64_t from = cut ? cut - 1 : 0, to = std::min(size_, next_cut + slack);
        raw.resize(size_t(to - from));
        for (uint64_t done = 0; done < to - from;) {
            const ssize_t got = ::pread(fd_, raw.data() + done, size_t(to - from - done), off_t(from + done));
            if (got < 0) throw error(error_kind::io, "error while reading the query file");
            if (got == 0) throw error(error_kind::io, "error while reading the query file: it shrank while being read");
            done += uint64_t(got);
        }
        char const* const B = raw.data();
        const uint64_t n = to - from;
        const bool whole = to == size_;  // the rest of the file is in hand: a line without '\n' is the last line, not a lack of slack
        bool starved = false;
        /* index of the '\n' ending the line that starts at p; n when the file ends first (or the buffer: then `starved`) */
        auto line_end = [&](uint64_t p) -> uint64_t {
            if (p < n) {
                const size_t room = size_t(n - p) & (~size_t(0) >> 1);  // (n - p > 0; the mask only tells gcc that it is no huge number)
                if (void const* nl = memchr(B + p, '\n', room)) return uint64_t(static_cast<char const*>(nl) - B);
            }
            if (!whole) starved = true;
            return n;
        };
        /* does a record start at line start p? '@', and the line after next begins with '+' */
        auto record_starts = [&](uint64_t p) -> bool {
            if (B[p] != '@') return false;
            const uint64_t l1 = line_end(p) + 1;
            const uint64_t l2 = l1 < n ? line_end(l1) + 1 : n;
            return l2 < n && B[l2] == '+';
        };
        /* synchronise: the first record start at or behind the cut (size_: none) */
        uint64_t p = cut ? std::min(n, line_end(0) + 1) : 0;  // B[0] is the byte before the cut: its line ends, the next begins
        if (cut)
            while (p < n && !starved && !record_starts(p)) p = std::min(n, line_end(p) + 1);
        if (starved) continue;
        out.first_record = from + p;
        /* four lines per record from there: header, bases, '+', qualities, nothing validated (src/query.cpp:78-108) */
        uint64_t reads = 0, nb = 0;
        while (p < n && from + p < next_cut) {
            const uint64_t b0 = line_end(p) + 1;   // behind the header
            if (starved) break;
            if (b0 >= n) {
                if (!whole) {                      // the header's line end is the buffer's last byte: more of the file is needed, not less
                    starved = true;
                    break;
                }
                p = n;                             // a header and nothing behind it: the sequential reader ends here as well
                break;
            }
            const uint64_t e1 = line_end(b0);      // the bases: [b0, e1)
            if (starved) break;
            const uint64_t len = e1 - b0;
            if (len >= k) {
                if (nb + len > bases_capacity || reads + 1 >= offsets_capacity) {
                    out.overflow = true;
                    return out;
                }
                memcpy(bases + nb, B + b0, size_t(len));
                nb += len;
                offsets[++reads] = nb;
            }
            uint64_t q = std::min(n, e1 + 1);
            for (int skipped = 0; skipped < 2 && q < n && !starved; ++skipped) q = std::min(n, line_end(q) + 1);  // '+', qualities
            if (starved) break;
            p = q;
        }
        if (starved) continue;
        out.next_record = from + p;
        out.num_reads = reads;
        out.num_bases = nb;
        return out;
    }
}

bool load_reads(std::string const& filename, bool multiline, uint32_t k, read_batch& out) {
    read_stream in(filename, multiline, k);
    out.bases.clear();
    out.offsets.assign(1, 0);
    if (!in.supported()) return false;
    in.next(out, ~uint64_t(0));
    return true;
}

}  // namespace sshash_amd
