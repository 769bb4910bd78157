// reads.cpp -- see reads.hpp.
#include "reads.hpp"

#include "errors.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <stdexcept>

namespace sshash_amd {

namespace {

bool ends_with(std::string const& s, char const* suffix) {
    const size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

/* Lines of a (possibly gzip-compressed) file: the file is inflated a block of megabytes at a time (gzread) and split with
   memchr; a line is handed out as a view into the block -- no per-line gzgets, no per-line std::string. (Round 2 assembled
   every line with gzgets into a std::string: 0.2 GB/s on uncompressed FASTQ, below zlib's own inflate rate; this reader
   splits at several GB/s, so an uncompressed file is read at the page cache's pace and a .gz at zlib's.) */
struct line_reader {
    gzFile f;
    std::vector<char> buf;
    size_t begin = 0, end = 0;
    bool eof = false;
    explicit line_reader(std::string const& filename) : buf(size_t(8) << 20) {
        f = gzopen(filename.c_str(), "rb");
        if (!f) throw error(error_kind::io, "error in opening the file '" + filename + "'");
        gzbuffer(f, 1 << 20);
    }
    ~line_reader() { gzclose(f); }
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

bool load_reads(std::string const& filename, bool multiline, uint32_t k, read_batch& out) {
    read_stream in(filename, multiline, k);
    out.bases.clear();
    out.offsets.assign(1, 0);
    if (!in.supported()) return false;
    in.next(out, ~uint64_t(0));
    return true;
}

}  // namespace sshash_amd
