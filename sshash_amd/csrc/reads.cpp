// reads.cpp -- see reads.hpp.
#include "reads.hpp"

#include "errors.hpp"

#include <zlib.h>

#include <cstring>
#include <memory>
#include <stdexcept>

namespace sshash_amd {

namespace {

bool ends_with(std::string const& s, char const* suffix) {
    const size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

struct line_reader {
    gzFile f;
    std::vector<char> buf;
    explicit line_reader(std::string const& filename) : buf(1 << 16) {
        f = gzopen(filename.c_str(), "rb");
        if (!f) throw error(error_kind::io, "error in opening the file '" + filename + "'");
        gzbuffer(f, 1 << 20);
    }
    ~line_reader() { gzclose(f); }
    /* std::getline semantics: false once nothing at all could be read */
    bool next(std::string& line) {
        line.clear();
        bool any = false;
        for (;;) {
            if (!gzgets(f, buf.data(), int(buf.size()))) return any;
            any = true;
            const size_t n = strlen(buf.data());
            if (n && buf[n - 1] == '\n') {
                line.append(buf.data(), n - 1);
                return true;
            }
            line.append(buf.data(), n);
        }
    }
};

void push_read(read_batch& out, std::string const& s, uint32_t k) {
    if (s.size() < k) return;
    out.bases.insert(out.bases.end(), s.begin(), s.end());
    out.offsets.push_back(out.bases.size());
}

}  // namespace

struct read_stream::impl {
    line_reader in;
    enum { FASTQ, FASTA, FASTA_MULTILINE } format;
    uint32_t k;
    bool done = false;
    std::string line, seq, segment;
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
    while (out.bases.size() < max_bases) {
        if (r.format == impl::FASTQ) {
            if (!r.in.next(r.line) || !r.in.next(r.seq)) { r.done = true; break; }  // header, bases
            push_read(out, r.seq, r.k);
            r.in.next(r.line);  // '+'
            r.in.next(r.line);  // qualities
        } else if (r.format == impl::FASTA) {
            if (!r.in.next(r.line) || !r.in.next(r.seq)) { r.done = true; break; }
            push_read(out, r.seq, r.k);
        } else {
            if (!r.in.next(r.line)) {
                push_read(out, r.segment, r.k);
                r.segment.clear();
                r.done = true;
                break;
            }
            if (r.line.empty()) {
                push_read(out, r.segment, r.k);
                r.segment.clear();
            } else {
                r.segment += r.line;
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
