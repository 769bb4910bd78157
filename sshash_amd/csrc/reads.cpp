// reads.cpp -- see reads.hpp.
#include "reads.hpp"

#include "errors.hpp"

#include <zlib.h>

#include <cstring>
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

bool load_reads(std::string const& filename, bool multiline, uint32_t k, read_batch& out) {
    out.bases.clear();
    out.offsets.assign(1, 0);
    const bool fasta = ends_with(filename, ".fa") || ends_with(filename, ".fasta") || ends_with(filename, ".fa.gz") ||
                       ends_with(filename, ".fasta.gz");
    const bool fastq = ends_with(filename, ".fq") || ends_with(filename, ".fastq") || ends_with(filename, ".fq.gz") ||
                       ends_with(filename, ".fastq.gz");
    if (!fasta && !fastq) {
        /* the reference opens the file before looking at the extension (src/query.cpp:127-128) */
        line_reader probe(filename);
        return false;
    }
    line_reader in(filename);
    std::string line, seq;
    if (fastq) {
        for (;;) {
            if (!in.next(line)) break;   // header
            if (!in.next(seq)) break;    // bases
            push_read(out, seq, k);
            in.next(line);               // '+'
            in.next(line);               // qualities
        }
    } else if (!multiline) {
        for (;;) {
            if (!in.next(line)) break;  // header
            if (!in.next(seq)) break;
            push_read(out, seq, k);
        }
    } else {
        std::string segment;
        while (in.next(line)) {
            if (line.empty()) {
                push_read(out, segment, k);
                segment.clear();
            } else {
                segment += line;
            }
        }
        push_read(out, segment, k);
    }
    return true;
}

}  // namespace sshash_amd
