// reads.hpp -- query-file readers feeding the batched streaming query.
//
// Same record rules as the reference readers (src/query.cpp):
//   FASTQ           4 lines per read, line 2 carries the bases                 :78-108
//   FASTA           header line + ONE sequence line per record                 :49-76
//   FASTA multiline every line (headers included -- the reference does not special-case '>',
//                   such characters simply make k-mers invalid) is concatenated until an empty
//                   line ends the segment                                      :9-47, include/util.hpp:287-340
// Format is chosen by file extension (.fa/.fasta/.fq/.fastq, optionally .gz)  :131-171. A .gz is read through zlib (one
// thread: a deflate stream does not split); a BGZF file (bgzip: gzip members with their size in the header -- any gzip reader,
// the reference's included, reads it as a .gz) is recognised by its first header and inflated member by member on all cores.
// Reads shorter than k are dropped here (they contribute no k-mer: :63,:92).
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace sshash_amd {

struct read_batch {
    std::vector<char> bases;        // reads back to back
    std::vector<uint64_t> offsets;  // num_reads + 1
    uint64_t num_reads() const { return offsets.empty() ? 0 : offsets.size() - 1; }
};

/* The same records, a bounded number of bases at a time (whole reads; a batch ends with the first read that takes it to
   `max_bases` or beyond): a query file of hundreds of gigabytes never sits in host memory as a whole. */
class read_stream {
public:
    read_stream(std::string const& filename, bool multiline, uint32_t k);  // throws when the file cannot be opened
    ~read_stream();
    bool supported() const { return bool(m); }  // false: the extension is not a supported format
    /* false once the file is exhausted and `out` holds nothing */
    bool next(read_batch& out, uint64_t max_bases);

private:
    struct impl;
    std::unique_ptr<impl> m;
};

/* The whole file at once. Returns false when the extension is not a supported format; throws std::runtime_error
   ("error in opening the file ...") when the file cannot be opened. */
bool load_reads(std::string const& filename, bool multiline, uint32_t k, read_batch& out);

}  // namespace sshash_amd
