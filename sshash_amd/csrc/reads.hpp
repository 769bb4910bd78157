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

/* An UNCOMPRESSED FASTQ file read by several threads at once (the file query runs at the reader's pace: one thread splits
   9 GB/s of file, the streaming kernels take 40+ GB/s of bases, HISTORY.md). The file is cut at fixed byte positions;
   piece i takes the records whose header line STARTS in [cut i, cut i+1): it finds the first record start at or behind its cut
   -- a line beginning with '@' whose line after next begins with '+': of the four lines of a record only the header passes
   that test (a quality line may begin with '@', but then the line after next is a line of bases) -- and from there counts
   lines four at a time exactly like the sequential reader (src/query.cpp:78-108: header, bases, '+', qualities; nothing is
   validated). `parse` reports where it started and where it stopped; the caller checks that every piece started where its
   predecessor stopped -- then, by induction from offset 0, the pieces together hold exactly the records the sequential
   reader produces -- and falls back to the sequential reader if not (a file that is not four lines per record). Thread-safe
   (pread). */
class fastq_pieces {
public:
    /* .fq / .fastq by name and not gzip by content (a .fastq that is really a gzip stream goes through zlib as before) */
    static bool applicable(std::string const& filename);
    fastq_pieces(std::string const& filename, uint64_t piece_bytes);  // throws when the file cannot be opened
    ~fastq_pieces();
    fastq_pieces(fastq_pieces const&) = delete;
    fastq_pieces& operator=(fastq_pieces const&) = delete;
    uint64_t num_pieces() const { return pieces_; }
    uint64_t file_bytes() const { return size_; }
    uint64_t piece_bytes() const { return piece_; }
    struct parsed {
        uint64_t first_record = 0;  // file offset of the first record of the piece (where it synchronised)
        uint64_t next_record = 0;   // file offset behind its last record
        uint64_t num_reads = 0, num_bases = 0;
        bool overflow = false;      // the output did not fit (nothing usable was produced)
    };
    /* Piece i: the bases of its reads (those of at least k bases) back to back into `bases`, offsets[0 .. num_reads] (offsets[0]
       = 0). `raw` is the calling thread's scratch for the file bytes (kept between calls). */
    parsed parse(uint64_t i, uint32_t k, char* bases, uint64_t bases_capacity, uint64_t* offsets, uint64_t offsets_capacity,
                 std::vector<char>& raw) const;
    /* enough for any piece: bases <= the bytes looked at; a read of >= k bases takes >= k + 6 bytes of file ('@', four line ends, a
       '+', its bases -- nothing is validated, so the quality line may be empty: ADVICE r4, a file of such records used to overflow the
       offsets sized for 2 k + 6 and fall back to the sequential reader after partial work) */
    uint64_t bases_capacity() const { return piece_ + slack_; }
    uint64_t offsets_capacity(uint32_t k) const { return (piece_ + slack_) / (uint64_t(k) + 6) + 2; }

private:
    int fd_ = -1;
    uint64_t size_ = 0, piece_ = 0, pieces_ = 0;
    uint64_t slack_ = uint64_t(1) << 20;  // bytes read beyond the cut for the records that straddle it (grown on demand)
};

/* CPUs this process may use: affinity mask and cgroup quota (std::thread::hardware_concurrency() reports the machine's) */
unsigned usable_cpus();

/* The whole file at once. Returns false when the extension is not a supported format; throws std::runtime_error
   ("error in opening the file ...") when the file cannot be opened. */
bool load_reads(std::string const& filename, bool multiline, uint32_t k, read_batch& out);

}  // namespace sshash_amd
