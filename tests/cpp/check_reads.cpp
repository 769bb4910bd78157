// check_reads.cpp -- host-side check of the query-file readers (csrc/reads.cpp): read_stream hands a file over in bounded
// batches of whole reads; whatever the batch size, the reads that come out are those of load_reads (the whole file at
// once), in order. Formats as src/query.cpp: FASTQ (.fq), one-line FASTA (.fa), multiline FASTA. Plain g++ + zlib, no GPU.
// usage: check_reads <file> <multiline 0|1> <k>   -> prints "OK <reads> <bases> <checksum of the reads>"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../sshash_amd/csrc/reads.hpp"

using namespace sshash_amd;

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const std::string file = argv[1];
    const bool multiline = std::atoi(argv[2]) != 0;
    const uint32_t k = uint32_t(std::atoi(argv[3]));
    read_batch whole;
    if (!load_reads(file, multiline, k, whole)) return printf("unsupported\n"), 0;
    for (uint64_t batch : {uint64_t(1), uint64_t(7), uint64_t(100), uint64_t(4096), ~uint64_t(0)}) {
        read_stream in(file, multiline, k);
        if (!in.supported()) return printf("stream says unsupported\n"), 1;
        read_batch part;
        uint64_t read = 0, base = 0, batches = 0;
        while (in.next(part, batch)) {
            ++batches;
            if (part.num_reads() == 0) return printf("empty batch (batch size %llu)\n", (unsigned long long)batch), 1;
            for (uint64_t r = 0; r < part.num_reads(); ++r, ++read) {
                if (read >= whole.num_reads()) return printf("too many reads\n"), 1;
                const uint64_t len = part.offsets[r + 1] - part.offsets[r];
                if (len != whole.offsets[read + 1] - whole.offsets[read]) return printf("read %llu: length differs\n", (unsigned long long)read), 1;
                for (uint64_t j = 0; j < len; ++j)
                    if (part.bases[part.offsets[r] + j] != whole.bases[whole.offsets[read] + j]) return printf("read %llu: bases differ\n", (unsigned long long)read), 1;
                base += len;
            }
            /* a batch ends with the first read that takes it to the limit: without its last read it is below it */
            if (batch != ~uint64_t(0) && part.num_reads() > 1 && part.offsets[part.num_reads() - 1] >= batch)
                return printf("batch of %llu bases holds a read too many\n", (unsigned long long)batch), 1;
        }
        if (read != whole.num_reads() || base != whole.bases.size()) return printf("reads missing (batch size %llu)\n", (unsigned long long)batch), 1;
        if (in.next(part, batch)) return printf("a batch after the end\n"), 1;
        (void)batches;
    }
    /* an uncompressed FASTQ also goes through the piecewise reader (reads.hpp: fastq_pieces), whatever the piece size: every
       piece starts where its predecessor stopped and the pieces together hold the reads of load_reads, in order -- or the chain
       check says "irregular" (then the library falls back to the sequential reader), never different reads */
    if (fastq_pieces::applicable(file) && !multiline) {
        for (uint64_t piece : {uint64_t(4096), uint64_t(10007), uint64_t(1) << 16, uint64_t(1) << 20, uint64_t(1) << 26}) {
            fastq_pieces in(file, piece);
            std::vector<char> bases(in.bases_capacity()), raw;
            std::vector<uint64_t> offsets(in.offsets_capacity(k));
            uint64_t read = 0, expect = 0;
            bool regular = true;
            for (uint64_t i = 0; i < in.num_pieces() && regular; ++i) {
                const fastq_pieces::parsed got = in.parse(i, k, bases.data(), bases.size(), offsets.data(), offsets.size(), raw);
                if (got.overflow || got.first_record != expect) { regular = false; break; }
                expect = got.next_record;
                for (uint64_t r = 0; r < got.num_reads; ++r, ++read) {
                    if (read >= whole.num_reads()) return printf("pieces of %llu bytes: too many reads\n", (unsigned long long)piece), 1;
                    const uint64_t len = offsets[r + 1] - offsets[r];
                    if (len != whole.offsets[read + 1] - whole.offsets[read] ||
                        memcmp(bases.data() + offsets[r], whole.bases.data() + whole.offsets[read], len) != 0)
                        return printf("pieces of %llu bytes: read %llu differs\n", (unsigned long long)piece, (unsigned long long)read), 1;
                }
            }
            if (regular && (expect != in.file_bytes() || read != whole.num_reads()))
                return printf("pieces of %llu bytes: chain complete but %llu of %llu reads, stopped at %llu of %llu\n", (unsigned long long)piece,
                              (unsigned long long)read, (unsigned long long)whole.num_reads(), (unsigned long long)expect, (unsigned long long)in.file_bytes()), 1;
            fprintf(stderr, "pieces %llu %s\n", (unsigned long long)piece, regular ? "regular" : "irregular");
        }
    }
    uint64_t h = 1469598103934665603ull;  // FNV-1a over the bases and the read boundaries: equal files give equal lines
    for (char c : whole.bases) h = (h ^ uint64_t(uint8_t(c))) * 1099511628211ull;
    for (uint64_t o : whole.offsets) h = (h ^ o) * 1099511628211ull;
    printf("OK %llu %llu %016llx\n", (unsigned long long)whole.num_reads(), (unsigned long long)whole.bases.size(), (unsigned long long)h);
    return 0;
}
