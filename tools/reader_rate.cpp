// reader_rate.cpp -- how fast csrc/reads.cpp turns a query file into batches of reads (host only: no GPU involved).
//   g++ -O2 -std=c++17 tools/reader_rate.cpp sshash_amd/csrc/reads.cpp -lz -o /tmp/reader_rate && /tmp/reader_rate file.fastq[.gz] [k]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sys/stat.h>

#include "../sshash_amd/csrc/reads.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const uint32_t k = argc > 2 ? uint32_t(atoi(argv[2])) : 31;
    struct stat st;
    if (stat(argv[1], &st) != 0) return 2;
    const auto t0 = std::chrono::steady_clock::now();
    sshash_amd::read_stream in(argv[1], false, k);
    sshash_amd::read_batch b;
    uint64_t reads = 0, bases = 0;
    while (in.next(b, uint64_t(256) << 20)) {
        reads += b.num_reads();
        bases += b.bases.size();
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"file_bytes\": %llu, \"reads\": %llu, \"bases\": %llu, \"seconds\": %.3f, \"file_GBps\": %.3f, \"Gbases_per_s\": %.3f}\n",
           (unsigned long long)st.st_size, (unsigned long long)reads, (unsigned long long)bases, s, st.st_size / s / 1e9, bases / s / 1e9);
    return 0;
}
