// ref_lookup.cpp -- TEST INFRASTRUCTURE (never linked into, imported by or executed from the product).
//
// A driver over the REFERENCE'S OWN dictionary, compiled by `make -C oracle ref-full` against the reference sources where they lie
// under /root/reference -- possible only when the third-party sources the reference's lookup path includes are present
// (external/pthash with its nested bits / essentials / fastmod / xxHash / mm_file: an EMPTY submodule directory in the checkout this
// repo was built against, which is why SURVEY.md 8(f1) is blocked and why this file has never been compiled here). It holds no copy of
// reference code: it includes the reference's headers and translation units exactly as /root/reference/tools/sshash.cpp:1-16 does.
//
//   ref_lookup <index.sshash> lookup  <kmers.txt>        one ASCII k-mer per line -> one JSON line per k-mer: all eight lookup_result
//                                                        fields (include/util.hpp:38-62) of dictionary::lookup(char const*, true)
//   ref_lookup <index.sshash> query   <reads.fastq[.gz]> -> one JSON line: streaming_query_report (include/util.hpp:21-36) of
//                                                        dictionary::streaming_query_from_file (include/dictionary.hpp:81-82)
//   ref_lookup <index.sshash> info                       -> k, m, canonical, num_kmers, num_strings
//
// What it pins once it runs (tests/golden/make_reference_index.py stores its output; tests/test_reference_index.py compares):
// minimizer_found of a miss, the searches / extensions split, every id -- from the reference itself instead of the restatement.
#include <fstream>
#include <iostream>
#include <string>

#include "tools/common.hpp"            // open_dictionary: essentials::load (tools/common.hpp:19-29)
#include "src/builder/build.cpp"       // (the reference's tools are one translation unit: tools/sshash.cpp:9-12)
#include "src/dictionary.cpp"
#include "src/query.cpp"
#include "src/info.cpp"

using namespace sshash;

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cerr << "usage: ref_lookup <index.sshash> lookup <kmers.txt> | query <reads.fastq[.gz]> | info" << std::endl;
        return 2;
    }
    dictionary_type dict;
    open_dictionary(dict, argv[1], /* mmap */ false, /* verbose */ false);
    const std::string what = argv[2];
    if (what == "info") {
        std::cout << "{\"k\": " << dict.k() << ", \"m\": " << dict.m() << ", \"canonical\": " << (dict.canonical() ? "true" : "false")
                  << ", \"num_kmers\": " << dict.num_kmers() << ", \"num_strings\": " << dict.num_strings() << "}" << std::endl;
        return 0;
    }
    if (argc < 4) return 2;
    if (what == "lookup") {
        std::ifstream in(argv[3]);
        std::string line;
        while (std::getline(in, line)) {
            if (line.size() != dict.k()) continue;
            const lookup_result r = dict.lookup(line.c_str(), true);
            std::cout << "{\"kmer\": \"" << line << "\", \"kmer_id\": " << r.kmer_id << ", \"kmer_id_in_string\": " << r.kmer_id_in_string
                      << ", \"kmer_offset\": " << r.kmer_offset << ", \"kmer_orientation\": " << r.kmer_orientation
                      << ", \"string_id\": " << r.string_id << ", \"string_begin\": " << r.string_begin << ", \"string_end\": " << r.string_end
                      << ", \"minimizer_found\": " << (r.minimizer_found ? "true" : "false") << "}\n";
        }
        return 0;
    }
    if (what == "query") {
        const streaming_query_report r = dict.streaming_query_from_file(argv[3], /* multiline */ false);
        std::cout << "{\"num_kmers\": " << r.num_kmers << ", \"num_positive_kmers\": " << r.num_positive_kmers
                  << ", \"num_negative_kmers\": " << r.num_negative_kmers << ", \"num_invalid_kmers\": " << r.num_invalid_kmers
                  << ", \"num_searches\": " << r.num_searches << ", \"num_extensions\": " << r.num_extensions << "}" << std::endl;
        return 0;
    }
    return 2;
}
