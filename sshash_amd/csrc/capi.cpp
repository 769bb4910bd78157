// capi.cpp -- the C ABI declared in include/sshash_amd.h (thin shim over engine / index).
#include <condition_variable>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/sshash_amd.h"
#include "engine.hpp"
#include "hooks.hpp"
#include "reads.hpp"

using namespace sshash_amd;

struct sshash_dict {
    std::shared_ptr<host_index> idx;
    std::unique_ptr<engine> eng;
};

namespace {

thread_local std::string g_last_error;

sshash_status fail(sshash_status s, std::string const& msg) {
    g_last_error = msg;
    return s;
}

sshash_status classify(std::exception const& e) {
    g_last_error = e.what();
    if (auto const* typed = dynamic_cast<error const*>(&e)) return sshash_status(int(typed->kind));
    return SSHASH_ERR_INTERNAL;
}

template <typename Fn>
sshash_status guarded(Fn&& fn) {
    try {
        test_hooks_refresh();  // (hooks.hpp: the environment is read here, on the caller's thread, never by the library's own threads)
        fn();
        return SSHASH_OK;
    } catch (std::exception const& e) { return classify(e); } catch (...) {
        return fail(SSHASH_ERR_INTERNAL, "unknown exception");
    }
}

build_options to_options(sshash_build_config const* cfg) {
    build_options o;
    if (cfg) {
        if (cfg->k) o.k = cfg->k;
        if (cfg->m) o.m = cfg->m;
        o.seed = cfg->seed;
        o.canonical = cfg->canonical != 0;
        o.num_threads = cfg->num_threads ? cfg->num_threads : std::max(1u, std::thread::hardware_concurrency());
        if (cfg->lambda > 0) o.lambda = cfg->lambda;
        o.verbose = cfg->verbose != 0;
        o.weighted = cfg->weighted != 0;
        o.num_shards = cfg->num_shards ? cfg->num_shards : 1;
        o.shard_id = cfg->shard_id;
    }
    return o;
}

sshash_dict* wrap(std::shared_ptr<host_index> idx) {
    auto* d = new sshash_dict;
    d->idx = std::move(idx);
    d->eng = std::make_unique<engine>(d->idx);
    return d;
}

result_view to_view(sshash_results const* r) {
    result_view v{};
    if (r) {
        v.kmer_id = r->kmer_id;
        v.kmer_id_in_string = r->kmer_id_in_string;
        v.kmer_offset = r->kmer_offset;
        v.string_id = r->string_id;
        v.string_begin = r->string_begin;
        v.string_end = r->string_end;
        v.kmer_orientation = r->kmer_orientation;
        v.minimizer_found = r->minimizer_found;
    }
    return v;
}

bool wants_full(sshash_results const* r) {
    return r->kmer_id_in_string || r->kmer_offset || r->string_id || r->string_begin || r->string_end ||
           r->kmer_orientation || r->minimizer_found;
}

}  // namespace

extern "C" {

const char* sshash_last_error(void) { return g_last_error.c_str(); }

const char* sshash_build_info(void) {
    static const std::string info = std::string("isa_guard=") + sshash_amd::isa_guard_state() + ";arch=gfx950";
    return info.c_str();
}

void sshash_build_config_default(sshash_build_config* cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->k = 31;
    cfg->m = 20;
    cfg->seed = 1;
    cfg->num_threads = 1;
    cfg->lambda = 5.0;
    cfg->num_shards = 1;
}

sshash_status sshash_build_from_fasta(const char* filename, const sshash_build_config* cfg, sshash_dict** out) {
    if (!filename || !out) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    *out = nullptr;
    return guarded([&] {
        auto idx = std::make_shared<host_index>();
        build_from_fasta(*idx, filename, to_options(cfg));
        *out = wrap(std::move(idx));
    });
}

sshash_status sshash_build_from_packed(const uint64_t* words, const uint64_t* endpoints, uint64_t num_strings,
                                       const sshash_build_config* cfg, sshash_dict** out) {
    if (!words || !endpoints || !out || num_strings == 0) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    *out = nullptr;
    return guarded([&] {
        std::vector<uint64_t> e(endpoints, endpoints + num_strings + 1);
        std::vector<uint64_t> w(words, words + (2 * e.back() + 63) / 64);
        auto idx = std::make_shared<host_index>();
        build_from_packed(*idx, std::move(w), std::move(e), to_options(cfg));
        *out = wrap(std::move(idx));
    });
}

sshash_status sshash_save(const sshash_dict* d, const char* filename) {
    if (!d || !filename) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { save_index(*d->idx, filename); });
}

sshash_status sshash_load(const char* filename, sshash_dict** out) {
    if (!filename || !out) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    *out = nullptr;
    return guarded([&] {
        auto idx = std::make_shared<host_index>();
        load_index(*idx, filename);
        *out = wrap(std::move(idx));
    });
}

void sshash_free(sshash_dict* d) { delete d; }

sshash_status sshash_get_info(const sshash_dict* d, sshash_info* info) {
    if (!d || !info) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    host_index const& x = *d->idx;
    std::memset(info, 0, sizeof(*info));
    std::memcpy(info->version, x.version, 3);
    info->canonical = x.canonical;
    info->k = x.k;
    info->m = x.m;
    info->words_per_kmer = x.words_per_kmer();
    info->num_kmers = x.num_kmers;
    info->num_strings = x.num_strings;
    info->num_bases = x.num_bases;
    info->num_minimizers = x.num_minimizers();
    info->num_bits = x.num_bits();
    info->skew_partitions = x.skew_num_partitions;
    info->weighted = x.weighted() ? 1 : 0;
    info->num_shards = x.num_shards;
    info->shard_id = x.shard_id;
    return SSHASH_OK;
}

sshash_status sshash_bucket_stats(const sshash_dict* d, uint64_t out[64]) {
    if (!d || !out) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { bucket_statistics(*d->idx, out); });
}

int sshash_device_count(void) { return visible_device_count(); }

sshash_status sshash_to_device(sshash_dict* d, int device) {
    if (!d) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->to_device(device); });
}

sshash_status sshash_device_bytes(const sshash_dict* d, int device, uint64_t* bytes) {
    if (!d || !bytes) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { *bytes = d->eng->device_bytes(device); });
}

sshash_status sshash_device_stats(const sshash_dict* d, int device, uint64_t out[16]) {
    if (!d || !out) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->device_stats(device, out); });
}

sshash_status sshash_device_table_histogram(const sshash_dict* d, int device, uint64_t out[32]) {
    if (!d || !out) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->device_table_histogram(device, out); });
}

sshash_status sshash_lookup_packed_device(const sshash_dict* d, int device, const uint64_t* kmers, uint64_t n,
                                          int check_rc, const sshash_results* out, void* hip_stream) {
    if (!d || !out || (!kmers && n)) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        d->eng->lookup_packed_device(device, kmers, n, check_rc != 0, wants_full(out) ? out_mode::full : out_mode::ids,
                                     to_view(out), nullptr, hip_stream);
    });
}

sshash_status sshash_lookup_ascii_device(const sshash_dict* d, int device, const char* kmers, uint64_t n, int check_rc,
                                         const sshash_results* out, void* hip_stream) {
    if (!d || !out || (!kmers && n)) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        d->eng->lookup_ascii_device(device, kmers, n, check_rc != 0, wants_full(out) ? out_mode::full : out_mode::ids,
                                    to_view(out), nullptr, hip_stream);
    });
}

sshash_status sshash_lookup_packed(const sshash_dict* d, const uint64_t* kmers, uint64_t n, int check_rc,
                                   const sshash_results* out) {
    if (!d || !out || (!kmers && n)) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        d->eng->lookup_packed_host(kmers, n, check_rc != 0, wants_full(out) ? out_mode::full : out_mode::ids, to_view(out),
                                   nullptr);
    });
}

sshash_status sshash_lookup_ascii(const sshash_dict* d, const char* kmers, uint64_t n, int check_rc,
                                  const sshash_results* out) {
    if (!d || !out || (!kmers && n)) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        d->eng->lookup_ascii_host(kmers, n, check_rc != 0, wants_full(out) ? out_mode::full : out_mode::ids, to_view(out),
                                  nullptr);
    });
}

sshash_status sshash_neighbours_packed_device(const sshash_dict* d, int device, const uint64_t* kmers, uint64_t n,
                                              int check_rc, const sshash_results* out, void* hip_stream) {
    if (!d || !out || (!kmers && n)) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        d->eng->neighbours_packed_device(device, kmers, n, check_rc != 0, wants_full(out) ? out_mode::full : out_mode::ids,
                                         to_view(out), hip_stream);
    });
}

sshash_status sshash_neighbours_packed(const sshash_dict* d, const uint64_t* kmers, uint64_t n, int check_rc,
                                       const sshash_results* out) {
    if (!d || !out || (!kmers && n)) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        d->eng->neighbours_packed_host(kmers, n, check_rc != 0, wants_full(out) ? out_mode::full : out_mode::ids, to_view(out));
    });
}

sshash_status sshash_string_size(const sshash_dict* d, const uint64_t* string_ids, uint64_t n, uint64_t* out_sizes) {
    if (!d || (n && (!string_ids || !out_sizes))) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        host_index const& x = *d->idx;
        for (uint64_t i = 0; i < n; ++i) {
            if (string_ids[i] >= x.num_strings) throw error(error_kind::argument, "string_id out of range");
            out_sizes[i] = x.endpoints[string_ids[i] + 1] - x.endpoints[string_ids[i]] - x.k + 1;
        }
    });
}

sshash_status sshash_string_offsets(const sshash_dict* d, const uint64_t* string_ids, uint64_t n, uint64_t* out_begin, uint64_t* out_end) {
    if (!d || (n && (!string_ids || !out_begin || !out_end))) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        host_index const& x = *d->idx;
        for (uint64_t i = 0; i < n; ++i) {
            if (string_ids[i] >= x.num_strings) throw error(error_kind::argument, "string_id out of range");
            out_begin[i] = x.endpoints[string_ids[i]];
            out_end[i] = x.endpoints[string_ids[i] + 1];
        }
    });
}

sshash_status sshash_string_neighbours(const sshash_dict* d, const uint64_t* string_ids, uint64_t n, int check_rc,
                                       const sshash_results* out) {
    if (!d || !out || (!string_ids && n)) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        d->eng->string_neighbours_host(string_ids, n, check_rc != 0, wants_full(out) ? out_mode::full : out_mode::ids, to_view(out));
    });
}

sshash_status sshash_is_member_packed_device(const sshash_dict* d, int device, const uint64_t* kmers, uint64_t n,
                                             int check_rc, uint8_t* out, void* hip_stream) {
    if (!d || !out || (!kmers && n)) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        d->eng->lookup_packed_device(device, kmers, n, check_rc != 0, out_mode::member, result_view{}, out, hip_stream);
    });
}

sshash_status sshash_is_member_packed(const sshash_dict* d, const uint64_t* kmers, uint64_t n, int check_rc, uint8_t* out) {
    if (!d || !out || (!kmers && n)) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->lookup_packed_host(kmers, n, check_rc != 0, out_mode::member, result_view{}, out); });
}

sshash_status sshash_is_member_ascii(const sshash_dict* d, const char* kmers, uint64_t n, int check_rc, uint8_t* out) {
    if (!d || !out || (!kmers && n)) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->lookup_ascii_host(kmers, n, check_rc != 0, out_mode::member, result_view{}, out); });
}

sshash_status sshash_weight(const sshash_dict* d, const uint64_t* kmer_ids, uint64_t n, uint64_t* out_weights) {
    if (!d || (n && (!kmer_ids || !out_weights))) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        for (uint64_t i = 0; i < n; ++i) out_weights[i] = weight_of(*d->idx, kmer_ids[i]);
    });
}

sshash_status sshash_weight_device(const sshash_dict* d, int device, const uint64_t* kmer_ids, uint64_t n,
                                   uint64_t* out_weights, void* hip_stream) {
    if (!d || (n && (!kmer_ids || !out_weights))) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->weight_device(device, kmer_ids, n, out_weights, hip_stream); });
}

sshash_status sshash_access(const sshash_dict* d, uint64_t kmer_id, char* out_k_chars) {
    if (!d || !out_k_chars) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { access_kmer(*d->idx, kmer_id, out_k_chars); });
}

sshash_status sshash_access_packed(const sshash_dict* d, const uint64_t* kmer_ids, uint64_t n, uint64_t* out_words) {
    if (!d || (!kmer_ids && n) || (!out_words && n)) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        const uint32_t W = d->idx->words_per_kmer();
        const uint32_t nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        /* an id >= num_kmers throws inside a worker: parallel_ranges hands the first exception back to this thread */
        detail::parallel_ranges(n, n >= 4096 ? nt : 1, [&](uint64_t b, uint64_t e, uint32_t) {
            for (uint64_t i = b; i < e; ++i) access_kmer_packed(*d->idx, kmer_ids[i], out_words + i * W);
        });
    });
}

sshash_status sshash_access_packed_device(const sshash_dict* d, int device, const uint64_t* kmer_ids, uint64_t n,
                                          uint64_t* out_words, void* hip_stream) {
    if (!d || (!kmer_ids && n) || (!out_words && n)) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->access_packed_device(device, kmer_ids, n, out_words, hip_stream); });
}

sshash_status sshash_streaming_query_from_file(const sshash_dict* d, const char* filename, int multiline,
                                               sshash_streaming_report* report) {
    if (!d || !filename || !report) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    std::memset(report, 0, sizeof(*report));
    return guarded([&] {
        read_stream in(filename, multiline != 0, d->idx->k);
        if (!in.supported()) {
            /* src/query.cpp:169-171: unsupported extension -> message on stderr, empty report */
            fprintf(stderr, "unsupported query file format\n");
            return;
        }
        /* an uncompressed FASTQ is read and parsed by all the lanes at once (reads.hpp: fastq_pieces); a file that is not four
           lines per record comes back here and takes the sequential reader */
        if (!multiline && fastq_pieces::applicable(filename) && !test_hook_u64("sequential_reader", 0, 0, 1)) {
            streaming_report r;
            if (d->eng->streaming_query_fastq_pieces(filename, r)) {
                report->num_kmers = r.num_kmers;
                report->num_positive_kmers = r.num_positive_kmers;
                report->num_negative_kmers = r.num_negative_kmers;
                report->num_invalid_kmers = r.num_invalid_kmers;
                report->num_searches = r.num_searches;
                report->num_extensions = r.num_extensions;
                return;
            }
        }
        /* The file goes through in batches of ~256 MiB of bases (a FASTQ of hundreds of gigabytes must not be
           materialised): a reader thread decompresses batch i+1 while the devices work on batch i. */
        uint64_t batch_bases = uint64_t(256) << 20;
        batch_bases = test_hook_u64("query_batch_bases", batch_bases, 1, ~uint64_t(0));  // (tests: batch seams inside a small file)
        read_batch slot[2];
        std::mutex mu;
        std::condition_variable cv;
        int filled[2] = {0, 0};  // 0 = free, 1 = holds a batch, 2 = end of file
        std::exception_ptr reader_error;
        bool abandon = false;
        std::thread reader([&] {
            try {
                for (int at = 0;; at ^= 1) {
                    {
                        std::unique_lock<std::mutex> lock(mu);
                        cv.wait(lock, [&] { return filled[at] == 0 || abandon; });
                        if (abandon) return;
                    }
                    const bool more = in.next(slot[at], batch_bases);
                    std::lock_guard<std::mutex> lock(mu);
                    filled[at] = more ? 1 : 2;
                    cv.notify_all();
                    if (!more) return;
                }
            } catch (...) {
                std::lock_guard<std::mutex> lock(mu);
                reader_error = std::current_exception();
                filled[0] = filled[1] = 2;
                cv.notify_all();
            }
        });
        struct joiner {
            std::thread& t;
            std::mutex& mu;
            std::condition_variable& cv;
            bool& abandon;
            ~joiner() {
                {
                    std::lock_guard<std::mutex> lock(mu);
                    abandon = true;
                }
                cv.notify_all();
                t.join();
            }
        } join_reader{reader, mu, cv, abandon};
        double waited = 0, worked = 0;  // SSHASH_AMD_VERBOSE: where the wall clock of the file query went
        uint64_t batches = 0;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        for (int at = 0;; at ^= 1) {
            const double t0 = now();
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return filled[at] != 0; });
                if (filled[at] == 2) break;
            }
            const double t1 = now();
            const streaming_report r = d->eng->streaming_query_host(slot[at].bases.data(), slot[at].offsets.data(), slot[at].num_reads());
            waited += t1 - t0;
            worked += now() - t1;
            ++batches;
            report->num_kmers += r.num_kmers;
            report->num_positive_kmers += r.num_positive_kmers;
            report->num_negative_kmers += r.num_negative_kmers;
            report->num_invalid_kmers += r.num_invalid_kmers;
            report->num_searches += r.num_searches;
            report->num_extensions += r.num_extensions;
            std::lock_guard<std::mutex> lock(mu);
            filled[at] = 0;
            cv.notify_all();
        }
        if (std::getenv("SSHASH_AMD_VERBOSE"))
            fprintf(stderr, "[sshash_amd] file query: %llu batches; waiting for the reader %.3f s, devices at work %.3f s\n",
                    (unsigned long long)batches, waited, worked);
        if (reader_error) std::rethrow_exception(reader_error);
    });
}

sshash_status sshash_streaming_query(const sshash_dict* d, const char* bases, const uint64_t* read_offsets,
                                     uint64_t num_reads, sshash_streaming_report* report) {
    if (!d || !report || (num_reads && (!bases || !read_offsets))) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    std::memset(report, 0, sizeof(*report));
    return guarded([&] {
        const streaming_report r = d->eng->streaming_query_host(bases, read_offsets, num_reads);
        report->num_kmers = r.num_kmers;
        report->num_positive_kmers = r.num_positive_kmers;
        report->num_negative_kmers = r.num_negative_kmers;
        report->num_invalid_kmers = r.num_invalid_kmers;
        report->num_searches = r.num_searches;
        report->num_extensions = r.num_extensions;
    });
}

sshash_status sshash_streaming_query_device(const sshash_dict* d, int device, const char* bases,
                                            const uint64_t* read_offsets, uint64_t num_reads, uint64_t total_bases,
                                            uint64_t* report, void* hip_stream) {
    if (!d || !report || (num_reads && (!bases || !read_offsets))) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->streaming_query_device(device, bases, read_offsets, num_reads, total_bases, report, hip_stream); });
}

static void fill_report(sshash_streaming_report* report, streaming_report const& r) {
    report->num_kmers = r.num_kmers;
    report->num_positive_kmers = r.num_positive_kmers;
    report->num_negative_kmers = r.num_negative_kmers;
    report->num_invalid_kmers = r.num_invalid_kmers;
    report->num_searches = r.num_searches;
    report->num_extensions = r.num_extensions;
}

sshash_status sshash_streaming_lookup_device(const sshash_dict* d, int device, const char* bases, const uint64_t* read_offsets,
                                             uint64_t num_reads, uint64_t total_bases, const sshash_results* out, uint64_t* report,
                                             void* hip_stream) {
    if (!d || !out || (num_reads && (!bases || !read_offsets))) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->streaming_lookup_device(device, bases, read_offsets, num_reads, total_bases, to_view(out), report, hip_stream); });
}

sshash_status sshash_streaming_lookup(const sshash_dict* d, const char* bases, const uint64_t* read_offsets, uint64_t num_reads,
                                      const sshash_results* out, sshash_streaming_report* report) {
    if (!d || !out || (num_reads && (!bases || !read_offsets))) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    if (report) std::memset(report, 0, sizeof(*report));
    return guarded([&] {
        const streaming_report r = d->eng->streaming_lookup_host(bases, read_offsets, num_reads, to_view(out));
        if (report) fill_report(report, r);
    });
}

sshash_status sshash_route_packed_device(const sshash_dict* d, int device, const uint64_t* kmers, uint64_t n,
                                         uint32_t num_shards, uint32_t* owner_forward, uint32_t* owner_reverse,
                                         void* hip_stream) {
    if (!d || (n && (!kmers || !owner_forward || !owner_reverse)) || num_shards == 0) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->route_packed_device(device, kmers, n, num_shards, owner_forward, owner_reverse, hip_stream); });
}

sshash_status sshash_route_bucket_device(const sshash_dict* d, int device, const uint64_t* kmers, uint64_t n,
                                         uint32_t num_shards, int check_rc, uint64_t* cursors, uint64_t* send,
                                         uint32_t* slots, void* hip_stream) {
    if (!d || (n && !kmers) || !cursors) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->route_bucket_device(device, kmers, n, num_shards, check_rc != 0, false, cursors, send, slots, hip_stream); });
}

sshash_status sshash_route_bucket_by_key_device(const sshash_dict* d, int device, const uint64_t* kmers, uint64_t n,
                                                uint32_t num_shards, uint64_t* cursors, uint64_t* send, uint32_t* slots,
                                                void* hip_stream) {
    if (!d || (n && !kmers) || !cursors) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->route_bucket_device(device, kmers, n, num_shards, true, true, cursors, send, slots, hip_stream); });
}

sshash_status sshash_to_device_table_shard(sshash_dict* d, int device, uint32_t num_table_shards, uint32_t table_shard_id) {
    if (!d) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->to_device(device, num_table_shards, table_shard_id); });
}

sshash_status sshash_route_combine_device(const sshash_dict* d, int device, const uint64_t* replies, const uint32_t* slots,
                                          uint64_t m, uint64_t* out, void* hip_stream) {
    if (!d || (m && (!replies || !slots || !out))) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->route_combine_device(device, replies, slots, m, out, hip_stream); });
}

sshash_status sshash_sharded_lookup_device(const sshash_dict* d, int device, uint32_t num_ranks, int by_table_key,
                                           const uint64_t* kmers, uint64_t n, int check_rc, uint64_t* kmer_ids,
                                           const sshash_exchange* exchange, void* hip_stream) {
    if (!d || !exchange || (n && (!kmers || !kmer_ids))) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] {
        exchange_ops x{exchange->ctx, exchange->counts, exchange->data};
        d->eng->sharded_lookup_device(device, num_ranks, by_table_key != 0, kmers, n, check_rc != 0, kmer_ids, x, hip_stream);
    });
}

sshash_status sshash_sharded_lookup_rccl(const sshash_dict* d, int device, void* nccl_comm, int by_table_key, const uint64_t* kmers,
                                         uint64_t n, int check_rc, uint64_t* kmer_ids, void* hip_stream) {
    if (!d || !nccl_comm || (n && (!kmers || !kmer_ids))) return fail(SSHASH_ERR_ARGUMENT, "null argument");
    return guarded([&] { d->eng->sharded_lookup_rccl(device, nccl_comm, by_table_key != 0, kmers, n, check_rc != 0, kmer_ids, hip_stream); });
}

}  // extern "C"
