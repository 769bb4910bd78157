// hooks.hpp -- SSHASH_AMD_TEST_HOOKS: what the tests use to put a seam, an overflow or a fallback where a small input would have none.
// One variable, "name=value,name=value", read at every use (the tests change it inside one process). Not a tuning interface: the
// defaults are the measured ones (RESULTS.md); the hooks and what each one forces are listed in INTEGRATION.md.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace sshash_amd {

inline char const* test_hook_text(char const* name) {
    char const* e = std::getenv("SSHASH_AMD_TEST_HOOKS");
    if (!e) return nullptr;
    const size_t n = std::strlen(name);
    for (char const* p = e; *p;) {
        if (std::strncmp(p, name, n) == 0 && p[n] == '=') return p + n + 1;
        p = std::strchr(p, ',');
        if (!p) break;
        ++p;
    }
    return nullptr;
}

inline uint64_t test_hook_u64(char const* name, uint64_t fallback, uint64_t lo, uint64_t hi) {
    char const* t = test_hook_text(name);
    if (!t) return fallback;
    const uint64_t v = std::strtoull(t, nullptr, 10);
    return v >= lo && v <= hi ? v : fallback;
}

inline double test_hook_f64(char const* name, double fallback, double lo, double hi) {
    char const* t = test_hook_text(name);
    if (!t) return fallback;
    const double v = std::strtod(t, nullptr);
    return v >= lo && v <= hi ? v : fallback;
}

}  // namespace sshash_amd
