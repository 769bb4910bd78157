// hooks.hpp -- SSHASH_AMD_TEST_HOOKS: what the tests use to put a seam, an overflow or a fallback where a small input would have none.
// One variable, "name=value,name=value", read once per C-ABI call (the tests change it between calls inside one process). Not a tuning interface: the
// defaults are the measured ones (RESULTS.md); the hooks and what each one forces are listed in INTEGRATION.md.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>

namespace sshash_amd {

/* The variable is read ONCE PER C-ABI CALL, on the thread that makes the call (capi.cpp: guarded), into a snapshot the library's own
   threads -- the lanes of the host-buffer paths, of the file query -- read instead of the environment: getenv racing with a setenv
   from another thread of the embedding process is undefined behaviour, and round 5 called it from every worker at every use
   (ADVICE r5). What remains is the caller's: the variable must not be changed WHILE a call of this library is in flight. */
struct hook_snapshot {
    std::mutex m;
    std::shared_ptr<const std::string> text;
};
inline hook_snapshot& hook_state() {
    static hook_snapshot s;
    return s;
}
inline void test_hooks_refresh() {
    char const* e = std::getenv("SSHASH_AMD_TEST_HOOKS");
    hook_snapshot& st = hook_state();
    std::lock_guard<std::mutex> lock(st.m);
    if (!e || !*e) st.text.reset();
    else if (!st.text || *st.text != e) st.text = std::make_shared<const std::string>(e);
}

/* value of `name` in the snapshot ("" if absent); `found` says which */
inline std::string test_hook_value(char const* name, bool& found) {
    found = false;
    std::shared_ptr<const std::string> text;
    {
        hook_snapshot& st = hook_state();
        std::lock_guard<std::mutex> lock(st.m);
        text = st.text;
    }
    if (!text) return {};
    const size_t n = std::strlen(name);
    for (char const* p = text->c_str(); *p;) {
        if (std::strncmp(p, name, n) == 0 && p[n] == '=') {
            char const* v = p + n + 1;
            char const* end = std::strchr(v, ',');
            found = true;
            return end ? std::string(v, end) : std::string(v);
        }
        p = std::strchr(p, ',');
        if (!p) break;
        ++p;
    }
    return {};
}

inline uint64_t test_hook_u64(char const* name, uint64_t fallback, uint64_t lo, uint64_t hi) {
    bool found;
    const std::string t = test_hook_value(name, found);
    if (!found) return fallback;
    const uint64_t v = std::strtoull(t.c_str(), nullptr, 10);
    return v >= lo && v <= hi ? v : fallback;
}

inline double test_hook_f64(char const* name, double fallback, double lo, double hi) {
    bool found;
    const std::string t = test_hook_value(name, found);
    if (!found) return fallback;
    const double v = std::strtod(t.c_str(), nullptr);
    return v >= lo && v <= hi ? v : fallback;
}

}  // namespace sshash_amd
