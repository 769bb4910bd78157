// errors.hpp -- the one exception type that crosses module boundaries inside the library; the C ABI
// (capi.cpp) maps `kind` onto sshash_status. The reference reports every failure as std::runtime_error
// (include/util.hpp:191-195, src/query.cpp:128); messages keep its wording where it has one.
#pragma once

#include <stdexcept>
#include <string>

namespace sshash_amd {

enum class error_kind : int { argument = 1, io = 2, format = 3, version = 4, no_device = 5, hip = 6, build = 7, internal = 8 };

struct error : std::runtime_error {
    error_kind kind;
    error(error_kind k, std::string const& message) : std::runtime_error(message), kind(k) {}
};

}  // namespace sshash_amd
