// sktable.hpp -- builds the super-k-mer table of device_layout.hpp (5) for a replica whose atoms and
// endpoints are already resident (internal header).
#pragma once

#include "index.hpp"

namespace sshash_amd {

struct device_replica;

/* Leaves rep.view.sk disabled when the table does not apply (a minimizer shard, SSHASH_AMD_SKTABLE=0,
   not enough free HBM). Runs on the current device. */
void build_sk_table(device_replica& rep, host_index const& idx);

}  // namespace sshash_amd
