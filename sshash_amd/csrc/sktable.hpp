// sktable.hpp -- builds the super-k-mer table of device_layout.hpp (5) for a replica whose atoms and
// endpoints are already resident (internal header).
#pragma once

#include "index.hpp"

namespace sshash_amd {

struct device_replica;

/* Leaves rep.view.sk disabled when the table does not apply (a minimizer shard, SSHASH_AMD_SKTABLE=0,
   not enough free HBM). Runs on the current device. */
/* table_shards > 1: only the keys with sk_owner(key, table_shards) == table_shard_id get slots */
/* SSHASH_AMD_HBM_BUDGET: bytes one replica may hold in HBM (0: no limit) */
uint64_t hbm_budget();

void build_sk_table(device_replica& rep, host_index const& idx, uint32_t table_shards = 1, uint32_t table_shard_id = 0);

}  // namespace sshash_amd
