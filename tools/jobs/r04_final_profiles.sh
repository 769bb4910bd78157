#!/bin/bash
# the three lookup workloads of the driver's line: bench line, kernel stats, PMC passes -> gpurun_out/r04_prof_{c3,c2,c4}/summary.json (tools/make_traffic_json.py turns them into profiles/traffic.json)
cd "$(dirname "$0")/../.."
for wl in c3 c2 c4; do
  bash tools/jobs/r04_profile.sh r04_prof_$wl --workload $wl 2>&1 | tail -4 | cut -c1-600
done
