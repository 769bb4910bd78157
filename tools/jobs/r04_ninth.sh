#!/bin/bash
# streaming kernels: kernel stats + PMC on a high-hit and a low-hit read set (k = 31) and on config C4's (k = 63)
cd "$(dirname "$0")/../.."
bash tools/jobs/r04_profile.sh r04_prof_stream_k31_high --workload c3 --bases 1000000000 --streaming --reads 20000000 --positive 0.95 --stream-oracle-reads 20000 2>&1 | tail -12
bash tools/jobs/r04_profile.sh r04_prof_stream_k31_low --workload c3 --bases 1000000000 --streaming --reads 20000000 --positive 0.0 --stream-oracle-reads 20000 2>&1 | tail -12
bash tools/jobs/r04_profile.sh r04_prof_stream_c4 --workload c4 --streaming --reads 20000000 --stream-oracle-reads 20000 2>&1 | tail -12
