#!/bin/bash
# the in-wave finish under stress: every k-mer of two stand-ins through the id and the is_member instances, several launches;
# then the whole GPU suite three times
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_stress
python tools/debug/member_mismatch.py se_k31 20000000 2>&1 | tail -8
python tools/debug/member_mismatch.py human_k31 300000000 2>&1 | tail -8
for i in 1 2 3; do timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r03_stress/pytest_$i.log 2>&1; tail -2 gpurun_out/r03_stress/pytest_$i.log; done
