#!/bin/bash
# the switches that are kept for A/B runs must keep working: the whole GPU suite with the round-2 resume pass, and with the tail passes on the auxiliary stream
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_altpaths
SSHASH_AMD_INWAVE=0 timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r03_altpaths/pytest_inwave0.log 2>&1; tail -2 gpurun_out/r03_altpaths/pytest_inwave0.log
SSHASH_AMD_INWAVE=0 SSHASH_AMD_OVERLAP=1 timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r03_altpaths/pytest_overlap1.log 2>&1; tail -2 gpurun_out/r03_altpaths/pytest_overlap1.log
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r03_altpaths/pytest_default.log 2>&1; tail -2 gpurun_out/r03_altpaths/pytest_default.log
