#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_alltests
timeout 3300 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r03_alltests/pytest.log 2>&1
tail -40 gpurun_out/r03_alltests/pytest.log
