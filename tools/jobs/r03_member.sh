#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_member
for v in 1 0; do SSHASH_AMD_INWAVE=$v python tools/debug/member_mismatch.py se_k31 20000000 2>&1 | tail -10; done | tee gpurun_out/r03_member/log.txt
