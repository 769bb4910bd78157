#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_member_race; mkdir -p $out
for v in nokeep nokeep_first nokeep_skip; do
  SSHASH_AMD_LIBRARY=$PWD/tools/debug/libsshash_amd_$v.so timeout 600 python tools/debug/member_race.py se_k31 20000000 5 2>&1 | grep -v amdgpu.ids
done | tee $out/log.txt
echo "--- INWAVE=0, nokeep" | tee -a $out/log.txt
SSHASH_AMD_INWAVE=0 SSHASH_AMD_LIBRARY=$PWD/tools/debug/libsshash_amd_nokeep.so timeout 600 python tools/debug/member_race.py se_k31 20000000 2 2>&1 | grep -v amdgpu.ids | tee -a $out/log.txt
echo "--- regular build" | tee -a $out/log.txt
timeout 600 python tools/debug/member_race.py se_k31 20000000 2 2>&1 | grep -v amdgpu.ids | tee -a $out/log.txt
