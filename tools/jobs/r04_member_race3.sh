#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_member_race3; mkdir -p $out
for v in asm_lds32k asm_lds20k asm_lds16384 asm_regswap; do
  SSHASH_AMD_LIBRARY=$PWD/tools/debug/libsshash_amd_$v.so timeout 600 python tools/debug/member_race.py se_k31 20000000 3 brief 2>&1 | grep -v amdgpu.ids
done | tee $out/log.txt
SSHASH_AMD_LIBRARY=$PWD/tools/debug/libsshash_amd_asm_base.so timeout 600 python tools/debug/member_race.py se_k31 20000000 3 brief canonical 2>&1 | grep -v amdgpu.ids | tee -a $out/log.txt
