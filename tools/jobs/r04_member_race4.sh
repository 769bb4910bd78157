#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_member_race4; mkdir -p $out
for v in asm_swap_62_63 asm_swap_56_61; do
  SSHASH_AMD_LIBRARY=$PWD/tools/debug/libsshash_amd_$v.so timeout 600 python tools/debug/member_race.py se_k31 20000000 3 brief 2>&1 | grep -v amdgpu.ids
done | tee $out/log.txt
timeout 600 tools/debug/vgpr64_check 2>&1 | tee $out/vgpr64_check.jsonl
