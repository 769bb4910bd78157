#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_member_race2; mkdir -p $out
for v in asm_base asm_vgpr72 asm_nops_loop asm_nops_wide asm_loop_load_plain asm_all_loads_plain asm_nops_at_handover; do
  SSHASH_AMD_LIBRARY=$PWD/tools/debug/libsshash_amd_$v.so timeout 600 python tools/debug/member_race.py se_k31 20000000 3 brief 2>&1 | grep -v amdgpu.ids
done | tee $out/log.txt
