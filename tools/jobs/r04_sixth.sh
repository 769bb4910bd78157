#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_sixth; mkdir -p $out
timeout 900 tools/debug/vgpr64_check 67108864 64 more 2>&1 | tee $out/vgpr64_check_other_instructions.jsonl
( time timeout 2400 python -m pytest tests/test_gpu_switches.py -x -q -m gpu ) 2>&1 | tail -15 | tee $out/pytest_switches.txt
( time timeout 3000 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_switches.py ) 2>&1 | tail -15 | tee $out/pytest_gpu.txt
