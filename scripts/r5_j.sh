#!/bin/bash
# round-5 batch J: the GPU suite with non-default modes forced on the final build
mkdir -p gpurun_out
for v in "D4GS_LAZY_SORT=1" "D4GS_BWD_ROWS=sparse" "D4GS_EXACT_TILES=1" "D4GS_EXACT_TILES=1 D4GS_LAZY_SORT=1"; do
  echo "== $v"
  env $v timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8
done 2>&1 | tee gpurun_out/r05s_pytest_gpu_forced_modes.txt
