#!/bin/bash
# the GPU suite under forced modes (lazy far sort, sparse gradient rows, exact tiles, both) on the final library
mkdir -p gpurun_out
out=gpurun_out/r05s_pytest_gpu_forced_modes.txt
: > $out
for m in "D4GS_LAZY_SORT=1" "D4GS_BWD_ROWS=sparse" "D4GS_EXACT_TILES=1" "D4GS_EXACT_TILES=1 D4GS_LAZY_SORT=1"; do
  echo "== $m" >> $out
  env $m timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -4 >> $out
done
sha256sum deblur4dgs_amd/libd4gs.so >> $out
cat $out
