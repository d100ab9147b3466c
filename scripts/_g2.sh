set -x
scripts/run.sh sha r6b
timeout 2400 python -m pytest tests/test_gpu_flip_cause.py tests/test_gpu_refdefault_fullsize.py tests/test_gpu_threads.py tests/test_c_abi_demo.py "tests/test_gpu_frame.py::test_exact_tiles_auto_turning_off_under_deferred_size_check_counts_again" -q -m gpu -p no:cacheprovider 2>&1 | tail -60 | tee gpurun_out/r6b_pytest_new.txt
cp gpurun_out/flip_cause.json gpurun_out/r6b_flip_cause.json
date
for c in cfg2 refdefault cfg3 cfg5; do
  scripts/run.sh bench r6b_$c --config $c
  scripts/run.sh prof r6b $c
  date
done
