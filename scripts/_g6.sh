set -x
{
for S in 6 12 16 24; do
  D4GS_BENCH_S=$S bash scripts/ab_run.sh "--config cfg2" base seg16k
done
bash scripts/ab_run.sh "--config cfg2 --channels 4" base seg16k
} 2>&1 | grep "^\[" | tee gpurun_out/r6h_ab_seg_threshold.txt
python scripts/diag_flip_refdefault.py 2>&1 | tail -3 > gpurun_out/r6h_diag.txt
