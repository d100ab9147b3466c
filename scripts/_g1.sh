set -x
scripts/run.sh sha r6a
date
scripts/run.sh suite r6a
date
scripts/run.sh bench r6a
date
for c in cfg2 refdefault cfg3 cfg5; do
  scripts/run.sh lanes r6a $c
  PMC_MORE=1 scripts/run.sh pmc r6a $c
  date
done
ls gpurun_out | grep r6a
