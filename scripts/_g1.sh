set -x
scripts/run.sh sha r6a
date
scripts/run.sh lanes r6a refdefault
scripts/run.sh lanes r6a cfg3
date
PMC_MORE=1 scripts/run.sh pmc r6a refdefault
date
scripts/run.sh pmc r6a cfg3
date
scripts/run.sh pmc r6a cfg5
date
scripts/run.sh lanes r6a cfg5
date
ls gpurun_out | grep r6a
