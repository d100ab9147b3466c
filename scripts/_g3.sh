set -x
date
scripts/run.sh sha r6i
scripts/run.sh suite r6i --durations=12
date
scripts/run.sh frames r6i "--config cfg2" "--config cfg2 --graph" "--config cfg3" "--config cfg5" "--config refdefault" "--config cfg2 --share 2" "--config cfg2 --share 4" "--config cfg2 --share 8" "--config cfg2 --scale-mul 4" "--config cfg1"
date
