#!/bin/bash
# One round's profile set (run on the GPU box through gpurun): rocprofv3 --kernel-trace --stats of the bench commands (headline
# record only: --no-secondary), then the PMC passes of tools/prof_pmc.sh for each.  Output under gpurun_out/<name>/; copy what
# is to be judged to profiles/ with tools/collect_profiles.py.
# usage: tools/prof_round.sh <name>
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in dense3 sparse5 perA; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -- python3 $R/bench.py --workload $wl --no-cpu-baseline --no-secondary > $OUT/stats_$wl.log 2>&1
  echo "stats $wl done"
done
for wl in dense3 sparse5 perA; do
  bash $R/tools/prof_pmc.sh $1/pmc_$wl --workload $wl --no-secondary
  echo "pmc $wl done"
done
