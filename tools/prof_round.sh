#!/bin/bash
# One round's profile set (run on the GPU box through gpurun): rocprofv3 --kernel-trace --stats of the two bench commands,
# then the PMC passes of tools/prof_pmc.sh for both.  Output under gpurun_out/<name>/; copy what is to be judged to profiles/.
# usage: tools/prof_round.sh <name>
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dense3 -- python3 $R/bench.py --no-cpu-baseline > $OUT/stats_dense3.log 2>&1
echo "stats dense3 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sparse5 -- python3 $R/bench.py --workload sparse5 --no-cpu-baseline > $OUT/stats_sparse5.log 2>&1
echo "stats sparse5 done"
bash $R/tools/prof_pmc.sh $1/pmc_dense3
echo "pmc dense3 done"
bash $R/tools/prof_pmc.sh $1/pmc_sparse5 --workload sparse5
echo "pmc sparse5 done"
