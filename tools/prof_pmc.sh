#!/bin/bash
# PMC passes for the solve kernel (run on the GPU box through gpurun): each counter group in its own rocprofv3 run,
# --pmc never combined with trace domains other than --kernel-trace.
# usage: tools/prof_pmc.sh <outdir-under-gpurun_out> [bench args]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F64" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
vals=collections.defaultdict(list)
for f in glob.glob("$OUT/pass*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if 'ipm_' in row['Kernel_Name'] and 'kernel' in row['Kernel_Name'] and 'pack' not in row['Kernel_Name']:
            vals[row['Counter_Name']].append(float(row['Counter_Value']))
with open("$OUT/summary.txt","w") as fo:
    fo.write("# per-launch means over the FULL-SIZE launches only (values > half of the maximum seen)\n")
    for k,v in sorted(vals.items()):
        mx=max(v); sel=[x for x in v if x>0.5*mx] if mx>0 else v
        line="%-28s n=%d mean=%.6g"%(k,len(sel),sum(sel)/len(sel)); print(line); fo.write(line+"\n")
PY
