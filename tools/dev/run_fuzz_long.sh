# a longer fuzz run: tools/dev/run_fuzz_long.sh <outdir under gpurun_out> <first seed> <last seed>
O=gpurun_out/$1; mkdir -p $O
for s in $(seq $2 $3); do
  FUZZ_SEED=$s FUZZ_N=30 timeout -k 10 250 python tests/dev/fuzz_gpu.py > $O/plain_$s.txt 2>&1; a=$(grep -c MISMATCH $O/plain_$s.txt)
  FUZZ_SEED=$s FUZZ_HSD=1 FUZZ_N=30 timeout -k 10 250 python tests/dev/fuzz_gpu.py > $O/hsd_$s.txt 2>&1; b=$(grep -c MISMATCH $O/hsd_$s.txt)
  FUZZ_SEED=$s FUZZ_HSD=1 FUZZ_SIGNED=1 FUZZ_N=30 timeout -k 10 250 python tests/dev/fuzz_gpu.py > $O/signed_$s.txt 2>&1; c=$(grep -c MISMATCH $O/signed_$s.txt)
  FUZZ_SEED=$s FUZZ_N=6 timeout -k 10 500 python tests/dev/fuzz_r3.py > $O/r3_$s.txt 2>&1; d=$(grep -c MISMATCH $O/r3_$s.txt)
  echo "seed $s mismatches: plain $a hsd $b signed $c r3 $d"
done
true
