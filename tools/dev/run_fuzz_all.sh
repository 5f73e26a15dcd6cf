# all fuzz sweeps on the current build: tools/dev/run_fuzz_all.sh <outdir under gpurun_out>
O=gpurun_out/$1; mkdir -p $O
for s in 5 6; do FUZZ_SEED=$s FUZZ_N=30 timeout -k 10 250 python tests/dev/fuzz_gpu.py > $O/fuzz_plain_$s.txt 2>&1; echo "plain $s rc $? mismatches $(grep -c MISMATCH $O/fuzz_plain_$s.txt)"; done
for s in 5 6; do FUZZ_SEED=$s FUZZ_HSD=1 FUZZ_N=30 timeout -k 10 250 python tests/dev/fuzz_gpu.py > $O/fuzz_hsd_$s.txt 2>&1; echo "hsd $s rc $? mismatches $(grep -c MISMATCH $O/fuzz_hsd_$s.txt)"; done
for s in 5 6; do FUZZ_SEED=$s FUZZ_HSD=1 FUZZ_SIGNED=1 FUZZ_N=30 timeout -k 10 250 python tests/dev/fuzz_gpu.py > $O/fuzz_hsd_signed_$s.txt 2>&1; echo "hsd signed $s rc $? mismatches $(grep -c MISMATCH $O/fuzz_hsd_signed_$s.txt)"; done
for s in 5 6 7; do FUZZ_SEED=$s FUZZ_N=8 timeout -k 10 500 python tests/dev/fuzz_r3.py > $O/fuzz_r3_$s.txt 2>&1; echo "r3 $s rc $? mismatches $(grep -c MISMATCH $O/fuzz_r3_$s.txt)"; done
true
