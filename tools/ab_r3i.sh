mkdir -p gpurun_out/r3i
timeout -k 10 200 python tests/dev/warm_band.py 6 7 8 > gpurun_out/r3i/warm_carry.txt 2>&1; echo "carry rc $?"
PYCLLP_HIP_LIB=$GRAFT_REPO_ROOT/proflib/libpycllp_hip_nocarry.so timeout -k 10 200 python tests/dev/warm_band.py 6 7 8 > gpurun_out/r3i/warm_nocarry.txt 2>&1; echo "nocarry rc $?"
grep "^seed" gpurun_out/r3i/warm_carry.txt gpurun_out/r3i/warm_nocarry.txt
timeout -k 10 500 python -m pytest tests -m gpu -q > gpurun_out/r3i/gputests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r3i/gputests.log
timeout -k 10 100 python bench.py --no-cpu-baseline --no-secondary --steps 20 > gpurun_out/r3i/bench_carry.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3i/bench_carry.json')); print('carry', d['value'], d['roofline']['kernel_ms'], d['parity']['ok'], d['mean_ipm_iterations'])"
PYCLLP_HIP_LIB=$GRAFT_REPO_ROOT/proflib/libpycllp_hip_nocarry.so timeout -k 10 100 python bench.py --no-cpu-baseline --no-secondary --steps 20 > gpurun_out/r3i/bench_nocarry.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3i/bench_nocarry.json')); print('nocarry', d['value'], d['roofline']['kernel_ms'], d['parity']['ok'], d['mean_ipm_iterations'])"
timeout -k 10 250 python tools/time_big.py > gpurun_out/r3i/time_big.txt 2>&1; cat gpurun_out/r3i/time_big.txt
