mkdir -p gpurun_out/r3k
timeout -k 10 400 python bench.py > gpurun_out/r3k/bench_line.json 2> gpurun_out/r3k/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3k/bench_line.json'))
print('headline', d['value'], d['roofline']['kernel_ms'], d['parity']['ok'])
for r in d.get('secondary', []):
    print(r.get('workload'), r.get('value'), r.get('error', ''), (r.get('parity') or {}).get('ok'))
PY
for s in 1 2 3; do FUZZ_SEED=$s FUZZ_HSD=1 FUZZ_SIGNED=1 FUZZ_N=30 timeout -k 10 200 python tests/dev/fuzz_gpu.py > gpurun_out/r3k/fuzz_hsd_signed_$s.txt 2>&1; echo "fuzz hsd signed $s rc $?"; grep -c MISMATCH gpurun_out/r3k/fuzz_hsd_signed_$s.txt; done
for s in 1 2; do FUZZ_SEED=$s FUZZ_HSD=1 FUZZ_N=30 timeout -k 10 200 python tests/dev/fuzz_gpu.py > gpurun_out/r3k/fuzz_hsd_$s.txt 2>&1; echo "fuzz hsd $s rc $?"; grep -c MISMATCH gpurun_out/r3k/fuzz_hsd_$s.txt; done
FUZZ_SEED=4 FUZZ_N=30 timeout -k 10 200 python tests/dev/fuzz_gpu.py > gpurun_out/r3k/fuzz_plain_4.txt 2>&1; echo "fuzz plain rc $?"; grep -c MISMATCH gpurun_out/r3k/fuzz_plain_4.txt
