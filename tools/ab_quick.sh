# quick A/B helper (GPU box): selected parity tests + the headline record only
mkdir -p gpurun_out/abq
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "baseline_config or full_size or alternative_kernel or shapes_up or warm or predictor or ragged" > gpurun_out/abq/sel.log 2>&1; echo "rc $?"; tail -3 gpurun_out/abq/sel.log
for i in 1 2; do timeout -k 10 100 python bench.py --no-cpu-baseline --no-secondary --steps 20 > gpurun_out/abq/bench_$i.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/abq/bench_$i.json')); print('headline', d['value'], d['roofline']['kernel_ms'], d['parity']['ok'], d['mean_ipm_iterations'])"; done
