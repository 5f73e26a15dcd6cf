mkdir -p gpurun_out/r3f
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "baseline_config or full_size or alternative_kernel or hsd_objective or shapes_up or rank_deficient or warm" > gpurun_out/r3f/sel.log 2>&1; echo "rc $?" >> gpurun_out/r3f/sel.log; tail -4 gpurun_out/r3f/sel.log
timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --steps 20 > gpurun_out/r3f/bench_pipe.json 2>gpurun_out/r3f/bench_pipe.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('gpurun_out/r3f/bench_pipe.json')); print('pipe', d['value'], d['roofline']['kernel_ms'], d['parity'])"
timeout -k 10 120 python tools/time_hsd.py > gpurun_out/r3f/hsd.txt 2>&1; tail -5 gpurun_out/r3f/hsd.txt
