# what the driver runs at round end, in one GPU call: tools/dev/final_check.sh <tag>  (outputs under gpurun_out/)
T=${1:-last}
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_$T.log 2>&1; tail -2 gpurun_out/gputests_$T.log \
 && python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" \
 && python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err \
 && python -c "
import json; d=json.loads(open('gpurun_out/bench_$T.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], all(s['parity']['ok'] for s in d['secondary']))"
