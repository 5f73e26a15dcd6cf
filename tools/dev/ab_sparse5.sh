# same-box A/B of wave-kernel variants on BASELINE config 5's share: tools/dev/ab_sparse5.sh "extra bench flags" libA libB ...
mkdir -p gpurun_out/abs5
extra=$1; shift
for r in 1 2; do for L in "$@"; do
  PYCLLP_HIP_LIB=$GRAFT_REPO_ROOT/proflib/$L.so timeout -k 10 200 python bench.py --workload sparse5 $extra --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > /tmp/abs5.json 2>/tmp/abs5.err || tail -3 /tmp/abs5.err
  python -c "
import json; d=json.load(open('/tmp/abs5.json')); print('$L', round(d['value']), d['roofline']['kernel_ms'], d['parity']['ok'], d.get('mean_ipm_iterations'))" | tee -a gpurun_out/abs5/$L.txt
done; done
