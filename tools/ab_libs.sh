# A/B of alternative builds under proflib/ in ONE box: usage tools/ab_libs.sh libA.so libB.so ...  (headline record only, 3 rounds)
for r in 1 2 3; do for L in "$@"; do
  PYCLLP_HIP_LIB=$GRAFT_REPO_ROOT/proflib/$L timeout -k 10 100 python bench.py --no-cpu-baseline --no-secondary --steps 20 > /tmp/ab.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/ab.json')); print('$L', round(d['value']), d['roofline']['kernel_ms'], d['parity']['ok'])"
done; done
