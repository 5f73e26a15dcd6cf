# same-box A/B of dense (32, 96) variants: tools/dev/ab_dense3.sh "what..." libA libB ...   (names under proflib/, without .so)
mkdir -p gpurun_out/abd3
what=$1; shift
for r in 1 2; do for L in "$@"; do
  echo "== $L (round $r)"
  PYCLLP_HIP_LIB=$GRAFT_REPO_ROOT/proflib/$L.so timeout -k 10 150 python tools/dev/time_dense3.py $what 2>&1 | grep config3\\\|mixed | tee -a gpurun_out/abd3/$L.txt
done; done
