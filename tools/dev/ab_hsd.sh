# same-box A/B of HSD dense variants: tools/dev/ab_hsd.sh libA libB ...   (names under proflib/, without .so)
mkdir -p gpurun_out/abhsd
for r in 1 2; do for L in "$@"; do
  echo "== $L (round $r)"
  PYCLLP_HIP_LIB=$GRAFT_REPO_ROOT/proflib/$L.so timeout -k 10 150 python tools/time_hsd.py 2>&1 | grep -v "^$" | tee -a gpurun_out/abhsd/$L.txt
done; done
