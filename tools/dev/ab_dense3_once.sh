# one round of tools/dev/time_dense3.py per library: tools/dev/ab_dense3_once.sh "what" libA libB ...
what=$1; shift
for L in "$@"; do
  echo "== $L"
  PYCLLP_HIP_LIB=$GRAFT_REPO_ROOT/proflib/$L.so timeout -k 10 150 python tools/dev/time_dense3.py $what 2>&1 | grep config3
done
