# same-box A/B of two full libraries over the dense lane-group shapes: tools/dev/ab_shapes.sh libA libB  (names under proflib/)
mkdir -p gpurun_out/abshapes
for sh in "16 16 65536" "16 32 262144" "16 48 65536" "32 32 65536" "32 64 65536" "32 96 65536" "24 40 65536"; do for L in "$@"; do
  echo -n "$L  "; PYCLLP_HIP_LIB=$GRAFT_REPO_ROOT/proflib/$L.so timeout -k 10 120 python tools/dev/time_dense_shape.py $sh 2>&1 | grep "LPs/s" | cut -c1-150 | tee -a gpurun_out/abshapes/$L.txt
done; done
