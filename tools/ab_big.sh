mkdir -p gpurun_out/abbig
echo "== A (default: 2 WG/CU)"; timeout -k 10 200 python tools/time_big.py 2>&1 | grep "LPs/s" | cut -c1-110
for V in E F; do echo "== $V"; PYCLLP_HIP_LIB=$GRAFT_REPO_ROOT/proflib/libbig_$V.so timeout -k 10 200 python tools/time_big.py 2>&1 | grep "LPs/s" | cut -c1-110; done
