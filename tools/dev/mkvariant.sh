#!/bin/bash
# development variant of the library for same-box A/B runs: tools/dev/mkvariant.sh NAME [-Dmacro ...]
# builds ONLY the (32, 96) group kernels of ipm_dense.hip (PYCLLP_DEV_ONLY_3296) with the given macros and links them with the
# product's other objects into proflib/NAME.so (use with PYCLLP_HIP_LIB=$GRAFT_REPO_ROOT/proflib/NAME.so)
set -e
cd /root/repo
name=$1; shift
mkdir -p proflib
C=pycllp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -DPYCLLP_DEV_ONLY_3296 "$@" -c -o /tmp/dense_$name.o $C/ipm_dense.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o proflib/$name.so /tmp/dense_$name.o $C/ipm_wreg.o $C/ipm_wreg_da.o $C/ipm_wreg_pa.o \
    $C/ipm_wreg_pc.o $C/ipm_wreg_pcda.o $C/ipm_wreg_pcpa.o $C/ipm_big.o
echo built proflib/$name.so
