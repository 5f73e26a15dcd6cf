#!/bin/bash
# development variant of the library with ONLY the (8, 6) table variant of the wave kernel rebuilt (BASELINE config 5):
#   tools/dev/mkvariant_w.sh NAME [-Dmacro ...]  ->  proflib/NAME.so   (the other objects come from the product build)
set -e
cd /root/repo
name=$1; shift
mkdir -p proflib
C=pycllp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -DWREG_PART=0 -DPYCLLP_DEV_ONLY_W86 "$@" -c -o /tmp/wreg_$name.o $C/ipm_wreg.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o proflib/$name.so $C/ipm_dense.o /tmp/wreg_$name.o $C/ipm_wreg_da.o $C/ipm_wreg_pa.o \
    $C/ipm_wreg_pc.o $C/ipm_wreg_pcda.o $C/ipm_wreg_pcpa.o $C/ipm_big.o
echo built proflib/$name.so
