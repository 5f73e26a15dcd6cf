#!/bin/bash
# diagnostic build with in-kernel phase stamps: proflib/libpycllp_hip_prof.so (use with PYCLLP_HIP_LIB=...)
# (the per-problem-A and predictor-corrector wave kernels are linked unstamped from the product build)
set -e
cd /root/repo
mkdir -p proflib
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -DPYCLLP_PROFILE"
C=pycllp_amd/csrc
/opt/rocm/bin/hipcc $F -DWREG_PART=0 -c -o /tmp/wreg_prof.o $C/ipm_wreg.hip &
/opt/rocm/bin/hipcc $F -DWREG_PART=1 -c -o /tmp/wreg_da_prof.o $C/ipm_wreg.hip &
/opt/rocm/bin/hipcc $F $PROF_EXTRA -c -o /tmp/dense_prof.o $C/ipm_dense.hip &
/opt/rocm/bin/hipcc $F -c -o /tmp/big_prof.o $C/ipm_big.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o proflib/libpycllp_hip_prof.so /tmp/dense_prof.o /tmp/wreg_prof.o /tmp/wreg_da_prof.o \
    $C/ipm_wreg_pa.o $C/ipm_wreg_pc.o $C/ipm_wreg_pcda.o $C/ipm_wreg_pcpa.o /tmp/big_prof.o
ls -la proflib/
