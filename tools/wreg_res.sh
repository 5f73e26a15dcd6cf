#!/bin/bash
# compile ipm_wreg.hip only and print the per-kernel register/scratch summary (no GPU needed)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -DWREG_PART=${WREG_PART:-0} $EXTRA -c -o /tmp/ipm_wreg.o /root/repo/pycllp_amd/csrc/ipm_wreg.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep "error\|Function Name\|Scratch\|VGPRs Spill" | grep -v selftest | sed 's/.*remark: //; s/\[-Rpass.*//; s/_ZN12_GLOBAL__N_1//'
