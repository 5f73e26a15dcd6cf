#!/bin/bash
# list the scratch (spill) instructions inside the iteration loop of one wave kernel (default: ipm, MB=8, NQ=6)
K=${1:-ipm_wreg_kernelILi8ELi6E}
mkdir -p /tmp/asm && cd /tmp/asm
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DWREG_PART=${WREG_PART:-0} $EXTRA -S --cuda-device-only -o wreg.s /root/repo/pycllp_amd/csrc/ipm_wreg.hip 2>/dev/null
a=$(grep -n "^_ZN.*${K}.*:" wreg.s | head -1 | cut -d: -f1)
b=$(awk -v a=$a 'NR>a && /^\.Lfunc_end/{print NR; exit}' wreg.s)
sed -n "${a},${b}p" wreg.s > k.s
h=$(grep -n "This Loop Header: Depth=2" k.s | head -1 | cut -d: -f1)
echo "iteration loop starts at line $h of /tmp/asm/k.s; scratch ops inside:"
grep -n "scratch_" k.s | awk -F: -v h=$h '$1>h' | cut -c1-95
