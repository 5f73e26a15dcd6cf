#!/bin/bash
# disassembly of the (32, 96) dense development build: tools/dev/disasm3296.sh [-Dmacro ...] -> /tmp/d3.dis, prints the kernels' line ranges
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -DPYCLLP_DEV_ONLY_3296 "$@" --cuda-device-only -c -o /tmp/d3.o /root/repo/pycllp_amd/csrc/ipm_dense.hip 2>&1 | grep -v hip-link
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/d3.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/d3.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn /tmp/d3.co > /tmp/d3.dis
grep -n "^[0-9a-f]* <" /tmp/d3.dis | cut -c1-100
