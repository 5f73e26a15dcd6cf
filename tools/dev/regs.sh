#!/bin/bash
# register / spill figures of the (32, 96) group kernels for a set of macros, without building a library:
#   tools/dev/regs.sh [-Dmacro ...]     (device-only assembly goes to /tmp/regs.s)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -DPYCLLP_DEV_ONLY_3296 "$@" --cuda-device-only -S -o /tmp/regs.s /root/repo/pycllp_amd/csrc/ipm_dense.hip 2>&1 | grep -v "hip-link" 
python - <<'PY'
import re
t = open("/tmp/regs.s").read()
for m in re.finditer(r"\.name:\s+(\S*group_kernel\S*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", t):
    pass
# metadata blocks: fields are alphabetical; parse per block
for blk in t.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if "group_kernel" not in name: continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    print("%-60s agpr %s vgpr %s spill %s scratch %s" % (name[:60], blk.split("\n")[0].strip(), g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size")))
PY
