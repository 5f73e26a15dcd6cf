#!/usr/bin/env python
"""Register / spill / scratch metadata of the kernels in the SHIPPED library, read from its gfx950 code objects
(a claim like "no scratch inside the iteration loop" must be checkable against the binary).

    python tools/kernel_resources.py [pattern ...] > profiles/rNN/kernel_resources.txt

Unbundles every gfx950 code object of pycllp_amd/csrc/libpycllp_hip.so (clang-offload-bundler), reads the AMDGPU
metadata note (llvm-readelf --notes) and prints, per kernel: VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs, scratch
bytes per lane, LDS bytes, and -- from the disassembly -- the number of scratch_* instructions in the kernel.  No GPU needed.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pycllp_amd", "csrc", "libpycllp_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(anonymous namespace\)::", "", o).split("(")[0].replace("void ", "") for o in out[:len(names)]]


def code_objects(lib, tmp):
    """The gfx950 device code objects embedded in the host library (.hip_fatbin holds one offload bundle per TU)."""
    sec = os.path.join(tmp, "fatbin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, sec])
    data = open(sec, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    outs, pos, k = [], 0, 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        j = data.find(magic, i + 1)
        blob = data[i:j if j > 0 else len(data)]
        bpath = os.path.join(tmp, "bundle%d" % k)
        open(bpath, "wb").write(blob)
        opath = os.path.join(tmp, "co%d.o" % k)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + bpath,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + opath], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(opath) and os.path.getsize(opath) > 0:
            outs.append(opath)
        pos, k = i + 1, k + 1
    return outs


def kernels_of(co):
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    recs = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, "0"])[1]
        recs.append(dict(name=g("name"), vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), sgpr=int(g("sgpr_count")),
                         vspill=int(g("vgpr_spill_count")), sspill=int(g("sgpr_spill_count")),
                         scratch=int(g("private_segment_fixed_size")), lds=int(g("group_segment_fixed_size"))))
    return recs


def scratch_ops(co):
    """scratch_load / scratch_store instruction counts per kernel symbol, from the disassembly."""
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    counts, cur = {}, None
    for line in dis.split("\n"):
        mm = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if mm:
            cur = mm.group(1)
            counts[cur] = [0, 0]
        elif cur and "scratch_load" in line:
            counts[cur][0] += 1
        elif cur and "scratch_store" in line:
            counts[cur][1] += 1
    return counts


def main():
    pats = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        rows = []
        for co in code_objects(LIB, tmp):
            ks = kernels_of(co)
            ops = scratch_ops(co)
            for k in ks:
                k["ops"] = ops.get(k["name"], [0, 0])
            rows += ks
    names = demangle([r["name"] for r in rows])
    print("# %s: per-kernel resources from the gfx950 code objects of the shipped library" % os.path.relpath(LIB, ROOT))
    print("# %-58s %5s %5s %5s %7s %7s %9s %8s %s" % ("kernel", "VGPR", "AGPR", "SGPR", "vspill", "sspill", "scratch B", "LDS B", "scratch_load/store instr"))
    for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
        if pats and not any(p in n for p in pats):
            continue
        print("%-60s %5d %5d %5d %7d %7d %9d %8d %d/%d" % (n[:60], r["vgpr"], r["agpr"], r["sgpr"], r["vspill"], r["sspill"],
                                                          r["scratch"], r["lds"], r["ops"][0], r["ops"][1]))


if __name__ == "__main__":
    main()
