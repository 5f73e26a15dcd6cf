"""Where does a kernel spill?  Reads device assembly (hipcc --cuda-device-only -S) and prints, for one kernel, a strip chart:
per bucket of 200 instructions the counts of landmarks (MFMA, DPP FMA, ds_read/ds_write, swizzle) and scratch loads/stores.
    python tools/dev/spill_map.py file.s 'hsd_group_kernelILi32ELi96ELb1'"""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if (l.startswith("_Z") and pat in l.split(":")[0]))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
ins = [l.strip() for l in lines[start:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
print("kernel at line %d, %d instructions" % (start, len(ins)))
keys = [("mfma", "v_mfma"), ("dppfma", "v_fmac_f64_dpp"), ("dsr", "ds_read"), ("dsw", "ds_write"), ("swz", "ds_swizzle"),
        ("perm", "v_permlane"), ("rcp", "v_rcp_f64"), ("sld", "scratch_load"), ("sst", "scratch_store"), ("br", "s_cbranch"), ("gld", "global_load")]
print("%6s " % "instr" + " ".join("%6s" % k for k, _ in keys))
for b0 in range(0, len(ins), bucket):
    seg = ins[b0:b0 + bucket]
    print("%6d " % b0 + " ".join("%6d" % sum(1 for s in seg if s.startswith(p)) for _, p in keys))
