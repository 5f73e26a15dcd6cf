"""Basic-block view of one kernel of an llvm-objdump disassembly: for every block (split at branch targets and branches) its
size and instruction mix, so that the hot path of a persistent loop can be read off and its issue cycles added up.
    python tools/dev/bb_mix.py file.dis <first line> <last line> [min size]"""
import re, sys, collections
path, l0, l1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
minsz = int(sys.argv[4]) if len(sys.argv) > 4 else 1
L = open(path).read().split("\n")[l0:l1]
ins = []
for l in L:
    m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", l)
    if m: ins.append((int(m.group(2), 16), m.group(1)))
targets = set()
for a, t in ins:
    m = re.search(r"s_c?branch\S*\s+(\d+)", t)
    if m: targets.add(a + 4 + 4 * ((int(m.group(1)) + 0x8000) % 0x10000 - 0x8000))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if "f64_dpp" in op or (op.startswith("v_fmac_f64") and "dpp" in op): return "dppf64"
    if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64")): return "f64"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div")): return "f64x"
    if op.startswith(("v_mov_b32_dpp", "v_mov_b64_dpp", "v_permlane", "v_readlane", "v_writelane", "v_readfirstlane")): return "xlane"
    if op.startswith("v_cndmask"): return "sel"
    if op.startswith("v_cmp"): return "cmp"
    if op.startswith("v_"): return "ivalu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "vmem"
cols = ["mfma", "dppf64", "f64", "f64x", "xlane", "sel", "cmp", "ivalu", "lds", "wait", "nop", "salu", "vmem"]
print("%8s %5s " % ("addr", "size") + " ".join("%6s" % c for c in cols) + "   ends with")
blk, start = collections.Counter(), None
def flush(last):
    global blk, start
    n = sum(blk.values())
    if n >= minsz:
        print("%8x %5d " % (start, n) + " ".join("%6d" % blk[c] for c in cols) + "   " + last[:60])
    blk, start = collections.Counter(), None
prev = ""
for a, t in ins:
    if a in targets and start is not None: flush(prev)
    if start is None: start = a
    op = t.split()[0]
    full = t
    blk[cls(op if "dpp" not in full else (op + "_dpp" if not op.endswith("dpp") else op))] += 1
    prev = t
    if op.startswith(("s_branch", "s_cbranch", "s_endpgm")): flush(t)
if start is not None: flush(prev)
