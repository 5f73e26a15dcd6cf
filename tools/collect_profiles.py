"""Copies what tools/prof_round.sh left under gpurun_out/<name>/ into profiles/<name>/ (kernel stats, per-launch durations, PMC
summaries with derived fractions, the bench lines printed under rocprofv3) and rewrites profiles/hbm_traffic.json.
usage: python tools/collect_profiles.py r02 <commit the run was made at>"""
import csv, glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, commit = sys.argv[1], sys.argv[2]
G, P = os.path.join(R, "gpurun_out", name), os.path.join(R, "profiles", name)
os.makedirs(P, exist_ok=True)

def newest(pattern):
    """gpurun_out/ accumulates across calls: of several runs under one name, the most recent one counts"""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] 

def rd(f):
    d = {}
    for l in open(f):
        if l.startswith("#") or "mean=" not in l or " n=" not in l:
            continue
        d[l.split()[0]] = float(l.split("mean=")[1])
    return d

traffic = {}
for wl, lps, kern in (("dense3", 65536, "ipm_group_kernel<32,96,slack-aware>"), ("sparse5", 16384, "ipm_wreg_kernel<8,6>"),
                      ("perA", 16384, "ipm_wreg_kernel<8,6,per-problem A>")):
    if not newest(G + "/stats_%s/*/*kernel_stats.csv" % wl):
        continue
    shutil.copy(newest(G + "/stats_%s/*/*kernel_stats.csv" % wl)[0], P + "/kernel_stats_%s.csv" % wl)
    rows = list(csv.DictReader(open(newest(G + "/stats_%s/*/*kernel_trace.csv" % wl)[0])))
    ds = []
    with open(P + "/kernel_trace_durations_%s.txt" % wl, "w") as fo:
        fo.write("# per-launch durations (ms) of the solve kernels, in launch order, from rocprofv3 --kernel-trace of\n# `python3 bench.py %s--no-cpu-baseline` "
                 "(3 warm-up + 10 timed full-size launches, then the parity solve)\n" % ("--workload %s " % wl if wl != "dense3" else ""))
        for r in rows:
            n = r["Kernel_Name"]
            if "ipm_" in n or "hsd_" in n:
                d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                fo.write("%-60s grid %s  %.3f ms\n" % (n[:60], r.get("Grid_Size_X", r.get("Grid_Size", "?")), d)); ds.append(d)
        big = [d for d in ds if d > 0.5 * max(ds)]
        steady = sum(big[-10:]) / len(big[-10:])
        fo.write("# full-size launches: n=%d mean %.3f ms min %.3f max %.3f (steady = last 10: mean %.3f ms)\n" % (len(big), sum(big) / len(big), min(big), max(big), steady))
    for l in open(G + "/stats_%s.log" % wl):
        if l.startswith('{"metric"'):
            open(P + "/bench_line_under_rocprof_%s.json" % wl, "w").write(l)
    d = rd(G + "/pmc_%s/summary.txt" % wl)
    wc = d["SQ_WAVE_CYCLES"]; simd = d["GRBM_GUI_ACTIVE"] / 8 * 1024
    f, w = d["FETCH_SIZE"] * 1024 * 2, d["WRITE_SIZE"] * 1024
    lines = [l.rstrip("\n") for l in open(G + "/pmc_%s/summary.txt" % wl)]
    lines += ["", "# derived (%s, %d LPs per launch, steady launch %.3f ms by rocprofv3 --kernel-trace; commit %s)" % (wl, lps, steady, commit),
              "waves issuing an instruction:   SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = %.1f %%" % (100 * d["SQ_ACTIVE_INST_ANY"] / wc),
              "waves stalled on issue:         SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   = %.1f %%" % (100 * d["SQ_WAIT_INST_ANY"] / wc),
              "waves waiting on LDS:           SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES   = %.1f %%" % (100 * d["SQ_WAIT_INST_LDS"] / wc),
              "VALU share of instructions:     SQ_INSTS_VALU / (VALU+SALU+LDS+VMEM) = %.1f %%" % (100 * d["SQ_INSTS_VALU"] / (d["SQ_INSTS_VALU"] + d["SQ_INSTS_SALU"] + d["SQ_INSTS_LDS"] + d["SQ_INSTS_VMEM"])),
              "f64 FMA + MFMA among VALU:      (FMA_F64 + MFMA_F64) / SQ_INSTS_VALU = %.1f %%" % (100 * (d["SQ_INSTS_VALU_FMA_F64"] + d["SQ_INSTS_VALU_MFMA_F64"]) / d["SQ_INSTS_VALU"]),
              "MFMA busy:                      SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = %.1f %% of SIMD cycles" % (100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / simd),
              "MFMA f64 instructions per LP:   %.1f" % (d["SQ_INSTS_VALU_MFMA_F64"] / lps),
              "LDS bank-conflict cycles:       SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.1f %%" % (100 * d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]),
              "LDS instructions per LP:        %.0f" % (d["SQ_INSTS_LDS"] / lps),
              "HBM traffic per launch:         FETCH_SIZE x 2 (gfx950 correction) = %.1f MB, WRITE_SIZE = %.1f MB, total %.1f MB = %.0f B per LP; %.1f GB/s = %.2f %% of 8 TB/s"
              % (f / 1e6, w / 1e6, (f + w) / 1e6, (f + w) / lps, (f + w) / steady / 1e6, 100 * (f + w) / steady / 1e6 / 8000)]
    open(P + "/pmc_summary_%s.txt" % wl, "w").write("\n".join(lines) + "\n")
    traffic[wl] = {"lps_per_launch": lps, "bytes_per_launch": f + w, "fetch_bytes_corrected_x2": f, "write_bytes": w, "kernel": kern,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/prof_pmc.sh, via tools/prof_round.sh); KB units; FETCH_SIZE "
                             "doubled per the gfx950 correction of MI355X_MICROARCH.md section HBM; means over the full-size launches of the pass",
                   "source": "rocprofv3 PMC passes at commit %s (profiles/%s/pmc_summary_%s.txt); not re-measured in the bench run itself" % (commit, name, wl)}
    print(wl, "steady %.3f ms" % steady, "traffic %.0f B/LP" % ((f + w) / lps))
json.dump(traffic, open(os.path.join(R, "profiles", "hbm_traffic.json"), "w"), indent=1)
