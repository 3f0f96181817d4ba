#!/usr/bin/env python3
"""Copy the outputs of tools/gpu_final_refresh.sh <tag> (merged under gpurun_out/) into profiles/r01 and profiles/traffic_*.json,
and print the results table rows for BASELINE.md.  usage: tools/collect_profiles.py <tag>"""
import glob, json, os, shutil, subprocess, sys
tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
dst = os.path.join("profiles", "r01")
for wl in ("c3", "ntt", "c4"):
    stats = sorted(glob.glob(os.path.join(src, "stats_%s" % wl, "*", "*kernel_stats.csv")), key=os.path.getsize)[-1]
    shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % wl))
    shutil.copy(os.path.join(src, "pmc_summary_%s.txt" % wl), os.path.join(dst, "%s_pmc_fetch_write.txt" % wl))
    t = json.load(open(os.path.join(src, "traffic_%s.json" % wl)))
    t["source"] = "profiles/r01/%s_pmc_fetch_write.txt" % wl
    json.dump(t, open(os.path.join("profiles", "traffic_%s.json" % wl), "w"), indent=1)
for wl in ("c2", "c3", "c4", "c5", "ntt", "rotm", "rotnn", "dec", "enc"):
    shutil.copy(os.path.join(src, "bench_%s.json" % wl), os.path.join(dst, "bench_%s.json" % wl))
for wl in ("c3", "ntt"):
    out = subprocess.run([sys.executable, "tools/pmc_summary.py", os.path.join("gpurun_out", "%s_pmc_%s" % (tag, wl))], capture_output=True, text=True).stdout
    open(os.path.join(dst, "%s_pmc_sq_valu.txt" % wl), "w").write(out)
for wl in ("c3", "ntt", "c4", "c5", "c2", "rotm", "rotnn", "dec", "enc"):
    r = json.load(open(os.path.join(dst, "bench_%s.json" % wl)))
    rf, cb = r["roofline"], r["cpu_baseline"]
    print("%-6s %10.1f %-12s whole %.3f  kernel %6.0f GB/s (%.3f)  traffic %s  alg/launch %.3e  cpu %.2f / %.1f" % (
        wl, r["value"], r["unit"], r["roofline_whole_op"]["frac_of_hbm_peak"], rf["achieved"], rf["frac"], rf.get("traffic"),
        rf["alg_bytes_per_launch"], cb["value"], cb["all_cores"]["value"]))
