#!/usr/bin/env python3
"""Copy the outputs of tools/gpu_final_refresh.sh <tag> (merged under gpurun_out/<tag>) into profiles/<round> and
profiles/{traffic,valu}_*.json, and print one summary row per workload.  usage: tools/collect_profiles.py <tag> <round>"""
import glob, json, os, shutil, sys
tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
dst = os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
for wl in [w for w in ("c3", "ntt", "c4", "c2") if os.path.exists(os.path.join(src, "%s_kernel_stats.csv" % w))]:
    shutil.copy(os.path.join(src, "%s_kernel_stats.csv" % wl), os.path.join(dst, "%s_kernel_stats.csv" % wl))
    shutil.copy(os.path.join(src, "pmc_summary_%s.txt" % wl), os.path.join(dst, "%s_pmc_fetch_write.txt" % wl))
    shutil.copy(os.path.join(src, "pmc_valu_summary_%s.txt" % wl), os.path.join(dst, "%s_pmc_sq_valu.txt" % wl))
    for kind in ("traffic", "valu"):
        t = json.load(open(os.path.join(src, "%s_%s.json" % (kind, wl))))
        t["source"] = "profiles/%s/%s_pmc_%s.txt" % (rnd, wl, "fetch_write" if kind == "traffic" else "sq_valu")
        json.dump(t, open(os.path.join("profiles", "%s_%s.json" % (kind, wl)), "w"), indent=1)
for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
    r = json.load(open(f))
    rf = r.get("roofline") or {}
    rv = r.get("roofline_valu") or {}
    cb = r.get("cpu_baseline") or {}
    print("%-22s %12.1f %-12s whole %.3f  kernel %6.0f GB/s (%.3f)  valu %.2f  cpu %s / %s" % (
        os.path.basename(f)[6:-5], r["value"], r["unit"], r["roofline_whole_op"]["frac_of_hbm_peak"], rf.get("achieved", 0), rf.get("frac", 0),
        rv.get("frac", 0), cb.get("value"), (cb.get("all_cores") or {}).get("value")))
