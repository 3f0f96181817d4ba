#!/usr/bin/env python3
"""profiles/valu_<workload>.json from a rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE pass (raw *counter_collection.csv under
<dir>): per named kernel the median over its largest dispatches of SQ_INSTS_VALU (wave-instructions), GRBM_GUI_ACTIVE
(rocprofv3 reports the sum over the 8 XCDs: divided by 8 here) and the dispatch duration; several kernels are summed (a pair
that forms one launch unit, e.g. the two NTT passes).  bench.py turns this into `roofline_valu`.
usage: make_valu_json.py <pmc-dir> <workload> <label> <out.json> <kernel-substr>..."""
import collections, csv, glob, json, os, statistics, sys
root, workload, label, out = sys.argv[1:5]
names = sys.argv[5:]
rows = collections.defaultdict(dict)   # (file, dispatch) -> {counter: value, name, grid, dur}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (f, r["Dispatch_Id"])
        rows[k]["name"] = r["Kernel_Name"]
        rows[k]["grid"] = int(r["Grid_Size"])
        rows[k]["dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        rows[k][r["Counter_Name"]] = float(r["Counter_Value"])
tot = {"insts": 0.0, "gui": 0.0, "ms": 0.0}
used = []
for n in names:
    d = [v for v in rows.values() if n in v["name"] and "SQ_INSTS_VALU" in v and "GRBM_GUI_ACTIVE" in v]
    if not d:
        sys.exit("kernel %r not found under %s" % (n, root))
    g = max(v["grid"] for v in d)
    d = [v for v in d if v["grid"] == g]
    used.append("%s (grid %d, %d dispatches)" % (d[0]["name"].split("(")[0], g, len(d)))
    tot["insts"] += statistics.median(v["SQ_INSTS_VALU"] for v in d)
    tot["gui"] += statistics.median(v["GRBM_GUI_ACTIVE"] for v in d) / 8.0
    tot["ms"] += statistics.median(v["dur_ns"] for v in d) * 1e-6
json.dump({"kernel": " + ".join(used), "workload": label, "insts_valu_per_launch": tot["insts"],
           "gui_active_per_xcd_per_launch": tot["gui"], "launch_ms_in_pmc_pass": tot["ms"],
           "note": "SQ_INSTS_VALU counts wave-instructions; GRBM_GUI_ACTIVE is reported summed over the 8 XCDs and divided by 8 here "
                   "(MI355X_MICROARCH.md, DVFS give-back); medians over the largest dispatches of each kernel",
           "source": "profiles/%s" % root.split("gpurun_out/")[-1]}, open(out, "w"), indent=1)
print(workload, tot)
