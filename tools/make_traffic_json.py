#!/usr/bin/env python3
"""Turn a tools/pmc_summary.py text (FETCH_SIZE / WRITE_SIZE per kernel, KB, max over dispatches) into the
profiles/traffic_<workload>.json that bench.py reads.  usage: make_traffic_json.py <summary.txt> <workload> <label> <out.json> <kernel-substr>...
HBM bytes per launch = sum over the named kernels of (2 * FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests at 64 B); the largest dispatch of each kernel is taken (the bench's
dominant launch).  hbm_bytes_per_step = the same correction summed over EVERY dispatch of every kernel of the timed steps
(key upload and index-table kernels excluded), divided by the number of steps the pass ran = dispatches of the first named
kernel / its launches per step (env LAUNCHES_PER_STEP, default 1)."""
import json, os, re, sys
summary, workload, label, out = sys.argv[1:5]
names = sys.argv[5:]
cur, vals, sums, counts = None, {}, {}, {}
for line in open(summary):
    if not line.startswith(" "):
        cur = line.strip()
        continue
    m = re.match(r"\s+(\S+)\s+n=(\d+)\s+max=(\S+)\s+median=(\S+)(?:\s+sum=(\S+))?", line)
    if m and cur:
        vals.setdefault(cur, {})[m.group(1)] = float(m.group(3))
        counts.setdefault(cur, {})[m.group(1)] = int(m.group(2))
        if m.group(5):
            sums.setdefault(cur, {})[m.group(1)] = float(m.group(5))
fetch, write, used = [], [], []
for n in names:
    hit = [k for k in vals if n in k]
    if not hit:
        sys.exit("kernel %r not found in %s" % (n, summary))
    k = hit[0]
    used.append(k); fetch.append(vals[k]["FETCH_SIZE"]); write.append(vals[k]["WRITE_SIZE"])
total = int(sum(2 * f + w for f, w in zip(fetch, write)) * 1024)
per_step, per_kernel = None, {}
if sums:
    steps = counts[used[0]]["FETCH_SIZE"] / float(os.environ.get("LAUNCHES_PER_STEP", "1"))
    skip = ("to_mont_kernel", "galois_perm_kernel")
    for k, c in sums.items():
        if any(x in k for x in skip) or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        per_kernel[k] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / steps)
    per_step = sum(per_kernel.values())
json.dump({"kernel": " + ".join(used), "workload": label, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
           "hbm_bytes_per_step": per_step, "hbm_bytes_per_step_by_kernel": per_kernel,
           "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is",
           "hbm_bytes_per_launch": total, "source": "profiles/%s" % summary.split("gpurun_out/")[-1]}, open(out, "w"), indent=1)
print(workload, total)
