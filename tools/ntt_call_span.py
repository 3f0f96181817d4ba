#!/usr/bin/env python3
"""Per-call span of the two-stream NTT from a rocprofv3 kernel trace: the chunks of one fhe_hip_ntt_forward call run as
`chunks` column-pass launches and as many row-pass launches on two streams, overlapping each other, so per-kernel durations no
longer add up to the call; the unit that does is first launch start -> last launch end of a call.
usage: ntt_call_span.py <kernel_trace.csv> <chunks per call>"""
import csv, statistics, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ntt_cols_kernel" in r["Kernel_Name"] or "ntt_rows8_fwd_kernel" in r["Kernel_Name"]]
chunks = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per_call = 2 * chunks
spans, sums = [], []
for i in range(0, len(rows) - per_call + 1, per_call):
    grp = rows[i:i + per_call]
    spans.append((max(int(r["End_Timestamp"]) for r in grp) - int(grp[0]["Start_Timestamp"])) * 1e-3)
    sums.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp) * 1e-3)
tail = spans[len(spans) // 2:]  # the second half of the run: past the clock ramp
print("calls %d (of %d launches each); span per call, second half of the run: median %.1f us, min %.1f us; sum of the overlapping kernel "
      "durations per call: median %.1f us" % (len(spans), per_call, statistics.median(tail), min(tail), statistics.median(sums[len(sums) // 2:])))
