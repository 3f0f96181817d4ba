#!/bin/bash
# iteration loop on the GPU: parity tests, then bench lines.  usage: tools/gpu_iter.sh "<pytest -k expr>" <workloads...>
set -o pipefail
K="$1"; shift
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "$K" > gpurun_out/iter_tests.log 2>&1 || { tail -30 gpurun_out/iter_tests.log; exit 1; }
tail -2 gpurun_out/iter_tests.log
for wl in "$@"; do
  timeout -k 10 200 python -u bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/iter_$wl.json 2> gpurun_out/iter_$wl.err || { tail -20 gpurun_out/iter_$wl.err; exit 1; }
  python - <<PY
import json
r = json.load(open("gpurun_out/iter_$wl.json"))
print("$wl", "value=%.1f %s" % (r["value"], r["unit"]), "ms/step=%.3f" % r["ms_per_step"], "whole-op frac=%.3f" % r["roofline_whole_op"]["frac_of_hbm_peak"])
for k, v in r["kernel_ms"].items():
    print("    %-16s launches=%-4d total_ms=%8.3f avg=%.4f" % (k, v["launches"], v["total_ms"], v["total_ms"] / max(1, v["launches"])))
PY
done
