#!/bin/bash
# shader clock and power while a workload runs: tools/gpu_clock_probe.sh <workload> <steps>
WL=${1:-ntt}; STEPS=${2:-3000}
rm -f gpurun_out/clock_$WL.txt
timeout -k 10 200 python -u bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline > gpurun_out/clock_$WL.json 2>/dev/null &
BP=$!
sleep 14
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|mclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';' | tee -a gpurun_out/clock_$WL.txt; echo | tee -a gpurun_out/clock_$WL.txt
  sleep 1
done
wait $BP
python - <<PY
import json
r=json.load(open("gpurun_out/clock_$WL.json")); print("$WL", round(r["value"],1), {k:round(v["total_ms"]/v["launches"],4) for k,v in r["kernel_ms"].items()})
PY
