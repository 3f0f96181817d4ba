#!/bin/bash
# dependence of the measured rate on the length of the timed region (clock ramp): tools/gpu_steps.sh <workload> <warmup> <steps> ...
WL=$1; W=$2; shift; shift
for s in "$@"; do
  timeout -k 10 250 python -u bench.py --workload $WL --steps $s --warmup $W --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernel_ms']
print('$WL warmup=$W steps=$s', round(r['value'],1), 'ms/step', round(r['ms_per_step'],4))" || exit 1
done
