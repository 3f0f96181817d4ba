#!/bin/bash
# batch-size sweep of one bench workload; usage: tools/gpu_batch.sh <workload> <batch...>
wl=$1; shift
for b in "$@"; do
  timeout -k 10 150 python -u bench.py --workload $wl --batch $b --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$wl batch=$b', round(r['value'],1), r['unit'], 'ms/step', round(r['ms_per_step'],3))" || exit 1
done
