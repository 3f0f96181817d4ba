#!/bin/bash
# batch-size sensitivity of the stand-alone NTT (Infinity Cache residency): tools/gpu_ntt_batch.sh <batch> ...
for b in "$@"; do
  timeout -k 10 150 python -u bench.py --workload ntt --batch $b --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernel_ms']
print('batch=$b', round(r['value']), 'limb-NTT/s', {n:round(v['total_ms']/v['launches'],4) for n,v in k.items()})" || exit 1
done
