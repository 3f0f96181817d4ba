#!/bin/bash
# 8- vs 16-element forward row pass; usage: tools/gpu_rows8.sh <workload>...
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "ntt or rescale or mod_down or rotate_keyswitch" 2>&1 | tail -3 || exit 1
for wl in "$@"; do for v in 0 1; do
  export FHE_HIP_ROWS8=$v
  timeout -k 10 150 python -u bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('rows8=$v', '$wl', round(r['value'],1), {k:round(v['total_ms']/v['launches'],4) for k,v in r['kernel_ms'].items() if 'fwd_rows' in k})" || exit 1
done; done
