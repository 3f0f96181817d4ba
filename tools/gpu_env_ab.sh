#!/bin/bash
# A/B of an environment switch of the library over bench workloads; usage: tools/gpu_env_ab.sh <ENVVAR> <pytest -k expr> <workload>...
var=$1; k=$2; shift 2
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "$k" 2>&1 | tail -3 || exit 1
for wl in "$@"; do for v in 0 1; do
  export $var=$v
  timeout -k 10 150 python -u bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$var=$v', '$wl', round(r['value'],1), {k:round(v['total_ms']/v['launches'],4) for k,v in r['kernel_ms'].items() if 'ntt' in k or 'rot' in k})" || exit 1
done; done
