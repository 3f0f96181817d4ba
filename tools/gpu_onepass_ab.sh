#!/bin/bash
# one-pass NTT (N <= 2^14) against the pass pair: parity tests, then the c2 bench both ways; usage: tools/gpu_onepass_ab.sh <outdir>
set -o pipefail
OUT=gpurun_out/${1:-onepass}; mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "ntt or one_pass or rescale or mod_down or config2" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for v in 1 0; do
  timeout -k 10 200 python -u bench.py --workload c2 --steps 10 --warmup 2 --no-cpu-baseline --onepass $v > $OUT/c2_onepass$v.json 2> $OUT/c2_onepass$v.err || { tail -20 $OUT/c2_onepass$v.err; exit 1; }
  python - <<P
import json
r=json.load(open("$OUT/c2_onepass$v.json")); print("onepass=$v", round(r["value"]), r["unit"], {k:round(x["total_ms"]/x["launches"],4) for k,x in r["kernel_ms"].items()})
P
done
