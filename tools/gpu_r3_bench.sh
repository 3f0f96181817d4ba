#!/bin/bash
# round-3 iteration: GPU parity tests, then bench lines for the given "workload:primes" pairs (e.g. ntt:sp ntt:generic c3:sp)
# usage: tools/gpu_r3_bench.sh <tag> <pytest -k expr | "" | skip> <workload:primes[:extra+bench+args]>...
set -o pipefail
TAG="$1"; K="$2"; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ "$K" != "skip" ]; then
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread ${K:+-k "$K"} > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
  tail -2 $OUT/tests.log
fi
for wp in "$@"; do
  IFS=: read -r wl pk extra <<< "$wp"       # workload:primes[:extra+bench+args]
  extra=${extra//+/ }
  name=${wl}_${pk}$(echo "$extra" | tr -c 'a-zA-Z0-9\n' '_' | sed 's/_*$//')
  pk0=$pk; pk=$name
  timeout -k 10 240 python -u bench.py --workload $wl --primes $pk0 $extra --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-ntt-n16 > $OUT/bench_${pk}.json 2> $OUT/bench_${pk}.err || { tail -20 $OUT/bench_${pk}.err; exit 1; }
  python - <<PY
import json
r = json.load(open("$OUT/bench_${pk}.json"))
print("$pk", "value=%.1f %s" % (r["value"], r["unit"]), "ms/step=%.3f" % r["ms_per_step"], "whole-op frac=%.3f" % r["roofline_whole_op"]["frac_of_hbm_peak"])
for k, v in r["kernel_ms"].items():
    print("    %-22s launches=%-4d total_ms=%8.3f avg=%.4f" % (k, v["launches"], v["total_ms"], v["total_ms"] / max(1, v["launches"])))
PY
done
