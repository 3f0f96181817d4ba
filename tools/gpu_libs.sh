#!/bin/bash
# compare alternative builds of the library (fhe_amd/lib/<name>.so) on one workload; usage: tools/gpu_libs.sh <workload> <name|""> ...
wl=$1; shift
for v in "$@"; do
  if [ -n "$v" ]; then export FHE_HIP_LIB=$PWD/fhe_amd/lib/$v.so; else unset FHE_HIP_LIB; fi
  timeout -k 10 150 python -u bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('lib=$v', '$wl', round(r['value'],1), {k:round(v['total_ms']/v['launches'],4) for k,v in r['kernel_ms'].items() if 'bconv' in k or 'relin' in k})" || exit 1
done
