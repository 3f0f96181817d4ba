#!/bin/bash
# timing of ablation builds of the NTT kernels (fhe_amd/lib/abl<N>.so, built with -DNTT_ABL=<N>); usage: tools/gpu_abl.sh "" 1 2 ...
for v in "$@"; do
  if [ -n "$v" ]; then export FHE_HIP_LIB=$PWD/fhe_amd/lib/abl$v.so; else unset FHE_HIP_LIB; fi
  timeout -k 10 100 python -u bench.py --workload ntt --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('ABL=$v', {k:round(v['total_ms']/v['launches'],4) for k,v in r['kernel_ms'].items()})"
done
