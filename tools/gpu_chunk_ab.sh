#!/bin/bash
# A/B of the cache-resident orderings: stand-alone NTT in chunks that fit the Infinity Cache, and the
# relinearisation in groups of output limbs at batch 8 and 16.  usage: tools/gpu_chunk_ab.sh <tag>
set -o pipefail
OUT=gpurun_out/${1:-chunk}; mkdir -p $OUT
line() { python3 -c "
import json,sys; r=json.load(open('$1')); print('$2', round(r['value'],1), r['unit'], 'ms/step', round(r['ms_per_step'],3), {k: round(v['total_ms']/max(1,v['launches']),4) for k,v in r['kernel_ms'].items()})"; }
for ch in 0 64 128 256 512; do
  timeout -k 10 120 python3 -u bench.py --workload ntt --steps 20 --no-cpu-baseline --ntt-chunk $ch > $OUT/ntt_$ch.json 2> $OUT/ntt_$ch.err || { tail -5 $OUT/ntt_$ch.err; exit 1; }
  line $OUT/ntt_$ch.json "ntt chunk=$ch"
done
for b in 16 8; do for g in 0 1 2 4; do
  timeout -k 10 200 python3 -u bench.py --workload c3 --batch $b --steps 10 --no-cpu-baseline --relin-group $g > $OUT/c3_b${b}_g$g.json 2> $OUT/c3_b${b}_g$g.err || { tail -5 $OUT/c3_b${b}_g$g.err; exit 1; }
  line $OUT/c3_b${b}_g$g.json "c3 batch=$b group=$g"
done; done
