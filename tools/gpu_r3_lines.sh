#!/bin/bash
# Round-3 bench lines from the current tree (run after tools/gpu_profile_round.sh + tools/stamp_profiles.py, so that the lines carry
# the counter blocks): tools/gpu_r3_lines.sh <tag>  ->  gpurun_out/<tag>/bench_*.json  (tools/collect_profiles.py <tag> r03 copies them)
set -o pipefail
TAG=${1:-r03lines}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 400 python -u bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err || { tail -5 $OUT/bench_c3.err; exit 1; }
echo "default line done"
for wl in ntt c2 c4 c5 rotm rotnn dec enc encode decode; do
  timeout -k 10 300 python -u bench.py --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { tail -5 $OUT/bench_$wl.err; exit 1; }
  echo "bench $wl done"
done
# the generic-prime table (what a caller's own primes run), no CPU leg
for wl in ntt c2 c3 c4 c5 rotm rotnn dec enc; do
  timeout -k 10 300 python -u bench.py --workload $wl --primes generic --no-cpu-baseline --no-ntt-n16 > $OUT/bench_${wl}_generic.json 2> $OUT/bench_${wl}_generic.err || { tail -5 $OUT/bench_${wl}_generic.err; exit 1; }
  echo "bench $wl generic done"
done
timeout -k 10 200 python -u bench.py --workload c2 --onepass 0 --no-cpu-baseline > $OUT/bench_c2_passpair.json 2> $OUT/bench_c2_passpair.err || exit 1
timeout -k 10 200 python -u bench.py --workload ntt --ntt-streams 1 --no-cpu-baseline > $OUT/bench_ntt_one_stream.json 2> $OUT/bench_ntt_one_stream.err || exit 1
timeout -k 10 300 python -u bench.py --workload c5 --e2e --no-cpu-baseline > $OUT/bench_c5_e2e.json 2> $OUT/bench_c5_e2e.err || exit 1
# A/B lines of the round's two kernel changes: vector-ALU base conversion, Montgomery-MAC relinearisation kernel
FHE_BENCH_BCONV=1 timeout -k 10 200 python -u bench.py --workload c4 --no-cpu-baseline > $OUT/bench_c4_valu_bconv.json 2> $OUT/bench_c4_valu_bconv.err || exit 1
FHE_BENCH_RELIN_KERNEL=0 timeout -k 10 200 python -u bench.py --workload c3 --no-cpu-baseline --no-ntt-n16 > $OUT/bench_c3_montgomery_mac.json 2> $OUT/bench_c3_montgomery_mac.err || exit 1
echo lines done
