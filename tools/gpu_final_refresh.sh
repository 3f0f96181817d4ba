#!/bin/bash
# Everything profiles/<round> holds, from the current tree: tools/gpu_final_refresh.sh <tag>   (then tools/collect_profiles.py <tag> <round>)
# 1. rocprofv3 kernel stats + FETCH / WRITE / VALU counter passes for c3, ntt, c4, c2 (tools/gpu_profile_round.sh)
# 2. the driver's default line (c3 + ntt_n16 + cpu_baseline) and one bench line per other workload
set -o pipefail
TAG=${1:-final}
mkdir -p gpurun_out/$TAG
bash tools/gpu_profile_round.sh $TAG c3 ntt c4 c2 > gpurun_out/${TAG}_round.log 2>&1 || { tail -20 gpurun_out/${TAG}_round.log; exit 1; }
echo "profile passes done"
timeout -k 10 400 python -u bench.py > gpurun_out/$TAG/bench_c3.json 2> gpurun_out/$TAG/bench_c3.err || { tail -5 gpurun_out/$TAG/bench_c3.err; exit 1; }
echo "default line done"
for wl in ntt c2 c4 c5 rotm rotnn dec enc; do
  timeout -k 10 300 python -u bench.py --workload $wl > gpurun_out/$TAG/bench_$wl.json 2> gpurun_out/$TAG/bench_$wl.err || { tail -5 gpurun_out/$TAG/bench_$wl.err; exit 1; }
  echo "bench $wl done"
done
timeout -k 10 200 python -u bench.py --workload c2 --onepass 0 --no-cpu-baseline > gpurun_out/$TAG/bench_c2_passpair.json 2> gpurun_out/$TAG/bench_c2_passpair.err || exit 1
timeout -k 10 300 python -u bench.py --workload c5 --e2e --no-cpu-baseline > gpurun_out/$TAG/bench_c5_e2e.json 2> gpurun_out/$TAG/bench_c5_e2e.err || exit 1
echo refresh done
