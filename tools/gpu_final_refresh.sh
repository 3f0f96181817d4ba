#!/bin/bash
# Everything profiles/r01 holds, from the current tree: tools/gpu_final_refresh.sh <tag>
set -o pipefail
TAG=${1:-final}
bash tools/gpu_profile_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1 || { tail -20 gpurun_out/${TAG}_round.log; exit 1; }
for wl in rotm rotnn dec enc; do
  timeout -k 10 250 python -u bench.py --workload $wl > gpurun_out/$TAG/bench_$wl.json 2> gpurun_out/$TAG/bench_$wl.err || { tail -5 gpurun_out/$TAG/bench_$wl.err; exit 1; }
done
bash tools/gpu_pmc.sh c3 ${TAG}_pmc_c3 > gpurun_out/${TAG}_pmc_c3.log 2>&1 || { tail -5 gpurun_out/${TAG}_pmc_c3.log; exit 1; }
bash tools/gpu_pmc.sh ntt ${TAG}_pmc_ntt > gpurun_out/${TAG}_pmc_ntt.log 2>&1 || { tail -5 gpurun_out/${TAG}_pmc_ntt.log; exit 1; }
echo refresh done
