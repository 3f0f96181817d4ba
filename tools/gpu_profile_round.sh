#!/bin/bash
# Round profile: bench lines + rocprofv3 kernel stats + PMC traffic for the c3 (headline) and ntt workloads.
# usage: tools/gpu_profile_round.sh <tag>     outputs under gpurun_out/<tag>/
set -o pipefail
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err || { tail -20 $OUT/bench_c3.err; exit 1; }
timeout -k 10 200 python -u bench.py --workload ntt --steps 20 --warmup 3 > $OUT/bench_ntt.json 2> $OUT/bench_ntt.err || { tail -20 $OUT/bench_ntt.err; exit 1; }
for wl in c4 c5 c2; do
  timeout -k 10 300 python -u bench.py --workload $wl --steps 5 --warmup 1 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { tail -20 $OUT/bench_$wl.err; exit 1; }
done
cd /tmp && export TMPDIR=/tmp
for wl in c3 ntt c4; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats_$wl.log 2>&1 || { tail -20 $OUT/stats_$wl.log; exit 1; }
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline > $OUT/fetch_$wl.log 2>&1 || { tail -5 $OUT/fetch_$wl.log; exit 1; }
  timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline > $OUT/write_$wl.log 2>&1 || { tail -5 $OUT/write_$wl.log; exit 1; }
done
for wl in c3 ntt c4; do
  mkdir -p $OUT/pmc_$wl && cp -r $OUT/fetch_$wl $OUT/write_$wl $OUT/pmc_$wl/ 2>/dev/null
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc_$wl > $OUT/pmc_summary_$wl.txt
  rm -rf $OUT/pmc_$wl
done
python3 $GRAFT_REPO_ROOT/tools/make_traffic_json.py $OUT/pmc_summary_c3.txt c3 "c3, default batch per launch, N=2^16 Li=30" $OUT/traffic_c3.json relin_rows
python3 $GRAFT_REPO_ROOT/tools/make_traffic_json.py $OUT/pmc_summary_ntt.txt ntt "ntt, 2048 limbs/launch, N=2^16" $OUT/traffic_ntt.json "ntt_cols_kernel<true, false" "ntt_rows8_fwd_kernel<true, 0"
python3 $GRAFT_REPO_ROOT/tools/make_traffic_json.py $OUT/pmc_summary_c4.txt c4 "c4, default batch per launch, N=2^16 Li=K=30 (one base-conversion launch)" $OUT/traffic_c4.json bconv_kernel
cat $OUT/pmc_summary_c3.txt | head -40
