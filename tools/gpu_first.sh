#!/bin/bash
# first GPU pass: smoke, bench lines, rocprof kernel stats
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 120 python -u bench.py --workload ntt --steps 20 --warmup 3 > gpurun_out/bench_ntt.json 2> gpurun_out/bench_ntt.err || { tail -20 gpurun_out/bench_ntt.err; exit 1; }
cat gpurun_out/bench_ntt.json
timeout -k 10 300 python -u bench.py --workload c3 --steps 5 --warmup 1 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err || { tail -20 gpurun_out/bench_c3.err; exit 1; }
cat gpurun_out/bench_c3.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_c3.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/prof_c3.log; exit 1; }
find $GRAFT_REPO_ROOT/gpurun_out/prof_c3 -name "*kernel_stats*" | head
