#!/bin/bash
# PMC passes over the c4 workload (base conversion kernels): tools/gpu_pmc_c4.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 $B --workload c4 --steps 3 --warmup 1 --prewarm 0.3 --no-cpu-baseline --no-ntt-n16 > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/p$i > $OUT/pmc_$i.txt
  rm -rf $OUT/p$i
  grep -i "bconv" $OUT/pmc_$i.txt
done
