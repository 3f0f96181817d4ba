#!/bin/bash
# PMC passes over one bench workload: tools/gpu_pmc.sh <workload> <tag>
set -o pipefail
WL=${1:-ntt}; TAG=${2:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU && \
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM && \
run fetch FETCH_SIZE GRBM_GUI_ACTIVE && \
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum && \
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT
