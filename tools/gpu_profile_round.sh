#!/bin/bash
# Round profile: bench lines + rocprofv3 kernel stats + PMC traffic / VALU passes for the c3 (headline), ntt and c4 workloads.
# usage: tools/gpu_profile_round.sh <tag> [workloads...]    outputs under gpurun_out/<tag>/ (copy what is to be judged into profiles/<tag>/)
set -o pipefail
TAG=${1:-r03}; shift
WLS=${@:-c3 ntt c4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
for wl in $WLS; do
  # counter passes of the stand-alone NTT run its one-stream form (whole-call launches: per-launch figures are per call; rocprofv3
  # serialises dispatches under --pmc anyway); the kernel-trace pass runs the default two-stream form
  PMCX=""; [ $wl = ntt ] && PMCX="--ntt-streams 1"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -- python3 $B --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-ntt-n16 > $OUT/stats_$wl.log 2>&1 || { tail -20 $OUT/stats_$wl.log; exit 1; }
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$wl -- python3 $B --workload $wl $PMCX --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline --no-ntt-n16 > $OUT/fetch_$wl.log 2>&1 || { tail -5 $OUT/fetch_$wl.log; exit 1; }
  timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$wl -- python3 $B --workload $wl $PMCX --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline --no-ntt-n16 > $OUT/write_$wl.log 2>&1 || { tail -5 $OUT/write_$wl.log; exit 1; }
  timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/valu_$wl -- python3 $B --workload $wl $PMCX --steps 4 --warmup 1 --prewarm 0.5 --no-cpu-baseline --no-ntt-n16 > $OUT/valu_$wl.log 2>&1 || { tail -5 $OUT/valu_$wl.log; exit 1; }
done
for wl in $WLS; do
  mkdir -p $OUT/pmc_$wl && cp -r $OUT/fetch_$wl $OUT/write_$wl $OUT/pmc_$wl/ 2>/dev/null
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc_$wl > $OUT/pmc_summary_$wl.txt
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/valu_$wl > $OUT/pmc_valu_summary_$wl.txt
  rm -rf $OUT/pmc_$wl
  f=$(find $OUT/stats_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${wl}_kernel_stats.csv
  t=$(find $OUT/stats_$wl -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && [ $wl = ntt ] && python3 $GRAFT_REPO_ROOT/tools/ntt_call_span.py $t 16 > $OUT/ntt_call_span.txt
done
M=$GRAFT_REPO_ROOT/tools
for wl in $WLS; do
  case $wl in
    c3) python3 $M/make_traffic_json.py $OUT/pmc_summary_c3.txt c3 "c3, default batch per launch, N=2^16 Li=30" $OUT/traffic_c3.json relin_rows
        python3 $M/make_valu_json.py $OUT/valu_c3 c3 "c3, default batch per launch, N=2^16 Li=30" $OUT/valu_c3.json relin_rows ;;
    ntt) python3 $M/make_traffic_json.py $OUT/pmc_summary_ntt.txt ntt "ntt, 2048 limbs/launch, N=2^16" $OUT/traffic_ntt.json "ntt_cols_kernel<true, false" "ntt_rows8_fwd_kernel<true, 0"
        python3 $M/make_valu_json.py $OUT/valu_ntt ntt "ntt, 2048 limbs/launch, N=2^16" $OUT/valu_ntt.json "ntt_cols_kernel<true, false" "ntt_rows8_fwd_kernel<true, 0" ;;
    c2) python3 $M/make_traffic_json.py $OUT/pmc_summary_c2.txt c2 "c2, 8192 limbs/launch, N=2^14 (one-pass forward kernel)" $OUT/traffic_c2.json "ntt_onepass_fwd_kernel<true"
        python3 $M/make_valu_json.py $OUT/valu_c2 c2 "c2, 8192 limbs/launch, N=2^14 (one-pass forward kernel)" $OUT/valu_c2.json "ntt_onepass_fwd_kernel<true" ;;
    c4) python3 $M/make_traffic_json.py $OUT/pmc_summary_c4.txt c4 "c4, default batch per launch, N=2^16 Li=K=30 (last forward row pass: key product + mod-down tail, both halves)" $OUT/traffic_c4.json "ntt_rows8_fwd_kernel<true, 4"
        python3 $M/make_valu_json.py $OUT/valu_c4 c4 "c4, default batch per launch, N=2^16 Li=K=30 (last forward row pass: key product + mod-down tail, both halves)" $OUT/valu_c4.json "ntt_rows8_fwd_kernel<true, 4" ;;
  esac
done
# raw counter dumps are large: keep the summaries and the JSONs
rm -rf $OUT/fetch_* $OUT/write_* $OUT/valu_c3 $OUT/valu_ntt $OUT/valu_c4 $OUT/valu_c2 $OUT/stats_*/
ls $OUT
