#!/bin/bash
# shader clock and VALU occupancy of the relinearisation kernels with and without output-limb grouping (c3, batch 8):
# tools/gpu_group_sclk.sh <outdir>   (one rocprofv3 --pmc pass per setting; GRBM_GUI_ACTIVE / 8 / duration = sclk)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-group_sclk}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for g in 0 1 2; do
  timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/g$g -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --batch 8 --relin-group $g --steps 4 --warmup 1 --prewarm 0.5 --no-cpu-baseline > $OUT/g$g.log 2>&1 || { tail -5 $OUT/g$g.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/make_valu_json.py $OUT/g$g c3 "c3 batch 8 relin-group $g: relin_rows8p_mac_kernel" $OUT/rows_g$g.json relin_rows8p > /dev/null
  python3 $GRAFT_REPO_ROOT/tools/make_valu_json.py $OUT/g$g c3 "c3 batch 8 relin-group $g: relinearisation column pass" $OUT/cols_g$g.json "ntt_cols_kernel<true, false" > /dev/null
  python3 - <<P
import json
for k in ("rows","cols"):
    j=json.load(open("$OUT/%s_g$g.json"%k)); sclk=j["gui_active_per_xcd_per_launch"]/(j["launch_ms_in_pmc_pass"]*1e3)
    print("group=$g %-4s launch %.3f ms  sclk %.0f MHz  VALU issue %.1f %%  (%s)"%(k, j["launch_ms_in_pmc_pass"], sclk, 100*j["insts_valu_per_launch"]*4/1024/j["gui_active_per_xcd_per_launch"], j["kernel"][:60]))
P
  rm -rf $OUT/g$g
done
