#!/bin/bash
# SQ counter pass over timing-only ablations of one variant.  Usage: gpurun -- bash tools/gpu_ring_pmc_dbg.sh <tag> <variant> "<dbgs>"
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r05_pmc_dbg}; v=${2:-0}; dbgs=${3:-"0 8 10 9 13"}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R; export TMPDIR=/tmp
export CSLAM_HIP_LIB=$R/cslam_amd/libcslam_hip_abl.so
for dbg in $dbgs; do
  export CSLAM_MFMA_DBG=$dbg
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq$dbg -o s -- python tools/pmc_ring_target.py 100000 $v 2 > $O/sq$dbg.log 2>&1
  echo "== dbg $dbg"; python tools/pmc_ring_summary.py $O/sq$dbg | grep -E "frac|GRBM|SQ_WAIT_INST_LDS\"|SQ_WAVE_CYCLES\"|sim_topk"
  grep -E "^-?[0-9]+ \(" $O/sq$dbg.log
  rm -rf $O/sq$dbg
done 2>&1 | tee $O/summary.log
