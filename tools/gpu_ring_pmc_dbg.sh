#!/bin/bash
# SQ counter pass over the timing-only ablations (no loads / every request an L2 hit) and the product form of one variant
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r05_pmc_dbg}; v=${2:-0}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R; export TMPDIR=/tmp
export CSLAM_HIP_LIB=$R/cslam_amd/libcslam_hip_abl.so
for dbg in 0 1 2; do
  export CSLAM_MFMA_DBG=$dbg
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq$dbg -o s -- python tools/pmc_ring_target.py 100000 $v 2 > $O/sq$dbg.log 2>&1
  echo "== dbg $dbg"; python tools/pmc_ring_summary.py $O/sq$dbg | grep -E "frac|GRBM|launches|sim_topk"
  grep -E "^-?[0-9]+ \(" $O/sq$dbg.log
  rm -rf $O/sq$dbg
done 2>&1 | tee $O/summary.log
