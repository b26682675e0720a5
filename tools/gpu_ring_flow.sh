#!/bin/bash
# flow control of the persistent candidate stage: time and L2 hit rate per "window,bias_q,bias_b" (measurement build)
# Usage: gpurun -- bash tools/gpu_ring_flow.sh <tag> "<w,bq,bb> ..." [variant: 1 = look every 4th stage, 4 = every stage]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r05_flow}; v=${3:-1}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R; export TMPDIR=/tmp
export CSLAM_HIP_LIB=$R/cslam_amd/libcslam_hip_abl.so
for w in ${2:-3,0,0}; do
  echo "== variant $v flow window,bias_q,bias_b $w" | tee -a $O/flow.log
  CSLAM_RING_FLOW_W=$w timeout 300 python tools/perf_match_ring.py 100000 0,$v 0 2 2>&1 | grep -E "^nq|rror" | tee -a $O/flow.log
  d=$O/tcc; CSLAM_RING_FLOW_W=$w timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $d -o t -- python tools/pmc_ring_target.py 100000 $v 1 > $d.log 2>&1
  python tools/pmc_ring_summary.py $d | grep -E "l2_hit_rate|TCC_MISS" | tee -a $O/flow.log; rm -rf $d
done
