#!/bin/bash
# cache policy of the candidate stage's LDS-DMA requests (measurement build): time and L2 hit rate per policy code
# (DBG = bank << 6 | query << 9; 1 = sc0, 2 = nt, 4 = sc1).  Usage: gpurun -- bash tools/gpu_ring_policy.sh <tag> "<dbg codes>"
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r05_policy}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R; export TMPDIR=/tmp
export CSLAM_HIP_LIB=$R/cslam_amd/libcslam_hip_abl.so
for dbg in ${2:-0 128 1024 1152 256 2048 64 512}; do
  echo "== policy code $dbg" | tee -a $O/policy.log
  timeout 300 python tools/perf_match_ring.py 100000,1024 0 $dbg 2 2>&1 | grep -E "^nq|rror" | tee -a $O/policy.log
  d=$O/tcc; CSLAM_MFMA_DBG=$dbg timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $d -o t -- python tools/pmc_ring_target.py 100000 0 1 > $d.log 2>&1
  python tools/pmc_ring_summary.py $d | grep -E "l2_hit_rate|TCC_MISS" | tee -a $O/policy.log; rm -rf $d
done
