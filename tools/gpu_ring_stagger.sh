#!/bin/bash
# Does the XCD's L2 merge simultaneous misses?  Start-staggered patches (measurement build) against lock-step ones: time and L2 hit rate.
# Usage: gpurun -- bash tools/gpu_ring_stagger.sh <tag> "a,b,cycles a,b,cycles ..."
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r05_stag}; cfgs=${2:-"0,0,0 2,0,2000 0,1,2000 2,1,2000 1,1,3000"}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R; export TMPDIR=/tmp
export CSLAM_HIP_LIB=$R/cslam_amd/libcslam_hip_abl.so
CSLAM_RING_XCC=1 CSLAM_MFMA_RING=0 timeout 300 python tools/pmc_ring_target.py 100000 0 1 2>&1 | grep -E "ring|^0" | tee $O/xcc.log
for c in $cfgs; do
  export CSLAM_RING_STAGGER=$c
  echo "== stagger $c" | tee -a $O/stagger.log
  timeout 300 python tools/perf_match_ring.py 100000 0 0 2 2>&1 | grep "^nq" | tee -a $O/stagger.log
  d=$O/tcc_$(echo $c | tr ',' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $d -o t -- python tools/pmc_ring_target.py 100000 0 1 > $d.log 2>&1
  python tools/pmc_ring_summary.py $d | grep -E "l2_hit_rate|TCC_MISS" | tee -a $O/stagger.log
  rm -rf $d
done
