#!/bin/bash
# A/B of the persistent candidate stage against the one-workgroup-per-item kernel (measurement build), plus the matcher's GPU tests
# on the product build.  Usage: gpurun -- bash tools/gpu_ring_ab.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r05_ring}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_nns_gpu.py tests/test_nns_random_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
export CSLAM_HIP_LIB=$R/cslam_amd/libcslam_hip_abl.so
timeout 600 python tools/perf_match_ring.py 100000,16384,1024 -1,0,1,2,3 0 3 > $O/ab_dbg0.log 2>&1; cat $O/ab_dbg0.log
timeout 600 python tools/perf_match_ring.py 100000,1024 -1,1 1 2 > $O/ab_dbg1.log 2>&1; cat $O/ab_dbg1.log
timeout 600 python tools/perf_match_ring.py 100000,1024 -1,1 2 2 > $O/ab_dbg2.log 2>&1; cat $O/ab_dbg2.log
