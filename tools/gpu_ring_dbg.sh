#!/bin/bash
# timing-only ablations of the candidate stage (measurement build).  Usage: gpurun -- bash tools/gpu_ring_dbg.sh <tag> <variants> "<dbgs>" [nqs]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r05_dbg}; vs=${2:--1,0}; dbgs=${3:-"1 2"}; nqs=${4:-100000,1024}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
export CSLAM_HIP_LIB=$R/cslam_amd/libcslam_hip_abl.so
for dbg in $dbgs; do timeout 600 python tools/perf_match_ring.py $nqs $vs $dbg 2 2>&1 | grep -E "^nq|rror" | tee -a $O/dbg.log; done
