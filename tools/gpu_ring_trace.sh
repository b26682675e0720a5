#!/bin/bash
# placement (HW_REG_XCC_ID), tile-start spread of every XCD's patch and flow-control pauses of the persistent candidate stage; with
# CSLAM_MFMA_DBG=33 also the per-wave barrier stamps of workgroup 0 (measurement build).  Usage: gpurun -- bash tools/gpu_ring_trace.sh <tag> [variants] [nq]
cd ${GRAFT_REPO_ROOT:-/root/repo}; export CSLAM_HIP_LIB=$PWD/cslam_amd/libcslam_hip_abl.so; tag=${1:-r05_trace}; mkdir -p gpurun_out/$tag
for v in ${2:-0}; do echo "== variant $v dbg ${CSLAM_MFMA_DBG:-0} nq ${3:-100000}"; CSLAM_RING_XCC=1 timeout 300 python tools/pmc_ring_target.py ${3:-100000} $v 1 2>&1 | grep -E "ring|^[0-9]"; done 2>&1 | tee -a gpurun_out/$tag/trace.log
