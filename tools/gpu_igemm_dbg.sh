#!/bin/bash
# timing-only ablations of the pair-format implicit-GEMM loop (measurement build: CSLAM_CI_DBG bits, csrc/conv_igemm.hip):
# bash tools/gpu_igemm_dbg.sh <tag> "<dbg values>"
tag=${1:-igemm_dbg}; out=gpurun_out/$tag; mkdir -p $out
export CSLAM_HIP_LIB=cslam_amd/libcslam_hip_abl.so
for d in ${2:-0 1 2 3 4 8 16 48 64}; do
  echo "== CSLAM_CI_DBG=$d" | tee -a $out/dbg.log
  CSLAM_CI_DBG=$d python tools/perf_conv_igemm.py 1000 2>&1 | grep -E "layer1 |layer2 3x3|layer3 3x3|layer4 3x3" | sed 's/.*pair format in \/ out://' | tee -a $out/dbg.log
done
