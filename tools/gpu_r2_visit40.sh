#!/bin/bash
O=gpurun_out; mkdir -p $O; L=$O/r2v40_solve4_ab.log; : > $L
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
for v in "" "-DCS4_ROWS_UNROLL=4" "-DCS4_ROWS_GRID=512" "-DCS4_ROWS_UNROLL=4 -DCS4_ROWS_GRID=512" "-DCS4_ROWS_UNROLL=4 -DCS4_ROWS_GRID=1024"; do
  echo "== flags: $v" | tee -a $L
  touch cslam_amd/csrc/mac_kernels.hip; make -C cslam_amd/csrc CXXFLAGS="$base $v" > /dev/null 2>&1
  timeout 300 python tools/perf_solve4.py 2>&1 | grep -v amdgpu | tee -a $L
done
touch cslam_amd/csrc/mac_kernels.hip; make -C cslam_amd/csrc > /dev/null 2>&1
echo "== look-ahead with a high-priority side stream (on / off)" | tee -a $L
for v in "A=1" "CSLAM_FIEDLER_LOOKAHEAD=0"; do
  env $v CSLAM_MAC_TIMING=1 timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_hip 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300 | tee -a $L
done
