#!/bin/bash
O=gpurun_out; mkdir -p $O; L=$O/r2v52_solve4_ab.log; : > $L
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
for v in "" "-DCS4_THREADS=512" "-DCS4_THREADS=512 -DCS4_ROWS_GRID=256" "-DCS4_RPW=8" "-DCS4_THREADS=512 -DCS4_RPW=8 -DCS4_ROWS_GRID=256" "-DCS4_THREADS=1024 -DCS4_ROWS_GRID=256" "-DCS4_ROWS_UNROLL=8"; do
  echo "== flags: $v" | tee -a $L
  touch cslam_amd/csrc/mac_kernels.hip; make -C cslam_amd/csrc CXXFLAGS="$base $v" > /dev/null 2>&1 || echo "build failed" | tee -a $L
  timeout 300 python tools/perf_solve4.py 2>&1 | grep -v amdgpu | grep 32768 | tee -a $L
done
