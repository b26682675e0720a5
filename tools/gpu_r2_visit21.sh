#!/bin/bash
# Round 2, visit 21: conv2_1 A/B: transform split over both thread halves, LDS-staged output stores.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
L=$O/r2v21_ab.log; : > $L
for fl in "" "-DWH_NO_SPLIT_T" "-DWH_STAGE_STORES" "-DWH_NO_SPLIT_T -DWH_STAGE_STORES" ""; do
  echo "== flags: [$fl]" >> $L
  (cd cslam_amd/csrc && rm -f wino_fused_h.o && make CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $fl" >/dev/null 2>&1)
  timeout 300 python tools/perf_fused_h.py 256 5 2>&1 | grep "conv2_1 fp16 pairs" >> $L
done
cat $L
echo visit21 done
