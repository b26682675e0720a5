#!/bin/bash
# Round 2, visit 16: A/B of the transform arithmetic (shared-term B^T d, SDWA pack) on both one-kernel layers; MAC set-up split.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
L=$O/r2v16_ab.log; : > $L
for fl in "" "-DWH_BT_OLD" "-DWH_PACK_OLD" "-DWH_BT_OLD -DWH_PACK_OLD" ""; do
  echo "== flags: [$fl]" >> $L
  (cd cslam_amd/csrc && rm -f wino_fused_h.o && make CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $fl" >/dev/null 2>&1)
  timeout 300 python tools/perf_fused_h.py 256 5 2>&1 | grep "fp16 pairs" >> $L
  timeout 300 python tools/perf_stem.py 256 5 2>&1 | grep "stem kernel" >> $L
done
cat $L
CSLAM_MAC_TIMING=2 timeout 600 python tools/perf_mac.py 125000 3000 2>&1 | grep -v amdgpu | tail -4 | cut -c1-400 > $O/r2v16_mac_setup.log; cat $O/r2v16_mac_setup.log
echo visit16 done
