#!/bin/bash
# Builds csrc/wino_fused.hip alone with phases left out (-DWF_ABL=mask: 1 transform, 2 all but one frequency's MFMAs,
# 4 output stores, 8 patch loads of quarters 1..3, 16 weight loads of quarters 1..3;
# persistent form: 32 producer waves idle, 64 all but one frequency's MFMAs, 4 output stores;
# F(4x4) persistent form: 32 / 64 likewise, 1024 no transform arithmetic, 2048 no patch loads after the first two) -> tools/_abl/libwf_<mask>.so;
# tools/exp_fused_ablation.py times them on the GPU box.  Results are wrong by construction: timing only.
cd $(dirname $0)/..
for m in 0 32 64 96 128 160 1024 2048; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DWF_ABL=$m cslam_amd/csrc/wino_fused.hip tools/_abl/stub.cpp -o tools/_abl/libwf_$m.so &
done; wait; ls tools/_abl
