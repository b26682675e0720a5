#!/bin/bash
# Round 2, visit 26: stem kernel with the producer's row tiles interleaved into the matrix loop, A/B + tests.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
L=$O/r2v26_ab.log; : > $L
for fl in "" "-DWH_NO_STEM_ILV" ""; do
  echo "== flags: [$fl]" >> $L
  (cd cslam_amd/csrc && rm -f wino_fused_h.o && make CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $fl" >/dev/null 2>&1)
  timeout 300 python tools/perf_stem.py 256 5 2>&1 | grep "stem" >> $L
  timeout 300 python tools/prof_fused_phases.py 256 2>&1 | grep "stem=1" >> $L
done
cat $L
timeout 600 python -m pytest tests/test_heads_gpu.py -x -q -m gpu -k "stem" 2>&1 | tail -3
echo visit26 done
