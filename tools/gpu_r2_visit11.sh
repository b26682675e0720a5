#!/bin/bash
# Round 2, visit 11: frequency-split form (XS) of the one-kernel convolution and the stem kernel: tests, A/B timings.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 1200 python -m pytest tests/test_heads_gpu.py -x -q -m gpu -k "stem or fused_winograd" 2>&1 | tail -8 > $O/r2v11_tests.log; cat $O/r2v11_tests.log
L=$O/r2v11_perf.log; : > $L
for xs in 1 0 1; do
  echo "== CSLAM_WFH_XS=$xs" >> $L
  CSLAM_WFH_XS=$xs timeout 300 python tools/perf_fused_h.py 256 5 2>&1 | grep "fp16 pairs\|diff" >> $L
  CSLAM_WFH_XS=$xs timeout 300 python tools/perf_stem.py 256 5 2>&1 | grep -v amdgpu >> $L
done
cat $L
echo visit11 done
