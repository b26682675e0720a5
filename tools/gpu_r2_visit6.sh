#!/bin/bash
# Round 2, visit 6: fp16-pair one-kernel convolution after the loader re-ordering (parity, timing, ring depths, ablations).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests/test_heads_gpu.py -x -q -m gpu -k "fused_winograd" 2>&1 | tail -8 > $O/r2v6_fused_tests.log; tail -4 $O/r2v6_fused_tests.log
: > $O/r2v6_fused_h.log
for br in 2 4; do echo "== ring of $br pairs" >> $O/r2v6_fused_h.log; CSLAM_WFH_BR=$br timeout 300 python tools/perf_fused_h.py 256 3 2>&1 | grep -v amdgpu >> $O/r2v6_fused_h.log; done
for d in 1 4 5 8 16; do echo "== CSLAM_WFH_DBG=$d" >> $O/r2v6_fused_h.log; CSLAM_WFH_DBG=$d timeout 300 python tools/perf_fused_h.py 256 3 2>&1 | grep "fp16 pairs" >> $O/r2v6_fused_h.log; done
cat $O/r2v6_fused_h.log
echo visit6 done
