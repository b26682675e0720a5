#!/bin/bash
# Round 2, visit 15: stem kernel with the cheaper transform arithmetic, MAC with parallel finish / carry kernels; tests.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 300 python tools/prof_fused_phases.py 256 2>&1 | grep -v amdgpu > $O/r2v15_phases.log; cat $O/r2v15_phases.log
L=$O/r2v15_perf.log; : > $L
timeout 300 python tools/perf_fused_h.py 256 5 2>&1 | grep "fp16 pairs\|diff" >> $L
timeout 300 python tools/perf_stem.py 256 5 2>&1 | grep -v amdgpu >> $L
cat $L
timeout 1800 python -m pytest tests/test_heads_gpu.py tests/test_mac_gpu.py tests/test_c5_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/r2v15_tests.log; cat $O/r2v15_tests.log
CSLAM_MAC_TIMING=1 timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_gpu 2>&1 | grep -v amdgpu | grep "fiedler:\|per FW\|select" > $O/r2v15_perf_acm.log; cut -c1-300 $O/r2v15_perf_acm.log
echo visit15 done
