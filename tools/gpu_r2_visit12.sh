#!/bin/bash
# Round 2, visit 12: per-phase cycle counts of the one-kernel convolution forms; the reworked junction solve.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 300 python tools/prof_fused_phases.py 256 2>&1 | grep -v amdgpu > $O/r2v12_phases.log; cat $O/r2v12_phases.log
timeout 900 python -m pytest tests/test_mac_gpu.py tests/test_heads_gpu.py -x -q -m gpu -k "stem or cholesky or chain" 2>&1 | tail -5 > $O/r2v12_tests.log; cat $O/r2v12_tests.log
timeout 600 python tools/perf_chol.py 2>&1 | grep -v amdgpu | tail -3 > $O/r2v12_perf_chol.log; cat $O/r2v12_perf_chol.log
CSLAM_MAC_TIMING=1 timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_gpu 2>&1 | grep -v amdgpu | grep "fiedler:\|per FW\|select" > $O/r2v12_perf_acm.log; cut -c1-300 $O/r2v12_perf_acm.log
echo visit12 done
