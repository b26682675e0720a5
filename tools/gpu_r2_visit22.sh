#!/bin/bash
# Round 2, visit 22: state check after the reverted experiments (fused / stem / GEMM tests), Cholesky panel solve, C5 rehearsal
# at chunk 250 and 1000, MAC at 1e6 poses.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 1800 python -m pytest tests/test_heads_gpu.py tests/test_wino_gemm_gpu.py tests/test_mac_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/r2v22_tests.log; cat $O/r2v22_tests.log
timeout 600 python tools/perf_chol.py 2>&1 | grep -v amdgpu | grep "m=32768\|m=16384" > $O/r2v22_perf_chol.log; cat $O/r2v22_perf_chol.log
timeout 600 python tools/extract_leg.py --iters 4 2>&1 | grep -v amdgpu | tail -2 > $O/r2v22_extract.log; cat $O/r2v22_extract.log
CSLAM_MAC_TIMING=1 timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_gpu 2>&1 | grep -v amdgpu | grep "per FW\|select" > $O/r2v22_perf_acm.log; cut -c1-300 $O/r2v22_perf_acm.log
for ch in 250 1000; do timeout 2400 python tools/perf_c5.py 12500 8 1000 $ch 2>&1 | grep -v amdgpu | tail -5 >> $O/r2v22_perf_c5.log; done; cut -c1-400 $O/r2v22_perf_c5.log
echo visit22 done
