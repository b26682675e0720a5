#!/bin/bash
# Round 2, visit 2: the pair GEMM -- parity tests, per-layer timing against the library forms (with the two timing-only
# ablations), the rest of the GPU suite that visit 1 did not reach, trunk-level timing.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests/test_wino_gemm_gpu.py -x -q -m gpu 2>&1 | tail -30 > $O/r2v2_gemm_tests.log; tail -6 $O/r2v2_gemm_tests.log
timeout 900 python -m pytest tests/test_heads_gpu.py -x -q -m gpu -k "split16" 2>&1 | tail -30 > $O/r2v2_split16_tests.log; tail -6 $O/r2v2_split16_tests.log
timeout 600 python tools/perf_wino_gemm.py 2>&1 | grep -v amdgpu > $O/r2v2_perf_wino_gemm.log; cat $O/r2v2_perf_wino_gemm.log
CSLAM_WGEMM_DBG=1 timeout 600 python tools/perf_wino_gemm.py 2>&1 | grep -v amdgpu > $O/r2v2_perf_wino_gemm_dbg1.log; cat $O/r2v2_perf_wino_gemm_dbg1.log
CSLAM_WGEMM_DBG=2 timeout 600 python tools/perf_wino_gemm.py 2>&1 | grep -v amdgpu > $O/r2v2_perf_wino_gemm_dbg2.log; cat $O/r2v2_perf_wino_gemm_dbg2.log
timeout 600 python tools/extract_leg.py --iters 4 2>&1 | grep -v amdgpu | tail -5 > $O/r2v2_extract_leg.log; cat $O/r2v2_extract_leg.log
CSLAM_WINO_H3=1 timeout 600 python tools/extract_leg.py --iters 4 2>&1 | grep -v amdgpu | tail -5 > $O/r2v2_extract_leg_h3.log; cat $O/r2v2_extract_leg_h3.log
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_c5_gpu.py --deselect tests/test_wino_gemm_gpu.py 2>&1 | tail -25 > $O/r2v2_tests_gpu.log; tail -5 $O/r2v2_tests_gpu.log
echo visit2 done
