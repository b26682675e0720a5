#!/bin/bash
# Round 2, visit 4: the fp16-pair form of the one-kernel convolution (parity, timing against the f32 form), GEMM shapes with
# interleaved timing, trunk-level timing and the affected GPU tests.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests/test_heads_gpu.py -x -q -m gpu -k "fused_winograd" 2>&1 | tail -30 > $O/r2v4_fused_tests.log; tail -6 $O/r2v4_fused_tests.log
timeout 600 python tools/perf_fused_h.py 2>&1 | grep -v amdgpu > $O/r2v4_perf_fused_h.log; cat $O/r2v4_perf_fused_h.log
timeout 600 python tools/perf_wino_gemm.py 2>&1 | grep -v amdgpu > $O/r2v4_perf_wino_gemm.log; cat $O/r2v4_perf_wino_gemm.log
timeout 600 python tools/extract_leg.py --iters 4 2>&1 | grep -v amdgpu | tail -3 > $O/r2v4_extract_leg.log; cat $O/r2v4_extract_leg.log
CSLAM_WINO_FUSED_H=0 timeout 600 python tools/extract_leg.py --iters 4 2>&1 | grep -v amdgpu | tail -3 >> $O/r2v4_extract_leg.log; tail -1 $O/r2v4_extract_leg.log
timeout 1800 python -m pytest tests/test_heads_gpu.py tests/test_full_loop_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -30 > $O/r2v4_heads_tests.log; tail -6 $O/r2v4_heads_tests.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 > $O/r2v4_smoke.log; cat $O/r2v4_smoke.log
echo visit4 done
