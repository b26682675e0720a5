#!/bin/bash
# Round 2, visit 3: tile / ring shapes of the pair GEMM (parity of every shape, per-layer timing of every shape), the
# multi-bank search (parity, C5-size test, C5 rehearsal), bench line with the new peak micro-benchmarks.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 1200 python -m pytest tests/test_wino_gemm_gpu.py -x -q -m gpu 2>&1 | tail -30 > $O/r2v3_gemm_tests.log; tail -4 $O/r2v3_gemm_tests.log
timeout 600 python tools/perf_wino_gemm.py 2>&1 | grep -v amdgpu > $O/r2v3_perf_wino_gemm.log; cat $O/r2v3_perf_wino_gemm.log
timeout 900 python -m pytest tests/test_nns_gpu.py tests/test_lcsm_gpu.py tests/test_full_loop_gpu.py tests/test_c5_gpu.py -x -q -m gpu 2>&1 | tail -30 > $O/r2v3_match_tests.log; tail -4 $O/r2v3_match_tests.log
timeout 1200 python tools/perf_c5.py 2>&1 | grep -v amdgpu > $O/r2v3_perf_c5.log; cat $O/r2v3_perf_c5.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/r2v3_bench.json 2> $O/r2v3_bench.err; cat $O/r2v3_bench.json; tail -3 $O/r2v3_bench.err
echo visit3 done
