#!/bin/bash
# Round 2, visit 1: the new C5-size test first (fast feedback), then the whole GPU suite, smoke, the default bench line
# (measured peaks, CPU extract leg, fp32-GEMM whole step), bench.py --gpus 2 launching its own two ranks on this 1-GPU
# box (--debug-shared-gpu), and the C5 rehearsal with the vectorised bookkeeping.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 1200 python -m pytest tests/test_c5_gpu.py -x -q -m gpu 2>&1 | tail -40 > $O/r2v1_c5_test.log; tail -5 $O/r2v1_c5_test.log
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_c5_gpu.py 2>&1 | tail -25 > $O/r2v1_tests_gpu.log; tail -5 $O/r2v1_tests_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 > $O/r2v1_smoke.log; cat $O/r2v1_smoke.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/r2v1_bench.json 2> $O/r2v1_bench.err; cat $O/r2v1_bench.json; tail -3 $O/r2v1_bench.err
timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --debug-shared-gpu --no-cpu-baseline > $O/r2v1_bench_2rank.json 2> $O/r2v1_bench_2rank.err; cat $O/r2v1_bench_2rank.json; tail -3 $O/r2v1_bench_2rank.err
timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --debug-shared-gpu --no-cpu-baseline --shard-mode robots > $O/r2v1_bench_2rank_robots.json 2>> $O/r2v1_bench_2rank.err; cat $O/r2v1_bench_2rank_robots.json
timeout 900 python bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline > $O/r2v1_bench_2gpu_refused.log 2>&1; echo "rc of --gpus 2 without a second GPU: $?" >> $O/r2v1_bench_2gpu_refused.log; tail -3 $O/r2v1_bench_2gpu_refused.log
timeout 1200 python tools/perf_c5.py 2>&1 | grep -v amdgpu > $O/r2v1_perf_c5.log; cat $O/r2v1_perf_c5.log
echo visit1 done
