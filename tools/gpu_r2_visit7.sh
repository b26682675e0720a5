#!/bin/bash
# Round 2, visit 7: the whole GPU suite, smoke, the bench line, then the rocprofv3 evidence (tools/gpu_r2_pmc.sh).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/r2v7_tests_gpu.log; tail -5 $O/r2v7_tests_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 > $O/r2v7_smoke.log; cat $O/r2v7_smoke.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/r2v7_bench.json 2> $O/r2v7_bench.err; cat $O/r2v7_bench.json; tail -3 $O/r2v7_bench.err
bash tools/gpu_r2_pmc.sh
echo visit7 done
