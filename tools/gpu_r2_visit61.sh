#!/bin/bash
# Round 2, visit 61: whole GPU suite, smoke, bench line, MAC at 1e6 poses at the end of the round (native Frank-Wolfe loop, device 4x4 algebra)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/r2v61_tests_gpu.log; tail -5 $O/r2v61_tests_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 > $O/r2v61_smoke.log; cat $O/r2v61_smoke.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/r2v61_bench.json 2> $O/r2v61_bench.err; cat $O/r2v61_bench.json | cut -c1-300; tail -3 $O/r2v61_bench.err
CSLAM_MAC_TIMING=1 timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_hip 2>&1 | grep -v amdgpu | grep "per FW\|select" | cut -c1-400 > $O/r2v61_perf_acm_1M.log; cat $O/r2v61_perf_acm_1M.log
echo visit61 done
