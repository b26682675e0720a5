#!/bin/bash
# Round 2, visit 30: whole GPU suite, smoke, bench line, rocprofv3 evidence (stats + PMC), phases incl. conv2_1, MAC at 1e6 poses.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 300 python tools/prof_fused_phases.py 256 2>&1 | grep -v amdgpu > $O/r2v30_phases.log; cat $O/r2v30_phases.log
CSLAM_MAC_TIMING=1 timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_gpu 2>&1 | grep -v amdgpu | grep "fiedler:\|per FW\|select" > $O/r2v30_perf_acm.log; tail -4 $O/r2v30_perf_acm.log | cut -c1-300
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/r2v30_tests_gpu.log; tail -5 $O/r2v30_tests_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 > $O/r2v30_smoke.log; cat $O/r2v30_smoke.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/r2v30_bench.json 2> $O/r2v30_bench.err; cat $O/r2v30_bench.json; tail -3 $O/r2v30_bench.err
bash tools/gpu_r2_pmc.sh
echo visit30 done
