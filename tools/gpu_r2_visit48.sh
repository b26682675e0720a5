#!/bin/bash
# Round 2, visit 48: whole GPU suite, smoke, bench line, MAC at 1e6 poses (cslam_fiedler and the torch-driven solver), C5 rehearsal,
# rocprofv3 evidence (stats + PMC).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/r2v48_tests_gpu.log; tail -5 $O/r2v48_tests_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 > $O/r2v48_smoke.log; cat $O/r2v48_smoke.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/r2v48_bench.json 2> $O/r2v48_bench.err; cat $O/r2v48_bench.json; tail -3 $O/r2v48_bench.err
for s in chain_hip chain_gpu; do
  CSLAM_MAC_TIMING=1 timeout 900 python tools/perf_acm.py 125000 20000 1000 $s 2>&1 | grep -v amdgpu | grep "fiedler\|per FW\|select" | cut -c1-400 >> $O/r2v48_perf_acm_1M.log
done
tail -3 $O/r2v48_perf_acm_1M.log
timeout 900 python tools/perf_c5.py 12500 8 1000 250 2>&1 | grep -v amdgpu | tail -6 > $O/r2v48_perf_c5.log; cat $O/r2v48_perf_c5.log
bash tools/gpu_r2_pmc.sh
echo visit48 done
