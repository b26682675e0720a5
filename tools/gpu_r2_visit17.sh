#!/bin/bash
# Round 2, visit 17: where the per-call set-up of the Fiedler solver goes (1e6 poses), after the host-side changes.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests/test_mac_gpu.py -x -q -m gpu 2>&1 | tail -3
CSLAM_MAC_TIMING=2 timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_gpu 2>&1 | grep -v amdgpu > $O/r2v17_perf_acm.log; head -12 $O/r2v17_perf_acm.log | cut -c1-600; tail -3 $O/r2v17_perf_acm.log | cut -c1-300
echo visit17 done
