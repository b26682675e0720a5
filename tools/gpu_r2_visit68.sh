#!/bin/bash
# Round 2, visit 68: whole GPU suite + smoke on the final tree
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/r2v68_tests_gpu.log; tail -3 $O/r2v68_tests_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 > $O/r2v68_smoke.log; cat $O/r2v68_smoke.log
echo visit68 done
