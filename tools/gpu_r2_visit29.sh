#!/bin/bash
# Round 2, visit 29: channels-last VLAD head, PCA with the unit bound; heads tests, extract leg.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 1500 python -m pytest tests/test_heads_gpu.py tests/test_full_loop_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/r2v29_tests.log; cat $O/r2v29_tests.log
timeout 600 python tools/extract_leg.py --iters 4 2>&1 | grep -v amdgpu | tail -1
timeout 600 python tools/extract_leg.py --iters 4 --batch 512 2>&1 | grep -v amdgpu | tail -1
echo visit29 done
