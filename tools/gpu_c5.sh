#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_lcsm_gpu.py tests/test_full_loop_gpu.py tests/test_scancontext_gpu.py tests/test_mac_gpu.py -x -q 2>&1 | tail -6
timeout 2000 python tools/perf_c5.py ${C5_P:-12500} 8 1000 250 2>&1 | grep -v amdgpu | tail -6
