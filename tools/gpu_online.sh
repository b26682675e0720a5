#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_heads_gpu.py -x -q 2>&1 | tail -12
timeout 600 python tools/perf_online.py 2>&1 | grep -v amdgpu | tail -5
