#!/bin/bash
# ScanContext bring-up on the GPU box: parity tests, then the perf tool.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_scancontext_gpu.py -x -q 2>&1 | tail -25
timeout 600 python tools/perf_sc.py 2>&1 | tee gpurun_out/perf_sc.log | tail -12
