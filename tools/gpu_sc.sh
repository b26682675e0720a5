#!/bin/bash
# ScanContext bring-up on the GPU box: parity tests, then the perf tools.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_scancontext_gpu.py -x -q 2>&1 | tail -15
timeout 600 python tools/perf_sc.py 2>&1 | tee gpurun_out/perf_sc.log | tail -12
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/sc_prof" -- python "$GRAFT_REPO_ROOT/tools/perf_sc.py" --nq 8192 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/sc_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep "sc_" "$f" | cut -c1-160
timeout 900 python tools/perf_online.py 2>&1 | tee gpurun_out/perf_online.log | tail -8
