#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_heads_gpu.py tests/test_full_loop_gpu.py -x -q 2>&1 | tail -8
for m in direct winograd2 winograd; do timeout 600 python tools/perf_c2.py 10000 1000 $m 2>&1 | tail -1; done
timeout 600 python tools/perf_c2.py 10000 250 winograd 2>&1 | tail -1
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
