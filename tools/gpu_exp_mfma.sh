#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for t in 256 128; do echo "== TILE=$t"; CSLAM_MFMA_TILE=$t timeout 600 python tools/perf_match.py 100000 4096 100000,90000,77777 2>&1 | grep "n=100000"; done
echo "== default tile, other banks"
timeout 300 python tools/perf_match.py 50000 4096 100000 2>&1 | grep "nq=100000"
timeout 300 python tools/perf_match.py 123457 4096 60000 2>&1 | grep "nq=60000"
timeout 300 python tools/perf_match.py 100000 512 100000 2>&1 | grep "nq=100000"
