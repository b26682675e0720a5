#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for t in 256 128; do echo "== TILE=$t"; CSLAM_MFMA_TILE=$t timeout 600 python tools/perf_match.py 100000 4096 8192,16384,32768,49152,65536,100000 2>&1 | grep "n=100000"; done
echo "== TILE=256 bank 25k / 50k / 200k"; 
CSLAM_MFMA_TILE=256 timeout 300 python tools/perf_match.py 25000 4096 100000 2>&1 | grep "nq=100000"
CSLAM_MFMA_TILE=256 timeout 300 python tools/perf_match.py 50000 4096 100000 2>&1 | grep "nq=100000"
CSLAM_MFMA_TILE=256 timeout 300 python tools/perf_match.py 200000 4096 50000 2>&1 | grep "nq=50000"
