#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for t in 128 256; do echo "== TILE=$t"; CSLAM_MFMA_TILE=$t timeout 600 python tools/perf_match.py 100000 4096 512,1024,2048,4096,8192 2>&1 | grep "n=100000"; done
