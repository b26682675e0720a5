#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests/test_nns_gpu.py -x -q -m gpu 2>&1 | tail -3
for iv in 1 0; do for t in 256 128; do echo "== ILV=$iv TILE=$t"; CSLAM_MFMA_ILV=$iv CSLAM_MFMA_TILE=$t timeout 600 python tools/perf_match.py 100000 4096 65536,100000 2>&1 | grep "n=100000"; done; done
echo "== ILV=1 d=512"; timeout 300 python tools/perf_match.py 100000 512 100000 2>&1 | grep "n=100000"
