#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for iv in 2 1 2 1; do echo "== ILV=$iv TILE=256"; CSLAM_MFMA_ILV=$iv CSLAM_MFMA_TILE=256 timeout 600 python tools/perf_match.py 100000 4096 65536,100000 2>&1 | grep "n=100000"; done
echo "== ILV=2 TILE=128"; CSLAM_MFMA_ILV=2 CSLAM_MFMA_TILE=128 timeout 600 python tools/perf_match.py 100000 4096 100000 2>&1 | grep "n=100000"
make -C oracle >/dev/null 2>&1
CSLAM_MFMA_ILV=2 timeout 900 python -m pytest tests/test_nns_gpu.py -x -q -m gpu 2>&1 | tail -2
