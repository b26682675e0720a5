#!/bin/bash
# Round 2, visit 9: the junction solve of the MAC Fiedler sweep as HIP kernels (cslam_chol_solve4_dev), Cholesky variants,
# and where select_candidates at 1e6 poses spends its time.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests/test_mac_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/r2v9_mac_tests.log; cat $O/r2v9_mac_tests.log
timeout 600 python tools/perf_chol.py 2>&1 | grep -v amdgpu > $O/r2v9_perf_chol.log; cat $O/r2v9_perf_chol.log
rm -f $O/r2v9_perf_acm.log
for c in blocked lib; do
  echo "== CSLAM_MAC_CHOL=$c" >> $O/r2v9_perf_acm.log
  CSLAM_MAC_CHOL=$c CSLAM_MAC_TIMING=1 timeout 900 python tools/perf_acm.py 125000 20000 1000 chain_gpu 2>&1 | grep -v amdgpu | tail -90 >> $O/r2v9_perf_acm.log
done
cut -c1-400 $O/r2v9_perf_acm.log
echo visit9 done
