#!/bin/bash
# Round 2, visit 5: where the time of the fp16-pair one-kernel convolution goes (timing-only ablations), C-ABI exchange test.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
: > $O/r2v5_fused_h_ablation.log
for d in 0 1 2 4 5 6 8 16 31; do
  echo "== CSLAM_WFH_DBG=$d (1 weights from L1, 2 no transform, 4 no MFMA, 8 no patch loads, 16 no output transform)" >> $O/r2v5_fused_h_ablation.log
  CSLAM_WFH_DBG=$d timeout 300 python tools/perf_fused_h.py 256 3 2>&1 | grep "fp16 pairs" >> $O/r2v5_fused_h_ablation.log
done
cat $O/r2v5_fused_h_ablation.log
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_c_client_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/r2v5_comm_tests.log; tail -4 $O/r2v5_comm_tests.log
timeout 900 python -m pytest tests/test_heads_gpu.py -x -q -m gpu -k "fused_winograd_h" 2>&1 | tail -8 > $O/r2v5_fused_h_tests.log; tail -4 $O/r2v5_fused_h_tests.log
echo visit5 done
