#!/bin/bash
O=gpurun_out; mkdir -p $O
L=$O/r2v46_cumask_ab.log; : > $L
for v in "CSLAM_FIEDLER_LOOKAHEAD=cu" "A=1"; do
  echo "== $v" | tee -a $L
  env $v CSLAM_MAC_TIMING=1 timeout 300 python tools/perf_fiedler.py 125000 16000 3 2>&1 | grep -v amdgpu | cut -c1-400 | tee -a $L
done
timeout 900 python -m pytest tests/test_heads_gpu.py -x -q -s -k "float64_model_on_distinct" 2>&1 | grep -v amdgpu | tail -30 | cut -c1-600 | tee $O/r2v43_descriptor_f64.log
