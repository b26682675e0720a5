#!/bin/bash
# A/B of two builds of the library on ONE box, interleaved: cslam_amd/libcslam_hip_before.so (a copy made before the change) against the
# in-tree build.  conv2_2 stand-alone, the trunk's input transform + GEMM by layer, and the bench's extract leg.
#   gpurun --timeout 900 -- 'bash tools/gpu_lib_ab.sh r04_v67'
T=${1:-ab}
mkdir -p gpurun_out
python -c "import torch; torch.zeros(1, device='cuda')" > /dev/null 2>&1
OUT=gpurun_out/${T}_lib_ab.log
: > $OUT
for round in 1 2 3; do
  for which in before after; do
    if [ $which = before ]; then export CSLAM_HIP_LIB=$PWD/cslam_amd/libcslam_hip_before.so; else unset CSLAM_HIP_LIB; fi
    echo "== round $round, $which" >> $OUT
    python tools/perf_direct_conv.py 256 2>&1 | grep -E "conv2_2.*direct|max" | head -2 >> $OUT
    python tools/perf_wino_gemm.py 256 2>&1 | grep -E "pair|input" | tail -12 >> $OUT
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], 'extract_only', d['extract_only'])" >> $OUT
  done
done
grep -E "==|bench|conv2_2" $OUT
