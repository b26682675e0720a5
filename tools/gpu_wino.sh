#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_heads_gpu.py tests/test_full_loop_gpu.py -x -q 2>&1 | tail -5
for mode in ${MODES:-winograd}; do
  timeout 900 python bench.py --steps 3 --warmup 1 --backbone-conv $mode --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$mode', d['value'], d['extract_only'], d['match_only'], d['backbone_conv'], d['roofline']['frac'], d['roofline_extract'])"
done
ls tunableop* 2>/dev/null
