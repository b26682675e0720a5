#!/bin/bash
# End-of-round robustness pass: N=2 control flow on a shared GPU (gloo exchange), bench with other step counts, second full test run.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --debug-shared-gpu --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('steps10', d['value'], d['extract_only'], d['match_only'], d['ms_per_step'], d['roofline']['frac'])"
timeout 900 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('steps1 warmup0', d['value'], d['extract_only'], d['match_only'], d['ms_per_step'])"
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
