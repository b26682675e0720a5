#!/bin/bash
# Round 2, visit 23: extract chunk size (frames per backbone pass) 128 / 256 / 384 / 512.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
: > $O/r2v23_chunk.log
for c in 256 128 384 512 256; do
  echo "== --extract-chunk $c" >> $O/r2v23_chunk.log
  timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --extract-chunk $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('value', d['value'], 'extract_only', d['extract_only'], 'ms_per_step', d['ms_per_step'])" >> $O/r2v23_chunk.log
done
cat $O/r2v23_chunk.log
echo visit23 done
