#!/bin/bash
# Round 2, visit 63: the N-rank control flow of bench.py after this round's changes (2 ranks sharing the box's GPU, both shard
# modes; numbers meaningless), and the refusal without a second GPU.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
L=$O/r2v63_two_rank.log; : > $L
for m in rows robots; do
  echo "== python bench.py --gpus 2 --debug-shared-gpu --shard-mode $m" >> $L
  timeout 900 python bench.py --gpus 2 --debug-shared-gpu --shard-mode $m --steps 1 --warmup 1 --no-cpu-baseline --match-queries 8192 2>&1 | grep -v amdgpu | tail -1 | cut -c1-700 >> $L
done
echo "== python bench.py --gpus 2 (one GPU visible)" >> $L
timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "bench.py:" | head -2 >> $L
cat $L
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
echo visit63 done
