#!/bin/bash
# row-sharded bank: merge-kernel parity tests, then the 2-rank control flow of both shard modes on the one GPU of
# the box (--debug-shared-gpu: collectives staged through gloo/CPU, numbers meaningless), then the default bench line
mkdir -p gpurun_out
python -m pytest tests/test_nns_gpu.py -q -x -k "row_sharded" 2>&1 | tail -3
for mode in rows robots; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 2 --steps 1 --warmup 1 --batch 256 --match-queries 8192 --debug-shared-gpu --shard-mode $mode \
    2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['value'], d['config']['parallelism'], d['config']['bank_rows_per_gpu'], d['match_only'], d['roofline']['frac'])"
done
python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -c 600 gpurun_out/bench.log
