#!/bin/bash
# Final visit of round 1: everything of tools/gpu_round4.sh (tests, smoke, bench, probes, rocprof stats, extract trace +
# PMC, match PMC) plus the row-sharded extras of tools/gpu_round5.sh.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
bash tools/gpu_round4.sh
timeout 600 python tools/perf_rows_shapes.py 2>&1 | grep -v amdgpu > $O/perf_rows_shapes.log; cat $O/perf_rows_shapes.log
for mode in rows robots; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 2 --steps 2 --warmup 1 --debug-shared-gpu --no-cpu-baseline --shard-mode $mode 2>&1 | grep '"metric"' | cut -c1-1400
done > $O/two_rank_shared_gpu.log; cut -c1-300 $O/two_rank_shared_gpu.log
echo round6 done
