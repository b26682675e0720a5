#!/bin/bash
# Final visit of the round after the row-sharded mode: full parity suite, smoke, bench line, kernel stats of the same
# command, per-rank shapes of the row-sharded step, 2-rank control flow of both shard modes on the shared GPU.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/tests_gpu.log; tail -3 $O/tests_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 > $O/smoke.log; cat $O/smoke.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench.log 2> $O/bench.err; cut -c1-700 $O/bench.log
timeout 600 python tools/perf_rows_shapes.py 2>&1 | grep -v amdgpu > $O/perf_rows_shapes.log; cat $O/perf_rows_shapes.log
for mode in rows robots; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 2 --steps 2 --warmup 1 --debug-shared-gpu --no-cpu-baseline --shard-mode $mode 2>&1 | grep '"metric"' | cut -c1-1400
done > $O/two_rank_shared_gpu.log; cat $O/two_rank_shared_gpu.log
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_r01
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01 -o r01 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1
find $O/prof_r01 -name "*kernel_stats.csv" | head -1 | xargs head -8
echo all done
