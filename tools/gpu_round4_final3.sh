#!/bin/bash
# Round 4, last visit (tag r04_v75): the bench command under rocprofv3 --kernel-trace --stats, the extract pass by kernel, two ranks on one GPU.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=r04_v75
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_bench_trace -o b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_traced.json 2> gpurun_out/${T}_bench_traced.err
cp gpurun_out/${T}_bench_trace/*/b_kernel_stats.csv gpurun_out/${T}_kernel_stats.csv 2>/dev/null || cp gpurun_out/${T}_bench_trace/b_kernel_stats.csv gpurun_out/${T}_kernel_stats.csv
head -8 gpurun_out/${T}_kernel_stats.csv | cut -c1-150
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_trace -o k -- python tools/extract_leg.py --iters 4 > gpurun_out/${T}_trace.log 2>&1
python tools/kernel_trace_summary.py $(find gpurun_out/${T}_trace -name "k_kernel_trace.csv" | head -1) > gpurun_out/${T}_extract_kernels.txt 2>&1; head -14 gpurun_out/${T}_extract_kernels.txt | cut -c1-120
for m in rows robots; do timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --debug-shared-gpu --no-cpu-baseline --shard-mode $m > gpurun_out/${T}_two_rank_$m.json 2> gpurun_out/${T}_two_rank_$m.err; tail -1 gpurun_out/${T}_two_rank_$m.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('two ranks $m', d['value'], d['ms_per_step'], d['sharded_check'], d.get('board') is not None)"; done
find gpurun_out/${T}* -name "*.csv" -size +8M -delete
