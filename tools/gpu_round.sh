#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.  Outputs -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
make -C oracle >/dev/null 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -40 > $O/tests_gpu.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 1200 python bench.py --steps 3 --warmup 1 > $O/bench.log 2> $O/bench.err
tail -3 $O/tests_gpu.log; tail -2 $O/smoke.log; cat $O/bench.log; tail -5 $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01 -o r01 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1
ls -R $O/prof_r01 | head -20
