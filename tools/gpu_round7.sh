#!/bin/bash
# Visit after the fused conv1_2 kernel: parity tests, smoke, bench, rocprof kernel stats of the bench, steady-state
# kernel trace + FETCH/WRITE_SIZE of the extract leg (the match-leg PMC passes of tools/gpu_round4.sh are unchanged).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/tests_gpu.log; tail -3 $O/tests_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 > $O/smoke.log; cat $O/smoke.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench.log 2> $O/bench.err; cat $O/bench.log
[ -n "${EXTRA:-}" ] && { timeout 600 python $EXTRA 2>&1 | grep -v amdgpu > $O/extra.log; cat $O/extra.log; }
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_r01 $O/ext_trace $O/ext_fetch $O/ext_write
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01 -o r01 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1
E="python $R/tools/extract_leg.py --iters 4"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/ext_trace -o e -- $E > $O/ext_trace.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/ext_fetch -o f -- $E > $O/ext_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/ext_write -o w -- $E > $O/ext_write.log 2>&1
cd $R
python tools/kernel_trace_summary.py $(find $O/ext_trace -name "*kernel_trace.csv" | head -1) > $O/extract_kernels.txt 2>&1; cat $O/extract_kernels.txt
rm -f $O/extract_pmc.txt
for c in FETCH_SIZE WRITE_SIZE; do d=ext_fetch; [ $c = WRITE_SIZE ] && d=ext_write
  python tools/kernel_trace_summary.py $(find $O/$d -name "*counter_collection.csv" | head -1) --pmc $c >> $O/extract_pmc.txt 2>&1; done
cat $O/extract_pmc.txt
echo round7 done
