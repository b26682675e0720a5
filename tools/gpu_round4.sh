#!/bin/bash
# Full GPU-box visit (round-1 final state): parity tests, smoke, bench, rocprof kernel stats of the bench,
# steady-state kernel trace + PMC traffic of the extract leg, PMC passes of the match leg, perf probes.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/tests_gpu.log; tail -3 $O/tests_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 > $O/smoke.log; cat $O/smoke.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench.log 2> $O/bench.err; cat $O/bench.log
timeout 900 python tools/perf_online.py 2>&1 | grep -v amdgpu > $O/perf_online.log; cat $O/perf_online.log
timeout 600 python tools/perf_sc.py 2>&1 | grep -v amdgpu > $O/perf_sc.log; tail -5 $O/perf_sc.log
timeout 300 python tools/perf_heads.py 256 2>&1 | grep -v amdgpu > $O/perf_heads.log; cat $O/perf_heads.log
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_r01 $O/ext_trace $O/ext_fetch $O/ext_write $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/pmc_sq
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01 -o r01 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1
E="python $R/tools/extract_leg.py --iters 4"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/ext_trace -o e -- $E > $O/ext_trace.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/ext_fetch -o f -- $E > $O/ext_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/ext_write -o w -- $E > $O/ext_write.log 2>&1
CMD="python $R/bench.py --steps 1 --warmup 1 --no-extract --no-cpu-baseline --match-queries 100000 --batch 1024"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $CMD > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $CMD > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -o t -- $CMD > $O/pmc_tcc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o s -- $CMD > $O/pmc_sq.log 2>&1
cd $R
python tools/kernel_trace_summary.py $(find $O/ext_trace -name "*kernel_trace.csv" | head -1) > $O/extract_kernels.txt 2>&1; cat $O/extract_kernels.txt
for c in FETCH_SIZE WRITE_SIZE; do d=ext_fetch; [ $c = WRITE_SIZE ] && d=ext_write
  python tools/kernel_trace_summary.py $(find $O/$d -name "*counter_collection.csv" | head -1) --pmc $c >> $O/extract_pmc.txt 2>&1; done
cat $O/extract_pmc.txt
# afterwards, in the build container: python tools/pmc_summary.py gpurun_out <tag>; python tools/pmc_extract_summary.py gpurun_out <tag>
echo all done
