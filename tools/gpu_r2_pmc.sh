#!/bin/bash
# rocprofv3 evidence of the round-2 kernels: kernel-trace stats of the bench command, steady-state kernel split of the
# extract leg, and the PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit-miss, each in its own run) of the priced launches.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/r2_prof $O/r2_ext_trace $O/r2_pmc
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2_prof -o r02 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2_prof_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/r2_ext_trace -o e -- python $R/tools/extract_leg.py --iters 4 > $O/r2_ext_trace.log 2>&1
T="python $R/tools/pmc_targets.py"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r2_pmc/fetch -o f -- $T > $O/r2_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r2_pmc/write -o w -- $T > $O/r2_pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/r2_pmc/tcc -o t -- $T > $O/r2_pmc_tcc.log 2>&1
cd $R
python tools/kernel_trace_summary.py $(find $O/r2_ext_trace -name "*kernel_trace.csv" | head -1) > $O/r2_extract_kernels.txt 2>&1; cat $O/r2_extract_kernels.txt
cp $(find $O/r2_prof -name "*kernel_stats.csv" | head -1) $O/r2_kernel_stats.csv 2>/dev/null; head -12 $O/r2_kernel_stats.csv
python tools/pmc_by_kernel.py $O/r2_pmc "rocprofv3 --pmc passes of tools/pmc_targets.py (tools/gpu_r2_pmc.sh), round 2" 2>&1 | tee $O/r2_pmc_by_kernel.log
cp profiles/pmc_by_kernel.json $O/r2_pmc_by_kernel.json
echo pmc done
