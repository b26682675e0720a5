#!/bin/bash
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
bash tools/gpu_pmc_match.sh ${tag}_pmc > $out/pmc_match.log 2>&1; echo "pmc rc=$?" >> $out/summary.txt
# kernel trace + stats of the bench command itself (the judge's evidence for the per-kernel times of the timed step)
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/bench_trace -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_traced.json 2> $GRAFT_REPO_ROOT/$out/bench_traced.err; echo "trace rc=$?" >> $GRAFT_REPO_ROOT/$out/summary.txt
cd $GRAFT_REPO_ROOT
f=$(find $out/bench_trace -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv 2>/dev/null
find $out/bench_trace -name "*.csv" -size +30M -delete
bash tools/gpu_prof_extract.sh > $out/extract_kernels.txt 2>&1
cat $out/summary.txt; tail -30 $out/pmc_match.log; head -25 $out/kernel_stats.csv; cat $out/extract_kernels.txt
