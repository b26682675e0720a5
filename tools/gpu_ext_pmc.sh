#!/bin/bash
# extract leg: throughput + FETCH_SIZE per kernel (halo re-read check)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_heads_gpu.py -x -q -k "winograd or first_layer" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
rm -rf $O/ext_trace $O/ext_fetch
E="python $R/tools/extract_leg.py --iters 4"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/ext_trace -o e -- $E > $O/ext_trace.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/ext_fetch -o f -- $E > $O/ext_fetch.log 2>&1
cd $R
python tools/kernel_trace_summary.py $(find $O/ext_trace -name "*kernel_trace.csv" | head -1) | head -14
python tools/kernel_trace_summary.py $(find $O/ext_fetch -name "*counter_collection.csv" | head -1) --pmc FETCH_SIZE | head -6
