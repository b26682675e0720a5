#!/bin/bash
# PMC passes (each counter set in its own run) over the kernels of one extract pass: tools/pmc_extract_pass.py sums them per kernel
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r03_pmcx}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R; export TMPDIR=/tmp
CMD="python tools/extract_leg.py --iters 2"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- $CMD > $O/write.log 2>&1
python tools/pmc_extract_pass.py $O $tag > $O/summary.json 2> $O/summary.err; head -60 $O/summary.json; tail -3 $O/summary.err
cp profiles/${tag}_extract_pass_bytes.json $O/ 2>/dev/null
find $O -name "*.csv" -size +20M -delete
