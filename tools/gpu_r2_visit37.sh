#!/bin/bash
# kernel-level profile of select_candidates at 1e6 poses through cslam_fiedler
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof37 -o acm -- python $R/tools/perf_acm.py 125000 20000 1000 chain_hip > $O/prof37.log 2>&1
cd $R
tail -2 $O/prof37.log
f=$(find $O/prof37 -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-260 > $O/r2v37_acm_1M_kernel_stats.csv
rm -rf $O/prof37
cat $O/r2v37_acm_1M_kernel_stats.csv | cut -c1-200
