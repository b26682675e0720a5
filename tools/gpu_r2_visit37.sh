#!/bin/bash
# kernel-level profile of select_candidates at 1e6 poses through cslam_fiedler
O=gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof37 -o acm -- python $GRAFT_REPO_ROOT/tools/perf_acm.py 125000 20000 1000 chain_hip > /tmp/prof37.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 /tmp/prof37.log
f=$(find /tmp/prof37 -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-260 > $O/r2v37_acm_1M_kernel_stats.csv
cat $O/r2v37_acm_1M_kernel_stats.csv | cut -c1-200
