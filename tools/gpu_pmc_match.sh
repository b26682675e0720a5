#!/bin/bash
# PMC passes for the candidate stage of the matcher, each counter set in its own run (rocprofv3 --pmc with --kernel-trace only).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r03_pmc}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R; export TMPDIR=/tmp
CMD="python tools/pmc_match_target.py"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- $CMD > $O/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -o t -- $CMD > $O/tcc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o s -- $CMD > $O/sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- $CMD > $O/trace.log 2>&1
python tools/pmc_match_summary.py $O $tag > $O/summary.json 2> $O/summary.err; cat $O/summary.json; tail -3 $O/summary.err
# keep what travels back small: the per-dispatch CSVs of the counter passes only
find $O -name "*.csv" -size +20M -delete
