#!/bin/bash
# PMC passes for the dominant kernel (match-only bench), each counter set in its own run.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --no-extract --no-cpu-baseline --match-queries 100000 --batch 1024"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $CMD > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $CMD > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -o s -- $CMD > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -o t -- $CMD > $O/pmc_tcc.log 2>&1
ls $O/pmc_*; tail -2 $O/pmc_fetch.log
