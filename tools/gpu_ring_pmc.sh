#!/bin/bash
# PMC passes (each counter set in its own run, --kernel-trace only) over the candidate-stage variants.
# Usage: gpurun -- bash tools/gpu_ring_pmc.sh <tag> [nq] [variants]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r05_pmc}; nq=${2:-100000}; vs=${3:--1,0,1}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R; export TMPDIR=/tmp
export CSLAM_HIP_LIB=$R/cslam_amd/libcslam_hip_abl.so
CMD="python tools/pmc_ring_target.py $nq $vs 2"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -o t -- $CMD > $O/tcc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o s -- $CMD > $O/sq.log 2>&1
python tools/pmc_ring_summary.py $O > $O/summary.json 2> $O/summary.err; cat $O/summary.json; tail -3 $O/summary.err; tail -4 $O/sq.log
find $O -name "*.csv" -size +8M -delete
