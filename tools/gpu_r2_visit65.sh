#!/bin/bash
# MFMA-busy counters of the extract kernels (pmc_targets.py launches), one --pmc pass of its own
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/r2_pmc_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/r2_pmc_sq -o s -- python $R/tools/pmc_targets.py > $O/r2_pmc_sq.log 2>&1
cd $R
python tools/pmc_mfma_by_kernel.py $O/r2_pmc_sq 2>&1 | tail -20
cp profiles/r02_mfma_busy_by_kernel.json $O/
rm -rf $O/r2_pmc_sq
