#!/bin/bash
# round-2 PMC passes of the match leg's dominant kernel (sim_topk_mfma_kernel, 100k queries x 100k rows), summarised by kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_tcc
bash tools/gpu_pmc.sh > /dev/null 2>&1
cd $R
python tools/pmc_summary.py gpurun_out r02m 2>&1 | tail -25
cp profiles/r02m_pmc_summary.json $O/r02m_pmc_summary.json
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_tcc
