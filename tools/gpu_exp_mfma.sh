#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests/test_nns_gpu.py tests/test_heads_gpu.py -x -q -m gpu 2>&1 | tail -3
for t in 256 128; do echo "== TILE=$t"; CSLAM_MFMA_TILE=$t timeout 600 python tools/perf_match.py 100000 4096 100000,65536 2>&1 | grep "n=100000"; done
python tools/perf_heads.py 256 2>&1 | grep -v amdgpu | head -3
cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out
CMD="python $R/bench.py --steps 1 --warmup 1 --no-extract --no-cpu-baseline --match-queries 100000 --batch 1024"
rm -rf $O/pmc_fetch $O/pmc_tcc
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $CMD > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -o t -- $CMD > $O/pmc_tcc.log 2>&1
echo pmc done
