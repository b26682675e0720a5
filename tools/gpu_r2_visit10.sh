#!/bin/bash
# Round 2, visit 10: the stem kernel (conv1_1 folded into the one-kernel conv1_2) -- tests, timing -- and a host profile of one
# Fiedler pair at 1e6 poses / 16k loop edges.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests/test_heads_gpu.py -x -q -m gpu -k "stem or fused_winograd_f4 or fused_winograd_h_scales" 2>&1 | tail -8 > $O/r2v10_stem_tests.log; cat $O/r2v10_stem_tests.log
timeout 300 python tools/perf_stem.py 256 5 2>&1 | grep -v amdgpu > $O/r2v10_perf_stem.log; cat $O/r2v10_perf_stem.log
timeout 600 python -c "
import cProfile, pstats, sys, io, runpy
sys.argv = ['perf_mac.py', '125000', '16000']
pr = cProfile.Profile(); pr.enable()
runpy.run_path('tools/perf_mac.py', run_name='__main__')
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:7000])
" 2>&1 | grep -v amdgpu > $O/r2v10_mac_profile.log; cut -c1-180 $O/r2v10_mac_profile.log
echo visit10 done
