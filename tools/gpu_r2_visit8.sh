#!/bin/bash
# Round 2, visit 8: host profile of the C5 rehearsal's matching / exchange legs (where do 2.5 s + 3.1 s go?).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 1200 python -c "
import cProfile, pstats, sys, io
sys.argv = ['perf_c5.py', '6000', '8', '1000', '250']
sys.path.insert(0, 'tools')
import runpy
pr = cProfile.Profile()
pr.enable()
runpy.run_path('tools/perf_c5.py', run_name='__main__')
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
" 2>&1 | grep -v amdgpu > $O/r2v8_c5_profile.log; cat $O/r2v8_c5_profile.log | cut -c1-200
timeout 600 python -m pytest tests/test_nns_gpu.py -x -q -m gpu -k "callers_device or multi_bank" 2>&1 | tail -4
echo visit8 done
