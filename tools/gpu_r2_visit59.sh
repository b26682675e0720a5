#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m cProfile -s cumulative tools/perf_c5.py 6000 8 1000 250 drain 2>&1 | grep -v amdgpu | head -75 | cut -c1-200 > $O/r2v59_c5_profile.log
cat $O/r2v59_c5_profile.log | head -70
