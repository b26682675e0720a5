#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python tools/perf_solve4.py 2>&1 | grep -v amdgpu | tee $O/r2v53_solve4_bs.log
