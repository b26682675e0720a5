#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python tools/perf_dgemm.py 2>&1 | grep -v amdgpu | tee $O/r2v39_perf_dgemm.log
