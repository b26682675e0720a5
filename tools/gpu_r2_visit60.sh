#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python tools/exp_overlap.py 2>&1 | grep -v amdgpu | tail -3 | tee $O/r2v60_overlap.log
