#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_heads_gpu.py -x -q -s -k "float64_model_on_distinct" 2>&1 | grep -v amdgpu | tail -12 | tee $O/r2v43_descriptor_f64.log
