#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_heads_gpu.py -x -q -s -k "float64_model_on_distinct" > $O/r2v47_descriptor_f64_full.log 2>&1
grep -v amdgpu $O/r2v47_descriptor_f64_full.log | grep "descriptor error\|torch fp32\|passed\|failed\|Error\|assert" | cut -c1-600 | tee $O/r2v43_descriptor_f64.log
