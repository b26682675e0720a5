cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r05_v24_igemm_all; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_heads_gpu.py tests/test_configs_gpu.py -x -q -k "implicit_gemm or winograd_resnet or cosplace or c2" > $O/tests.log 2>&1; tail -4 $O/tests.log
python tools/perf_conv_igemm.py 1000 2>&1 | grep -v amdgpu.ids | head -3
timeout 600 python tools/perf_c2.py 4000 1000 winograd 2>&1 | tail -1 | tee $O/c2.log
timeout 600 python tools/perf_c2.py 10000 1000 winograd 2>&1 | tail -1 | tee -a $O/c2.log
