#!/bin/bash
# Round 4, second half: the evidence of the final tree (tag r04_v60), one gpurun visit.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=r04_v60
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/${T}_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 400 gpurun_out/${T}_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['extract_only'], d['match_only'], d['uncertified_queries'], d['roofline']['frac']); print(json.dumps(d['roofline_extract']['stem_conv'])[:700]); print(json.dumps(d['roofline_extract']['direct_conv'])[:1200])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_bench_trace -o b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_traced.json 2> gpurun_out/${T}_bench_traced.err
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_trace -o k -- python tools/extract_leg.py --iters 4 > gpurun_out/${T}_trace.log 2>&1
python tools/kernel_trace_summary.py gpurun_out/${T}_trace/k_kernel_trace.csv > gpurun_out/${T}_extract_kernels.txt 2>&1; head -20 gpurun_out/${T}_extract_kernels.txt
bash tools/pmc_kernel.sh ${T}_pmc_stem_direct conv_stem_direct_h_kernel python tools/pmc_stem_direct_target.py > gpurun_out/${T}_pmc_stem_direct.json 2>&1
bash tools/pmc_kernel.sh ${T}_pmc_direct_r_conv2_1 conv3x3_direct_r_kernel python tools/pmc_direct_r_target.py > gpurun_out/${T}_pmc_direct_r_conv2_1.json 2>&1
bash tools/pmc_kernel.sh ${T}_pmc_direct_conv2_2 conv3x3_direct_h_kernel python tools/pmc_direct_target.py 128 > gpurun_out/${T}_pmc_direct_conv2_2.json 2>&1
grep -h "mfma_busy_frac\|effective_clock\|kernel_ms_traced\|l2_miss_fabric_bytes\"" gpurun_out/${T}_pmc_*.json
CSLAM_HIP_LIB=cslam_amd/libcslam_hip_abl.so python tools/prof_stem_direct.py > gpurun_out/${T}_stem_direct_phases.log 2>&1
CSLAM_HIP_LIB=cslam_amd/libcslam_hip_abl.so python tools/prof_direct_r.py >> gpurun_out/${T}_stem_direct_phases.log 2>&1
python tools/perf_stem.py 256 5 >> gpurun_out/${T}_stem_direct_phases.log 2>&1
python tools/perf_direct_conv.py >> gpurun_out/${T}_stem_direct_phases.log 2>&1
CSLAM_CONV_DIRECT_R=0 python tools/perf_direct_conv.py 2>&1 | grep "conv2_1.*direct  " | sed 's/direct  /streaming direct kernel/' >> gpurun_out/${T}_stem_direct_phases.log
bash tools/gpu_pmc_match.sh ${T}m > gpurun_out/${T}m_pmc_match.log 2>&1; tail -5 gpurun_out/${T}m_pmc_match.log
for m in rows robots; do timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --debug-shared-gpu --no-cpu-baseline --shard-mode $m > gpurun_out/${T}_two_rank_$m.json 2> gpurun_out/${T}_two_rank_$m.err; tail -1 gpurun_out/${T}_two_rank_$m.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('two ranks $m', d['value'], d['ms_per_step'], d['sharded_check'])"; done
python -m pytest tests -x -q -m gpu > gpurun_out/${T}_tests_gpu.log 2>&1; tail -3 gpurun_out/${T}_tests_gpu.log
find gpurun_out/${T}* -name "*.csv" -size +8M -delete
