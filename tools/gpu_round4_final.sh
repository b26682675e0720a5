cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r04_v42_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_v42_bench.json 2> gpurun_out/r04_v42_bench.err
tail -c 600 gpurun_out/r04_v42_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r04_v42_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['extract_only'], d['match_only'], d['uncertified_queries'], d['roofline']['frac']); print(json.dumps(d['roofline_extract']['direct_conv'])[:1500]); print(json.dumps(d['roofline_extract']['pair_gemm'])[:1500])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_v42_bench_trace -o b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_v42_bench_traced.json 2> gpurun_out/r04_v42_bench_traced.err
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04_v42_trace -o k -- python tools/extract_leg.py --iters 4 > gpurun_out/r04_v42_trace.log 2>&1
python tools/kernel_trace_summary.py gpurun_out/r04_v42_trace/k_kernel_trace.csv > gpurun_out/r04_v42_extract_kernels.txt 2>&1; cat gpurun_out/r04_v42_extract_kernels.txt | head -20
bash tools/gpu_pmc_match.sh r04_v42m > gpurun_out/r04_v42m_pmc_match.log 2>&1; tail -5 gpurun_out/r04_v42m_pmc_match.log
bash tools/pmc_kernel.sh r04_v42_pmc_direct_conv2_2 conv3x3_direct_h_kernel python tools/pmc_direct_target.py 128 > gpurun_out/r04_v42_pmc_direct_conv2_2.json 2>&1
bash tools/pmc_kernel.sh r04_v42_pmc_direct_conv2_1 conv3x3_direct_h_kernel python tools/pmc_direct_target.py 64 > gpurun_out/r04_v42_pmc_direct_conv2_1.json 2>&1
grep -h "mfma_busy_frac\|effective_clock\|kernel_ms_traced\|l2_miss_fabric_bytes\"" gpurun_out/r04_v42_pmc_direct_*.json
for m in rows robots; do timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --debug-shared-gpu --no-cpu-baseline --shard-mode $m > gpurun_out/r04_v42_two_rank_$m.json 2> gpurun_out/r04_v42_two_rank_$m.err; tail -1 gpurun_out/r04_v42_two_rank_$m.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('two ranks $m', d['value'], d['ms_per_step'], d['sharded_check'])"; done
find gpurun_out/r04_v42* -name "*.csv" -size +8M -delete
