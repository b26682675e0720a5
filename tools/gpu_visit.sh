#!/bin/bash
# One GPU-box visit through gpurun:   gpurun -- 'bash tools/gpu_visit.sh <tag> <item> [<item> ...]'
# Everything is written under gpurun_out/<tag>/ (merged back by gpurun); what should be judged is copied into profiles/ by hand.
#   tests        whole GPU suite (pytest -m gpu)            smoke        __graft_entry__.smoke()
#   bench        python bench.py --steps 20 --warmup 2      trace        rocprofv3 --kernel-trace --stats of the bench command
#   fullsize     tests/test_fullsize_gpu.py -s (C3 / C4 / C5 at their real sizes, timing prints)
#   match_ab     candidate stage: fp16 pairs vs f32-input MFMA (tools/perf_match_ab.py), also with 128 x 128 tiles forced
#   match_ldm    pair stage: placement of the LDS-DMA requests (tools/perf_match_ldm.py)
#   match_patch  pair stage: XCD patch shapes (tools/perf_match_patch.py)
#   pmc_match    rocprofv3 --pmc passes of the candidate stage (tools/gpu_pmc_match.sh)
#   pmc_extract  HBM bytes of one extract pass by kernel (tools/gpu_pmc_extract.sh)
#   extract      kernel split of 256-frame extract passes (tools/gpu_prof_extract.sh)
#   zform        Z form vs 36-plane form per trunk layer (tools/perf_zform.py, tools/perf_zgemm_variants.py)
#   chunks       bench value by frames per backbone pass (256 / 512 / 1024)
#   acm_small    select_candidates on 2k-16k-pose graphs, default solver vs the reference's path (tools/perf_acm_small.py)
#   c5           C5 rehearsal on one GPU, drain and drain-async (tools/perf_c5.py)
#   two_rank     bench.py --gpus 2 --debug-shared-gpu in both shard modes (two ranks sharing the box's GPU, collectives through gloo)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; out=gpurun_out/$tag; mkdir -p $R/$out; cd $R; export TMPDIR=/tmp
note() { echo "$1 rc=$2" >> $out/summary.txt; }
for w in "$@"; do
  case $w in
    tests) timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests_gpu.log 2>&1; note tests $?; tail -3 $out/tests_gpu.log >> $out/summary.txt;;
    heads) timeout 900 python -m pytest tests/test_heads_gpu.py -x -q -s > $out/tests_heads.log 2>&1; note heads $?; tail -3 $out/tests_heads.log >> $out/summary.txt; grep 'vlad matrix' $out/tests_heads.log >> $out/summary.txt;;
    smoke) timeout 600 python __graft_entry__.py smoke > $out/smoke.log 2>&1; note smoke $?;;
    bench) timeout 900 python bench.py --steps 20 --warmup 2 > $out/bench.json 2> $out/bench.err; note bench $?;;
    trace) (cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/bench_trace -o b -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/$out/bench_traced.json 2> $R/$out/bench_traced.err); note trace $?
           cp "$(find $out/bench_trace -name '*kernel_stats.csv' | head -1)" $out/kernel_stats.csv 2>/dev/null; find $out/bench_trace -name "*.csv" -size +30M -delete;;
    fullsize) timeout 1800 python -m pytest tests/test_fullsize_gpu.py -x -q -s > $out/tests_fullsize.log 2>&1; note fullsize $?; grep "^C[345]" $out/tests_fullsize.log >> $out/summary.txt;;
    match_ab) timeout 600 python tools/perf_match_ab.py > $out/perf_match_ab.log 2>&1; note match_ab $?
              CSLAM_MFMA_TILE=128 timeout 300 python tools/perf_match_ab.py 100000 4096 1024,4096 > $out/perf_match_ab_tile128.log 2>&1;;
    match_ldm) timeout 600 python tools/perf_match_ldm.py > $out/perf_match_ldm.log 2>&1; note match_ldm $?;;
    match_patch) timeout 600 python tools/perf_match_patch.py > $out/perf_match_patch.log 2>&1; note match_patch $?;;
    pmc_match) bash tools/gpu_pmc_match.sh ${tag}_pmc > $out/pmc_match.log 2>&1; note pmc_match $?;;
    pmc_extract) bash tools/gpu_pmc_extract.sh ${tag}_pmcx > $out/pmc_extract.log 2>&1; note pmc_extract $?;;
    extract) bash tools/gpu_prof_extract.sh > $out/extract_kernels.txt 2>&1; note extract $?;;
    zform) timeout 600 python tools/perf_zform.py > $out/perf_zform.log 2>&1; note zform $?; timeout 300 python tools/perf_zgemm_variants.py > $out/zgemm_variants.log 2>&1;;
    chunks) for c in 256 512 1024; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --extract-chunk $c > $out/bench_chunk$c.json 2> $out/bench_chunk$c.err; done; note chunks $?;;
    lanes) for cl in "256 1" "256 2" "256 4" "512 1" "512 2" "342 3" "128 2"; do set -- $cl; timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --extract-chunk $1 --extract-lanes $2 > $out/bench_c$1_l$2.json 2> $out/bench_c$1_l$2.err; python -c "import json,sys; d=json.load(open('$out/bench_c$1_l$2.json')); print('chunk $1 lanes $2', d['value'], d['ms_per_step'], d.get('extract_only'))" >> $out/summary.txt; done; note lanes $?;;
    c2) for c in 250 500; do timeout 900 python tools/perf_c2.py 10000 $c winograd > $out/perf_c2_$c.log 2>&1; grep "^C2" $out/perf_c2_$c.log >> $out/summary.txt; done; note c2 $?;;
    acm_small) timeout 900 python tools/perf_acm_small.py > $out/perf_acm_small.log 2>&1; note acm_small $?;;
    c5) timeout 1200 python tools/perf_c5.py 12500 8 1000 250 drain > $out/perf_c5_drain.log 2>&1; note c5_drain $?
        timeout 1200 python tools/perf_c5.py 12500 8 1000 250 drain-async > $out/perf_c5_drain_async.log 2>&1; note c5_drain_async $?;;
    two_rank) timeout 900 python bench.py --gpus 2 --debug-shared-gpu --steps 2 --warmup 1 --no-cpu-baseline > $out/two_rank_rows.json 2> $out/two_rank_rows.err; note two_rank_rows $?
              timeout 900 python bench.py --gpus 2 --debug-shared-gpu --shard-mode robots --steps 2 --warmup 1 --no-cpu-baseline > $out/two_rank_robots.json 2> $out/two_rank_robots.err; note two_rank_robots $?;;
    *) echo "unknown item $w" >> $out/summary.txt;;
  esac
done
cat $out/summary.txt
