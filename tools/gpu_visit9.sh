#!/bin/bash
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python tools/perf_zgemm_variants.py > $out/zgemm_variants.log 2>&1
for c in 256 512 1024; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --extract-chunk $c > $out/bench_chunk$c.json 2> $out/bench_chunk$c.err; done
cat $out/zgemm_variants.log; for c in 256 512 1024; do python -c "
import json,sys
d=json.load(open('$out/bench_chunk$c.json')); print('chunk $c', d['value'], d['ms_per_step'], d['extract_only'], d['match_only'])"; done
