#!/bin/bash
O=gpurun_out; mkdir -p $O
s=$(date +%s); timeout 1500 python bench.py > $O/r2v62_bench_default.json 2> $O/r2v62_bench_default.err; e=$(date +%s)
echo "default bench.py wall: $((e-s)) s" | tee $O/r2v62_bench_wall.log
cut -c1-260 $O/r2v62_bench_default.json
