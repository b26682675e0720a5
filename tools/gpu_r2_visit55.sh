#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/r2v55_bench.json 2> $O/r2v55_bench.err; cut -c1-200 $O/r2v55_bench.json; tail -3 $O/r2v55_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2v55_bench.json"))
r = d["roofline_extract"]
print({k: r[k] for k in r if k not in ("pair_gemm", "fused_conv", "stem_conv", "fp32_input_transform")})
print(r["fp32_input_transform"]["kernel_ms"], r["fp32_input_transform"]["achieved"])
PY
