#!/bin/bash
# Board power / shader clock beside steady loads (tools/power_trace.py + tools/steady_loop.py): the whole step, then one kernel at a time.
#   gpurun --timeout 900 -- 'bash tools/gpu_power_visit.sh r04_v65'
T=${1:-power}
mkdir -p gpurun_out
python -c "import torch; torch.zeros(1, device='cuda')" > /dev/null 2>&1      # page the image in before anything is timed
{
python tools/power_trace.py -- python bench.py --steps 400 --warmup 5 --no-cpu-baseline
python tools/power_trace.py -- python tools/steady_loop.py peak16 12
python tools/power_trace.py -- python tools/steady_loop.py gemm 12 28 512 512
python tools/power_trace.py -- python tools/steady_loop.py gemm 12 56 256 256
python tools/power_trace.py -- python tools/steady_loop.py match 12 16384
python tools/power_trace.py -- python tools/steady_loop.py stem 12
} > gpurun_out/${T}_power_trace.jsonl 2> gpurun_out/${T}_power_trace.err
python - <<PY
import json
for l in open("gpurun_out/${T}_power_trace.jsonl"):
    d = json.loads(l)
    print(d["command"][-60:], "| cap", d["power_cap_W"], "| busy W", d["power_W_busy"], "| busy sclk", d["sclk_MHz_busy"], "|", d["child_stdout_tail"][:120])
PY
