#!/bin/bash
# kernel-level profile of the extract leg (NetVLAD, chunk 256)
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
mkdir -p "$R/gpurun_out"
cat > /tmp/extract_only.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from cslam_amd.vpr.netvlad import NetVLAD
torch.backends.cudnn.benchmark = True
nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096}, None)
fr = torch.randint(0, 256, (256, 480, 640, 3), device="cuda", dtype=torch.uint8)
for _ in range(2): nv.compute_embeddings_device(fr)
torch.cuda.synchronize()
import ctypes
for _ in range(4): nv.compute_embeddings_device(fr)
torch.cuda.synchronize()
PY
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/extract_prof" -- python /tmp/extract_only.py > /dev/null 2>&1
f=$(find "$R/gpurun_out/extract_prof" -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "naive_conv" not in r["Name"])
for r in rows[:22]:
    if "naive_conv" in r["Name"]: continue
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% x{r["Calls"]:>4}  {r["Name"][:110]}')
print("total (excl. MIOpen find-mode naive conv)", tot/1e6, "ms for 6 x 256 frames")
PY
