#!/bin/bash
# Record TunableOp selections for GEMM shapes missing from cslam_amd/vpr/tunableop_gfx950.csv (existing rows are kept).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
cp $R/cslam_amd/vpr/tunableop_gfx950.csv /tmp/tunableop_results0.csv
CSLAM_TUNED_GEMM=0 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=/tmp/tunableop_results.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=300 \
  timeout 1500 python $R/tools/extract_leg.py --iters 1 --batch 256 > $O/retune.log 2>&1
ls -la /tmp/tunableop_results*.csv; cp /tmp/tunableop_results0.csv $O/tunableop_gfx950.csv; cat $O/tunableop_gfx950.csv | cut -c1-120
cp /tmp/tunableop_results0.csv $R/cslam_amd/vpr/tunableop_gfx950.csv
cd $R; python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from cslam_amd.vpr.netvlad import NetVLAD
nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096}, None)
fr = torch.randint(0, 256, (256, 480, 640, 3), device="cuda", dtype=torch.uint8)
for _ in range(2): nv.compute_embeddings_device(fr)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): nv.compute_embeddings_device(fr)
torch.cuda.synchronize(); print("NetVLAD chunk 256 with the re-recorded table: %.0f frames/s" % (256 * 8 / (time.perf_counter() - t0)))
PY
