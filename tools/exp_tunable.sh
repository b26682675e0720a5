#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
cat > /tmp/thr.py <<'PY'
import sys, time, torch
sys.path.insert(0, "/root/repo")
from cslam_amd.vpr.netvlad import NetVLAD
nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096}, None)
fr = torch.randint(0, 256, (256, 480, 640, 3), device="cuda", dtype=torch.uint8)
t0 = time.perf_counter()
for _ in range(2): nv.compute_embeddings_device(fr)
torch.cuda.synchronize(); tw = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(6): nv.compute_embeddings_device(fr)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
print(f"{sys.argv[1]}: warm-up {tw:.1f} s, {256/dt:.0f} frames/s")
PY
python /tmp/thr.py default 2>&1 | tail -1
TORCH_BLAS_PREFER_HIPBLASLT=0 python /tmp/thr.py rocblas_preferred 2>&1 | tail -1
TORCH_BLAS_PREFER_HIPBLASLT=1 python /tmp/thr.py hipblaslt_preferred 2>&1 | tail -1
cd /tmp && PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=300 timeout 1200 python /tmp/thr.py tunableop 2>&1 | tail -1
ls -la /tmp/tunableop_results*.csv 2>/dev/null; head -20 /tmp/tunableop_results0.csv 2>/dev/null | cut -c1-200
cp /tmp/tunableop_results0.csv "$GRAFT_REPO_ROOT/gpurun_out/" 2>/dev/null
