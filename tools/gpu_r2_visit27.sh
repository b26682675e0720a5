#!/bin/bash
# Round 2, visit 27: PCA projection on fp16 pairs (tests, timing against the f32-MFMA form, extract leg).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
timeout 900 python -m pytest tests/test_heads_gpu.py tests/test_wino_gemm_gpu.py -x -q -m gpu -k "pca or pair_gemm or descriptors or extractors" 2>&1 | tail -4 > $O/r2v27_tests.log; cat $O/r2v27_tests.log
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu > $O/r2v27_perf_pca.log
import torch, sys
sys.path.insert(0, ".")
from cslam_amd.vpr import heads
g = torch.Generator(device="cuda").manual_seed(0)
comp = heads.padded_rows(4096, 32768, "cuda"); comp.copy_(torch.randn((4096, 32768), generator=g, device="cuda") / 181.0)
pairs = heads.pca_pair_weights(comp)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in (256, 512, 64):
    x = heads.padded_rows(B, 32768, "cuda"); x.copy_(torch.randn((B, 32768), generator=g, device="cuda")); x /= x.norm(dim=1, keepdim=True)
    for tag, pr in (("f32 MFMA", None), ("fp16 pairs", pairs), ("f32 MFMA", None), ("fp16 pairs", pairs)):
        heads.pca_project(x, comp, None, None, pr); torch.cuda.synchronize()
        e0.record()
        for _ in range(10): heads.pca_project(x, comp, None, None, pr)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"PCA 32768 -> 4096, B = {B}: {tag:10s} {ms:.3f} ms = {2.0 * B * 32768 * 4096 / ms / 1e9:.0f} TFLOP/s fp32-equivalent")
PY
cat $O/r2v27_perf_pca.log
for pp in 1 0 1; do echo "== CSLAM_PCA_PAIRS=$pp" >> $O/r2v27_extract.log; CSLAM_PCA_PAIRS=$pp timeout 600 python tools/extract_leg.py --iters 4 2>&1 | grep -v amdgpu | tail -1 >> $O/r2v27_extract.log; done; cat $O/r2v27_extract.log
echo visit27 done
