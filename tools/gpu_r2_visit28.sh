#!/bin/bash
# Round 2, visit 28: input transform with 16-byte stores (lane-pair exchange), A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
make -C oracle >/dev/null 2>&1
CSLAM_WIN_PAIR16=1 timeout 900 python -m pytest tests/test_wino_gemm_gpu.py tests/test_heads_gpu.py -x -q -m gpu -k "input_transform or split16" 2>&1 | tail -3
L=$O/r2v28_ab.log; : > $L
for v in 0 1 0 1; do
  echo "== CSLAM_WIN_PAIR16=$v" >> $L
  CSLAM_WIN_PAIR16=$v timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu >> $L
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from cslam_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
p = lambda t: C.c_void_p(t.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, hw, cin in (("conv2_2", 112, 128), ("conv3_2", 56, 256), ("conv4_2", 28, 512)):
    x = torch.relu(torch.randn((256, cin, hw, hw), device="cuda")).contiguous(memory_format=torch.channels_last)
    slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    _lib.check(lib.cslam_absmax_dev(p(x), x.numel(), p(slot), st))
    T = 256 * (hw // 4) ** 2
    V2 = torch.empty((36, T, cin), device="cuda")
    f = lambda: _lib.check(lib.cslam_wino4_input_h2_dev(p(x), 256, hw, hw, cin, p(slot), p(V2), st))
    f(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: input transform {ms:.3f} ms = {(x.numel() + V2.numel()) * 4 / ms / 1e6:.0f} GB/s")
PY
done
cat $L
for v in 0 1; do echo "== CSLAM_WIN_PAIR16=$v" >> $O/r2v28_extract.log; CSLAM_WIN_PAIR16=$v timeout 600 python tools/extract_leg.py --iters 4 2>&1 | grep -v amdgpu | tail -1 >> $O/r2v28_extract.log; done; cat $O/r2v28_extract.log
echo visit28 done
