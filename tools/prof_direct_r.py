#!/usr/bin/env python
"""Where a workgroup of the register-resident direct convolution (csrc/conv_direct_r.hip, VGG-16 conv2_1) spends its cycles: s_memtime
per phase of wave 0 of workgroup 0 (measurement build).
    CSLAM_HIP_LIB=cslam_amd/libcslam_hip_abl.so python tools/prof_direct_r.py [frames=256]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cslam_amd import _lib  # noqa: E402
from cslam_amd.vpr import winograd as wg  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
buf = torch.zeros(8, dtype=torch.int64, device="cuda")
assert lib.cslam_debug_dr_prof_dev(C.c_void_p(buf.data_ptr())) == 0
torch.manual_seed(3)
x = torch.relu(torch.randn((B, 64, 112, 112), device="cuda")).contiguous(memory_format=torch.channels_last)
w = torch.randn(128, 64, 3, 3, device="cuda") / 24.0
b = torch.randn(128, device="cuda")
Wr = wg.direct_r_pair_weights(w)
slot = torch.zeros(1, dtype=torch.float32, device="cuda")
_lib.check(lib.cslam_absmax_dev(x.data_ptr(), x.numel(), slot.data_ptr(), torch.cuda.current_stream().cuda_stream))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
wg.conv3x3_direct_r(x, Wr, b, True, False, slot, None)
for _ in range(3):
    buf.zero_()
    torch.cuda.synchronize()
    e0.record()
    wg.conv3x3_direct_r(x, Wr, b, True, False, slot, None)
    e1.record()
    torch.cuda.synchronize()
    h = [int(v) for v in buf.cpu().numpy()]
    n = max(h[3], 1)
    print(f"{e0.elapsed_time(e1):.3f} ms per launch; wave 0 of workgroup 0: {n} blocks, per block: six columns + staging {h[0] / n:.0f}, "
          f"epilogue {h[1] / n:.0f}, barrier {h[2] / n:.0f}, sum {sum(h[:3]) / n:.0f} cycles")
lib.cslam_debug_dr_prof_dev(None)
