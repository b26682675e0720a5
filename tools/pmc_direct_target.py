"""rocprofv3 target: the direct one-kernel convolution (csrc/conv_direct_h.hip) on conv2_2's shape (default) or conv2_1's.
    python tools/pmc_direct_target.py [cin=128] [frames=256]"""
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd import _lib
from cslam_amd.vpr import winograd as wg
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pool = cin == 128
torch.manual_seed(3)
x = torch.relu(torch.randn((B, cin, 112, 112), device="cuda")).contiguous(memory_format=torch.channels_last)
w = torch.randn(128, cin, 3, 3, device="cuda") / (3.0 * cin ** 0.5)
b = torch.randn(128, device="cuda")
Wd = wg.direct_pair_weights(w)
slot = torch.zeros(1, dtype=torch.float32, device="cuda")
_lib.check(_lib.load().cslam_absmax_dev(x.data_ptr(), x.numel(), slot.data_ptr(), torch.cuda.current_stream().cuda_stream))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(4):
    e0.record()
    y = wg.conv3x3_direct_h(x, Wd, b, True, pool, slot, None)
    e1.record()
    torch.cuda.synchronize()
    print("cin", cin, "frames", B, "kernel ms", round(e0.elapsed_time(e1), 3), "fp16 TFLOP/s issued", round(3 * 2.0 * B * 112 * 112 * 9 * cin * 128 / e0.elapsed_time(e1) / 1e9, 1))

import os, ctypes
if os.environ.get("CSLAM_CD_PROF"):
    lib = ctypes.CDLL(os.environ["CSLAM_HIP_LIB"])
    buf = torch.zeros(4, dtype=torch.int64, device="cuda")
    lib.cslam_debug_cd_prof_dev(ctypes.c_void_p(buf.data_ptr()))
    y = wg.conv3x3_direct_h(x, Wd, b, True, pool, slot, None)
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    print("prof (wave 0 of workgroup 0): per block: stage loop %.0f ticks, epilogue %.0f ticks, blocks %d" % (t[0] / max(t[2], 1), t[1] / max(t[2], 1), t[2]))
