"""PMC target: the trunk's input transform (wino4_input_h2_kernel) on conv3_1's shape, x [256,56,56,128] -> V2 [36,50176,128 pairs], the
launch bench.py's `roofline_extract` times: python tools/pmc_wino_input_target.py"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd import _lib

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B, H, Cc = 256, 56, 128
x = torch.relu(torch.randn((B, H, H, Cc), device="cuda")).contiguous()
slot = torch.zeros(1, dtype=torch.float32, device="cuda")
_lib.check(lib.cslam_absmax_dev(C.c_void_p(x.data_ptr()), x.numel(), C.c_void_p(slot.data_ptr()), st))
V = torch.empty((36, B * (H // 4) * (H // 4), Cc), device="cuda")
for _ in range(6):
    _lib.check(lib.cslam_wino4_input_h2_dev(C.c_void_p(x.data_ptr()), B, H, H, Cc, C.c_void_p(slot.data_ptr()), C.c_void_p(V.data_ptr()), st))
torch.cuda.synchronize()
