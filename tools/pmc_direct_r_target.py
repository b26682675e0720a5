"""rocprofv3 target: the register-resident direct convolution (csrc/conv_direct_r.hip) on conv2_1's shape.
    python tools/pmc_direct_r_target.py [frames=256]"""
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd import _lib
from cslam_amd.vpr import winograd as wg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(3)
x = torch.relu(torch.randn((B, 64, 112, 112), device="cuda")).contiguous(memory_format=torch.channels_last)
w = torch.randn(128, 64, 3, 3, device="cuda") / 24.0
b = torch.randn(128, device="cuda")
Wr = wg.direct_r_pair_weights(w)
slot = torch.zeros(1, dtype=torch.float32, device="cuda")
_lib.check(_lib.load().cslam_absmax_dev(x.data_ptr(), x.numel(), slot.data_ptr(), torch.cuda.current_stream().cuda_stream))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(4):
    e0.record()
    y = wg.conv3x3_direct_r(x, Wr, b, True, False, slot, None)
    e1.record()
    torch.cuda.synchronize()
    print("frames", B, "kernel ms", round(e0.elapsed_time(e1), 3), "fp16 TFLOP/s issued", round(3 * 2.0 * B * 112 * 112 * 9 * 64 * 128 / e0.elapsed_time(e1) / 1e9, 1))
