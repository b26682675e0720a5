"""rocprofv3 target: the direct stem kernel (csrc/conv_stem_direct_h.hip), VGG-16 conv1_1 + conv1_2 + MaxPool2d at the 256-frame chunk.
    python tools/pmc_stem_direct_target.py [frames=256]"""
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd import _lib
from cslam_amd.vpr import winograd as wg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(3)
x0 = torch.rand((B, 3, 224, 224), device="cuda") * 4.8 - 2.2
w1 = torch.randn((64, 3, 3, 3), device="cuda") / 5.0
b1 = torch.randn(64, device="cuda")
w2 = torch.randn((64, 64, 3, 3), device="cuda") / 24.0
b2 = torch.randn(64, device="cuda")
stem = wg.stem_pair_weights(w1)
Wr = wg.stem_direct_pair_weights(w2)
slot = torch.zeros(1, dtype=torch.float32, device="cuda")
_lib.check(_lib.load().cslam_absmax_dev(x0.data_ptr(), x0.numel(), slot.data_ptr(), torch.cuda.current_stream().cuda_stream))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
fl = B * 224 * 224 * (3 * 2.0 * 9 * 64 * 64 + (180.0 / 128.0) * 3 * 2.0 * 32 * 64)
for rep in range(4):
    e0.record()
    y = wg.conv_stem_direct_h(x0, stem, b1, Wr, b2, True, slot, None)
    e1.record()
    torch.cuda.synchronize()
    print("frames", B, "kernel ms", round(e0.elapsed_time(e1), 3), "fp16 TFLOP/s issued", round(fl / e0.elapsed_time(e1) / 1e9, 1))
