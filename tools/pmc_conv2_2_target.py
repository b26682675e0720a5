"""PMC / trace target: VGG-16 conv2_2 (128 -> 128 @112 x 112, + ReLU + MaxPool2d) on 256 frames through the register-resident kernel on
output-channel halves (csrc/conv_direct_r.hip: conv3x3_direct_r2_kernel): python tools/pmc_conv2_2_target.py [launches]"""
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr import winograd as wg

x = torch.randn((256, 128, 112, 112), device="cuda").relu().contiguous(memory_format=torch.channels_last)
w = torch.randn((128, 128, 3, 3), device="cuda") / 34
bias = torch.randn(128, device="cuda") * 0.1
Wr2 = wg.direct_r2_pair_weights(w)
slot = torch.full((1,), float(x.abs().max()), device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    wg.conv3x3_direct_r2(x, Wr2, bias, True, True, slot)
torch.cuda.synchronize()
