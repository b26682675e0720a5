"""PMC / trace target: ResNet layer1's 64 -> 64 convolution between pair-format maps through csrc/conv_direct_p.hip, 1000 frames of
56 x 56, pair-format shortcut: python tools/pmc_direct_p_target.py [B] [res: 0 | 1] [launches]."""
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr import winograd as wg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
use_res = (sys.argv[2] if len(sys.argv) > 2 else "1") != "0"
ws = wg._Workspace()
x = torch.randn((B, 64, 56, 56), device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn((64, 64, 3, 3), device="cuda") / 24
bias = torch.randn(64, device="cuda") * 0.1
Wp = wg.stem_direct_pair_weights(w)
slots = torch.zeros(12, device="cuda")
slots[0] = x.abs().max()
a0 = wg.PairAct(x, False, x.shape, slots[0:1], slots[0:1])
w1 = torch.eye(64, device="cuda").reshape(64, 64, 1, 1).contiguous()
ap = wg.conv_igemm_p(ws, a0, wg.igemm_pair_weights(w1), None, (1, 1), 1, 0, False, None, 1.0, 0.0, slots[1:2], slots[2:3], True)
wl1 = float(w.abs().sum(dim=(1, 2, 3)).max())
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 12):
    slots[5].zero_()
    wg.conv3x3_direct_p(ap, Wp, bias, True, ap if use_res else None, wl1, 0.1, slots[5:6], slots[6:7], True)
torch.cuda.synchronize()
