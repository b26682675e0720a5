"""Target of tools/pmc_kernel.sh for the ResNet trunk's kernels (csrc/conv_igemm.hip): ONE shape, 6 launches.
    python tools/pmc_igemm_target.py layer1|layer2|layer3|layer4|stem [frames]
layerN: the stride-1 3x3 layer of that stage, pair-format input and output (conv_igemm_h2_kernel<.., 2>); stem: the 7x7 / 2 stem with its
max-pool (conv_stem_pool_patch_kernel)."""
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr import winograd as wg

which = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ws = wg._Workspace()
if which == "stem":
    x = torch.randn((B, 3, 224, 224), device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn((64, 3, 7, 7), device="cuda") / (7 * 3 ** 0.5)
    Wg = wg.igemm_pair_weights(w)
    slot = torch.full((1,), float(x.abs().max()), device="cuda")
    run = lambda: wg.conv_igemm(ws, x, Wg, None, (7, 7), 2, 3, True, amax_in=slot, pool=True)      # noqa: E731
elif which == "layer1f":                           # layer1's first convolution: float32 input (the pooled stem output), pair-format output
    x = torch.randn((B, 64, 56, 56), device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn((64, 64, 3, 3), device="cuda") / (3 * 8.0)
    sl = torch.zeros(8, device="cuda")
    sl[0] = x.abs().max()
    a0 = wg.PairAct(x, False, x.shape, sl[0:1], sl[0:1])
    Wg, wl1 = wg.igemm_pair_weights(w), float(w.abs().sum(dim=(1, 2, 3)).max())
    run = lambda: wg.conv_igemm_p(ws, a0, Wg, None, (3, 3), 1, 1, True, None, wl1, 0.0, sl[3:4], sl[4:5], True)   # noqa: E731
else:
    c, hw = {"layer1": (64, 56), "layer2": (128, 28), "layer3": (256, 14), "layer4": (512, 7)}[which]
    x = torch.randn((B, c, hw, hw), device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn((c, c, 3, 3), device="cuda") / (3 * c ** 0.5)
    sl = torch.zeros(8, device="cuda")
    sl[0] = x.abs().max()
    eye = torch.eye(c, device="cuda").reshape(c, c, 1, 1).contiguous()
    ap = wg.conv_igemm_p(ws, wg.PairAct(x, False, x.shape, sl[0:1], sl[0:1]), wg.igemm_pair_weights(eye), None, (1, 1), 1, 0, False, None,
                         1.0, 0.0, sl[1:2], sl[2:3], True)
    Wg, wl1 = wg.igemm_pair_weights(w), float(w.abs().sum(dim=(1, 2, 3)).max())
    run = lambda: wg.conv_igemm_p(ws, ap, Wg, None, (3, 3), 1, 1, True, None, wl1, 0.0, sl[3:4], sl[4:5], True)   # noqa: E731
for _ in range(6):
    run()
torch.cuda.synchronize()
