"""Interleaved A/B of the VGG-16 trunk with the conv2_1 -> conv2_2 map in pair format (winograd.VGG_PAIRS) and in float32:
ms per 256-frame trunk pass on one stream, and the two kernels alone: python tools/perf_vgg_pairs_ab.py [frames]."""
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr import winograd as wg
from cslam_amd.vpr.backbones import vgg16_features_trunk

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
enc = vgg16_features_trunk().cuda().eval()
x = torch.rand((B, 3, 224, 224), device="cuda") * 4.6 - 2.2
t = wg.WinogradTrunk(enc, 64, 4)
t.input_bound = 2.7


def timed(fn, n=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for rep in range(4):
    res = {}
    for pairs in (True, False):
        wg.VGG_PAIRS = pairs
        res[pairs] = timed(lambda: t(x))
    print(f"trunk pass of {B} frames: pairs between conv2_1 and conv2_2 {res[True]:.3f} ms, float32 {res[False]:.3f} ms", flush=True)
wg.VGG_PAIRS = False
# the two kernels alone
xs = torch.relu(torch.randn((B, 64, 112, 112), device="cuda")).contiguous(memory_format=torch.channels_last)
c1 = torch.randn((128, 64, 3, 3), device="cuda") / 24
c2 = torch.randn((128, 128, 3, 3), device="cuda") / 34
b = torch.randn(128, device="cuda") * 0.1
Wr, Wd = wg.direct_r_pair_weights(c1), wg.direct_pair_weights(c2)
sl = torch.zeros(4, device="cuda")
sl[0] = xs.abs().max()
wl1 = float(c1.abs().sum(dim=(1, 2, 3)).max())
xp = wg.conv3x3_direct_r_pairs(xs, Wr, b, wl1, 0.1, sl[0:1], sl[1:2])
yf = wg.conv3x3_direct_r(xs, Wr, b, True, False, sl[0:1], sl[2:3])
for rep in range(3):
    a = timed(lambda: wg.conv3x3_direct_r_pairs(xs, Wr, b, wl1, 0.1, sl[0:1], sl[1:2]))
    c = timed(lambda: wg.conv3x3_direct_r(xs, Wr, b, True, False, sl[0:1], None))
    d = timed(lambda: wg.conv3x3_direct_hp(xp, (B, 128, 112, 112), sl[1:2], Wd, b, True, True, None))
    e = timed(lambda: wg.conv3x3_direct_h(yf, Wd, b, True, True, sl[2:3], None))
    print(f"conv2_1 pairs out {a:.3f} ms / float32 out {c:.3f} ms;  conv2_2 pairs in {d:.3f} ms / float32 in {e:.3f} ms", flush=True)
