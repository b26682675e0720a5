"""Interleaved A/B of the VGG-16 trunk with the Z form of the pair products (24 planes of M instead of 36: csrc/wino_gemm.hip
`wino_zgemm_h2_kernel` + `wino4_output_z_kernel`) on more layers than conv2_2's shape: winograd.Z_FORM_MAX = 128 * 128 (default: no layer of
the default trunk, conv2_2 runs the direct kernel), 256 * 256 (conv3_1 .. conv3_3, the HBM-bound products), 512 * 512 (every Winograd layer).
ms per 256-frame trunk pass on one stream: python tools/perf_zform_trunk_ab.py [frames]"""
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


def timed(fn, n=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = None
for rep in range(4):
    res = {}
    for zmax in (128 * 128, 256 * 256, 512 * 512):
        wg.Z_FORM_MAX = zmax
        res[zmax] = timed(lambda: t(x))
        y = t(x)
        if ref is None:
            ref = y
        elif rep == 0:
            print("Z_FORM_MAX %d: max |y - y_default| / max |y| = %.1e" % (zmax, float((y - ref).abs().max() / ref.abs().max())))
    print("trunk pass of %d frames: Z form nowhere %.3f ms | on conv3_x %.3f ms | on conv3_x .. conv5_x %.3f ms" % (B, res[128 * 128], res[256 * 256], res[512 * 512]), flush=True)
wg.Z_FORM_MAX = 128 * 128
