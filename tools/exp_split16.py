"""Split-fp16 GEMMs in the F(4x4) Winograd layers (CSLAM_WINO_SPLIT16, vpr/winograd.py `split16_weights`):
(1) one layer against a float64 convolution, beside the fp32 three-kernel form; (2) NetVLAD VGG-16 extraction rate and
descriptor agreement per setting of the channel threshold.  Run on the GPU box."""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cslam_amd.vpr import winograd as wg   # noqa: E402

dev = "cuda"
torch.manual_seed(0)


def layer(B, H, W, Cin, Cout, relu, pool, amp, res=False):
    x = (torch.relu(torch.randn(B, Cin, H, W, device=dev)) * amp).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (3 * Cin ** 0.5)
    b = torch.randn(Cout, device=dev) * amp
    r = (torch.randn(B, Cout, H, W, device=dev) * amp).contiguous(memory_format=torch.channels_last) if res else None
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    if res:
        ref = ref + r.double()
    if relu:
        ref = torch.relu(ref)
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2, 2)
    ws = wg._Workspace()
    U, U4 = wg.wino_weights(w).to(dev), wg.wino_weights(w, 4).to(dev)
    U3 = wg.split16_weights(U4)
    y32 = wg.wino_conv3x3(ws, x, U, U4, b, relu, pool, r)
    y16 = wg.wino_conv3x3(ws, x, U, U4, b, relu, pool, r, U3=U3)
    s = float(ref.abs().max())
    e32, e16 = float((y32.double() - ref).abs().max()) / s, float((y16.double() - ref).abs().max()) / s
    print(f"layer B={B} {H}x{W} {Cin}->{Cout} relu={relu} pool={pool} res={res} amp={amp:g}: fp32 form {e32:.2e}  split16 {e16:.2e}",
          flush=True)
    return e32, e16


if "--layers" in sys.argv or len(sys.argv) == 1:
    for args in [(8, 56, 56, 128, 256, True, False, 1.0), (8, 56, 56, 128, 256, True, True, 1e3), (32, 14, 14, 512, 512, True, False, 1e-3),
                 (16, 28, 28, 256, 512, False, False, 1.0), (8, 30, 22, 64, 64, True, False, 1.0)]:
        layer(*args)
    layer(8, 56, 56, 64, 64, True, False, 1.0, res=True)

if "--trunk" in sys.argv or len(sys.argv) == 1:
    from cslam_amd.vpr.netvlad import NetVLAD
    frames = torch.randint(0, 256, (256, 480, 640, 3), device=dev, dtype=torch.uint8,
                           generator=torch.Generator(device=dev).manual_seed(7))
    base = None
    thrs = [int(a.split('=')[1]) for a in sys.argv if a.startswith('--thr=')] or [0, 512, 256, 128]
    for thr in thrs:
        os.environ["CSLAM_WINO_SPLIT16"] = str(thr)
        ex = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096,
                      "frontend.random_seed": 0, "frontend.backbone_conv": "winograd"}, None)
        d = ex.compute_embeddings_device(frames, None)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            d = ex.compute_embeddings_device(frames, None)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        if base is None:
            base = d.clone()
        print(f"CSLAM_WINO_SPLIT16={thr}: {min(ts) * 1e3:.2f} ms per 256 frames = {256 / min(ts):.0f} frames/s; "
              f"max |descriptor - fp32 form| {float((d - base).abs().max()):.2e}", flush=True)
        del ex
