"""Per-layer timing of the implicit-GEMM convolution on fp16 pairs (csrc/conv_igemm.hip) on the shapes of ResNet-18 at 224 x 224,
B frames: python tools/perf_conv_igemm.py [B].  TF = fp16 flop issued (3 products per multiply-add) per second."""
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr import winograd as wg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
shapes = [("stem 7x7/2 3->64 @224", 3, 64, 224, 7, 2, 3), ("layer1 3x3 64->64 @56", 64, 64, 56, 3, 1, 1),
          ("layer2.0 3x3/2 64->128 @56", 64, 128, 56, 3, 2, 1), ("layer2.0 1x1/2 64->128 @56", 64, 128, 56, 1, 2, 0),
          ("layer2 3x3 128->128 @28", 128, 128, 28, 3, 1, 1), ("layer3.0 3x3/2 128->256 @28", 128, 256, 28, 3, 2, 1),
          ("layer3 3x3 256->256 @14", 256, 256, 14, 3, 1, 1), ("layer4.0 3x3/2 256->512 @14", 256, 512, 14, 3, 2, 1),
          ("layer4 3x3 512->512 @7", 512, 512, 7, 3, 1, 1)]
ws = wg._Workspace()
torch.zeros(1 << 28, device="cuda").sum().item()                         # (a fresh box: first touch of the device, clocks)


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, cin, cout, hw, k, s, p in shapes:
    x = torch.randn((B, cin, hw, hw), device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn((cout, cin, k, k), device="cuda") / (k * cin ** 0.5)
    Wg = wg.igemm_pair_weights(w)
    slot = torch.full((1,), float(x.abs().max()), device="cuda")
    ms = timed(lambda: wg.conv_igemm(ws, x, Wg, None, (k, k), s, p, True, amax_in=slot))
    ho = (hw + 2 * p - k) // s + 1
    kk = (7 * 32) if cin == 3 else k * k * cin
    fl = 2.0 * 3 * B * ho * ho * cout * kk
    line = f"{name:32s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF issued  ({B * ho * ho} pixels, K = {kk})"
    if cin % 32 == 0:
        # the pair-format form: x as pairs (made by a 1x1 layer with pair output), pairs out
        slots = torch.zeros(8, device="cuda")
        slots[0] = slot[0]
        a0 = wg.PairAct(x, False, x.shape, slots[0:1], slots[0:1])
        w1 = torch.eye(cin, device="cuda").reshape(cin, cin, 1, 1).contiguous()
        ap = wg.conv_igemm_p(ws, a0, wg.igemm_pair_weights(w1), None, (1, 1), 1, 0, False, None, 1.0, 0.0, slots[1:2], slots[2:3], True)
        wl1 = float(w.abs().sum(dim=(1, 2, 3)).max())

        def pair_form():
            slots[3].zero_()
            wg.conv_igemm_p(ws, ap, Wg, None, (k, k), s, p, True, None, wl1, 0.0, slots[3:4], slots[4:5], True)
        mp = timed(pair_form)
        line += f"   pair format in / out: {mp:7.3f} ms  {fl / mp / 1e9:7.1f} TF"
    print(line, flush=True)
