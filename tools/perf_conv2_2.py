"""VGG-16 conv2_2 (128 -> 128 @112 x 112, + ReLU + MaxPool2d) on B frames: the direct kernel with weights through an LDS ring
(csrc/conv_direct_h.hip) against the register-resident form on output-channel halves (csrc/conv_direct_r.hip: conv3x3_direct_r2_kernel),
interleaved, with the float64 error of both: python tools/perf_conv2_2.py [B] [H] [W]"""
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr import winograd as wg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 112
W = int(sys.argv[3]) if len(sys.argv) > 3 else H
torch.zeros(1 << 28, device="cuda").sum().item()
torch.manual_seed(5)
x = torch.randn((B, 128, H, W), device="cuda").relu().contiguous(memory_format=torch.channels_last)
w = torch.randn((128, 128, 3, 3), device="cuda") / 34
bias = torch.randn(128, device="cuda") * 0.1
Wd, Wr2 = wg.direct_pair_weights(w), wg.direct_r2_pair_weights(w)
slot = torch.full((1,), float(x.abs().max()), device="cuda")
so = torch.zeros(2, device="cuda")


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


yh = wg.conv3x3_direct_h(x, Wd, bias, True, True, slot, so[0:1])
yr = wg.conv3x3_direct_r2(x, Wr2, bias, True, True, slot, so[1:2])
nb = min(B, 4)
ref = torch.nn.functional.max_pool2d(torch.nn.functional.conv2d(x[:nb].double(), w.double(), bias.double(), padding=1).relu(), 2)
eh = ((yh[:nb].double() - ref).abs().max() / ref.abs().max()).item()
er = ((yr[:nb].double() - ref).abs().max() / ref.abs().max()).item()
print(f"max |err| / max |y| against float64: LDS-ring form {eh:.2e}, register-resident halves {er:.2e}; max |y| slots {so.tolist()} (float64 {ref.max().item():.6f} on {nb} frames)")
fl = 2.0 * 3 * B * H * W * 128 * 1152
for rep in range(3):
    mh = timed(lambda: wg.conv3x3_direct_h(x, Wd, bias, True, True, slot))
    mr = timed(lambda: wg.conv3x3_direct_r2(x, Wr2, bias, True, True, slot))
    print(f"conv2_2 {B} x {H} x {W}: LDS-ring form {mh:.3f} ms ({fl / mh / 1e9:.0f} TF issued)   register-resident halves {mr:.3f} ms ({fl / mr / 1e9:.0f} TF)", flush=True)
