"""ResNet layer1's 64 -> 64 convolution between pair-format maps: the register-resident direct kernel (csrc/conv_direct_p.hip) against the
implicit GEMM (csrc/conv_igemm.hip), interleaved, with and without a pair-format shortcut: python tools/perf_direct_p.py [B] [H] [W].
TF = fp16 flop issued (3 products per multiply-add) per second."""
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr import winograd as wg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 56
W = int(sys.argv[3]) if len(sys.argv) > 3 else H
ws = wg._Workspace()
torch.zeros(1 << 28, device="cuda").sum().item()


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


x = torch.randn((B, 64, H, W), device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn((64, 64, 3, 3), device="cuda") / 24
bias = torch.randn(64, device="cuda") * 0.1
Wg, Wp = wg.igemm_pair_weights(w), wg.stem_direct_pair_weights(w)
slots = torch.zeros(12, device="cuda")
slots[0] = x.abs().max()
a0 = wg.PairAct(x, False, x.shape, slots[0:1], slots[0:1])
w1 = torch.eye(64, device="cuda").reshape(64, 64, 1, 1).contiguous()
ap = wg.conv_igemm_p(ws, a0, wg.igemm_pair_weights(w1), None, (1, 1), 1, 0, False, None, 1.0, 0.0, slots[1:2], slots[2:3], True)
wl1 = float(w.abs().sum(dim=(1, 2, 3)).max())
fl = 2.0 * 3 * B * H * W * 64 * 576
for res in (None, ap):
    def igemm():
        slots[3].zero_()
        return wg.conv_igemm_p(ws, ap, Wg, bias, (3, 3), 1, 1, True, res, wl1, 0.1, slots[3:4], slots[4:5], True)

    def direct():
        slots[5].zero_()
        return wg.conv3x3_direct_p(ap, Wp, bias, True, res, wl1, 0.1, slots[5:6], slots[6:7], True)
    yi, yd = igemm(), direct()
    d = (wg.pairs_to_float(yi) - wg.pairs_to_float(yd)).abs().max().item() / wg.pairs_to_float(yi).abs().max().item()
    for rep in range(3):
        mi, md = timed(igemm), timed(direct)
        print(f"shortcut {'pairs' if res is not None else 'none ':5s}  implicit GEMM {mi:7.3f} ms {fl / mi / 1e9:7.1f} TF   direct {md:7.3f} ms "
              f"{fl / md / 1e9:7.1f} TF   ({B} x {H} x {W}, max |diff| / max {d:.1e})", flush=True)

# the float32-input layer that opens the chain (pooled stem output -> pairs)
def igemm_f():
    slots[7].zero_()
    return wg.conv_igemm_p(ws, a0, Wg, bias, (3, 3), 1, 1, True, None, wl1, 0.1, slots[7:8], slots[8:9], True)


def direct_f():
    slots[9].zero_()
    return wg.conv3x3_direct_p(a0, Wp, bias, True, None, wl1, 0.1, slots[9:10], slots[10:11], True)


yi, yd = igemm_f(), direct_f()
d = (wg.pairs_to_float(yi) - wg.pairs_to_float(yd)).abs().max().item() / wg.pairs_to_float(yi).abs().max().item()
for rep in range(3):
    mi, md = timed(igemm_f), timed(direct_f)
    print(f"float32 input   implicit GEMM {mi:7.3f} ms {fl / mi / 1e9:7.1f} TF   direct {md:7.3f} ms {fl / md / 1e9:7.1f} TF   (max |diff| / max {d:.1e})", flush=True)
