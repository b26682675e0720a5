"""conv2_1 / conv2_2 of the VGG-16 trunk at the bench's chunk: the direct one-kernel convolution on fp16 pairs (csrc/conv_direct_h.hip,
the default) against round 3's F(4x4) forms (forms conv_direct = 0), interleaved rounds, HIP-event timed.
    python tools/perf_direct_conv.py [frames=256]"""
import os, sys, statistics
import torch
from torch import nn
sys.path.insert(0, ".")
from cslam_amd.vpr.winograd import WinogradTrunk

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(3)
layers = {"conv2_1 (64 -> 128, 112 x 112)": (64, False), "conv2_2 (128 -> 128, 112 x 112, + pool)": (128, True)}
for name, (cin, pool) in layers.items():
    seq = nn.Sequential(*([nn.Conv2d(cin, 128, 3, padding=1), nn.ReLU()] + ([nn.MaxPool2d(2, 2)] if pool else []))).cuda().eval()
    x = torch.relu(torch.randn((B, cin, 112, 112), device="cuda")).contiguous(memory_format=torch.channels_last)
    runners = {}
    for tag, env in (("direct", 1), ("wino F(4x4)", 0)):
        runners[tag] = WinogradTrunk(seq, 64, 4, fused64=True, forms={"conv_direct": env})
        if env == 1:
            runners[tag].direct_cins = (64, 128)
            runners[tag].refresh()
    res = {t: [] for t in runners}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    outs = {}
    for rnd in range(4):
        for tag, r in runners.items():
            e0.record()
            for _ in range(3):
                y = r(x)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[tag].append(e0.elapsed_time(e1) / 3)
            outs[tag] = y
    flop = 2.0 * B * 112 * 112 * 9 * cin * 128
    d = (outs["direct"] - outs["wino F(4x4)"]).abs().max().item() / outs["direct"].abs().max().item()
    for tag in runners:
        ms = statistics.median(res[tag])
        print(f"{name} x {B} frames, {tag:12s}: {ms:6.3f} ms  ({flop/ms/1e9:7.1f} TFLOP/s of direct-form flop; the direct kernel issues 3 x that in fp16)", flush=True)
    print(f"   max |direct - wino| / max |y| = {d:.2e}", flush=True)
