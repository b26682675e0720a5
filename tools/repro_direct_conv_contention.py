"""Repro of the round-4 race in the direct convolution: the same launch alone and beside a stream that streams 256 MB through the L2
(OTHER=copy | direct | small).  Before the fix (patch loads under a predicate, counted waits) the copy partner gave errors up to 0.7;
kept as a tool, the regression test is tests/test_heads_gpu.py::test_direct_conv_beside_a_stream_that_thrashes_the_l2_is_bit_identical."""
import os, sys, torch
sys.path.insert(0, ".")
from cslam_amd.vpr import winograd as wg
from cslam_amd import _lib
import ctypes as C
lib = _lib.load()
torch.manual_seed(0)
B, Cin, H, W = 12, int(os.environ.get("CIN", "128")), 188, 188
pool = bool(int(os.environ.get("POOL", "1")))
x = torch.relu(torch.randn(B, Cin, H, W, device="cuda")).contiguous(memory_format=torch.channels_last)
w = torch.randn(128, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)
b = torch.randn(128, device="cuda")
Wd = wg.direct_pair_weights(w)
slot = x.abs().max().reshape(1).clone()
ref = wg.conv3x3_direct_h(x, Wd, b, True, pool, slot)
torch.cuda.synchronize()
again = wg.conv3x3_direct_h(x, Wd, b, True, pool, slot)
print("single stream repeat equal:", torch.equal(ref, again))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
mode = os.environ.get("OTHER", "direct")
big = torch.empty(64 << 20, device="cuda")
x2 = x.clone()
for t in range(6):
    with torch.cuda.stream(s2):
        for _ in range(4):
            if mode == "direct":
                o = wg.conv3x3_direct_h(x2, Wd, b, True, pool, slot)
            elif mode == "copy":
                big.add_(1.0)
            elif mode == "small":
                for _ in range(20):
                    big[:1 << 16].add_(1.0)
    with torch.cuda.stream(s1):
        ys = [wg.conv3x3_direct_h(x, Wd, b, True, pool, slot) for _ in range(3)]
    torch.cuda.synchronize()
    for y in ys:
        if not torch.equal(y, ref):
            d = (y - ref).abs()
            idx = torch.nonzero(d > 0)
            imgs = sorted(set(idx[:, 0].tolist()))
            ch = idx[:, 1]; yy = idx[:, 2]; xx = idx[:, 3]
            print(f"try {t}: {idx.shape[0]} values differ, max {float(d.max()):.3e}; images {imgs}; channels {int(ch.min())}..{int(ch.max())} ({len(set(ch.tolist()))} distinct); "
                  f"rows {int(yy.min())}..{int(yy.max())}; cols {int(xx.min())}..{int(xx.max())}")
        else:
            print(f"try {t}: equal")
