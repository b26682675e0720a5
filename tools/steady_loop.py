#!/usr/bin/env python
"""A steady load of ONE kernel for `seconds` (tools/power_trace.py samples the board's sensors beside it):
    python tools/steady_loop.py gemm [seconds=12] [hw=28] [cin=512] [cout=512]    the trunk's pair GEMM (csrc/wino_gemm.hip), 256 frames
    python tools/steady_loop.py match [seconds=12] [nq=16384]                     the candidate stage + re-scoring on a 100k x 4096 bank
    python tools/steady_loop.py stem [seconds=12]                                 the direct stem kernel, 256 frames
    python tools/steady_loop.py conv21 | conv22 | input | copy [seconds=12]       conv2_1 / conv2_2 (direct kernels), the input transform of conv3_2, a 1 GiB copy
    python tools/steady_loop.py peak16 [seconds=12]                               csrc/peaks.hip's register-resident fp16 MFMA loop, non-zero operands
Prints ms per launch over the whole loop (HIP events) and the launches done."""
import ctypes as C
import sys
import time
import torch
sys.path.insert(0, ".")
from cslam_amd import _lib  # noqa: E402
from cslam_amd.vpr import winograd as wg  # noqa: E402

what = sys.argv[1]
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
torch.manual_seed(1)
if what == "gemm":
    hw, cin, cout = [int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((3, 28), (4, 512), (5, 512))]
    B = 256
    x = torch.relu(torch.randn((B, cin, hw, hw), device="cuda")).contiguous(memory_format=torch.channels_last)
    w = torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5)
    U2 = wg.split16_pair_weights(wg.wino_weights(w, 4).cuda())
    T = B * -(-hw // 4) * -(-hw // 4)
    slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    _lib.check(lib.cslam_absmax_dev(p(x), x.numel(), p(slot), st))
    V2 = torch.empty((36, T, cin), device="cuda")
    _lib.check(lib.cslam_wino4_input_h2_dev(p(x), B, hw, hw, cin, p(slot), p(V2), st))
    M = torch.empty((36, T, cout), device="cuda")
    flop = 3 * 2.0 * 36 * T * cin * cout
    fn = lambda: _lib.check(lib.cslam_wino_gemm_h2_dev(p(V2), p(U2[0]), T, cin, cout, p(M), st))  # noqa: E731
elif what == "match":
    from cslam_amd import nns_matching as nnm
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
    gen = torch.Generator(device="cuda").manual_seed(1234)
    bank = torch.randn((100_000, 4096), generator=gen, device="cuda")
    bank /= bank.norm(dim=1, keepdim=True)
    nn = nnm.NearestNeighborsMatching()
    nn.add_items_device(bank)
    q = torch.randn((nq, 4096), generator=gen, device="cuda")
    q /= q.norm(dim=1, keepdim=True)
    flop = 2.0 * nq * 100_000 * 4096
    fn = lambda: nn.search_device(q, 5, mode=nnm.MODE_MFMA)  # noqa: E731
elif what == "stem":
    from torch import nn as tnn
    seq = tnn.Sequential(tnn.Conv2d(3, 64, 3, padding=1), tnn.ReLU(), tnn.Conv2d(64, 64, 3, padding=1), tnn.ReLU(), tnn.MaxPool2d(2, 2)).cuda().eval()
    x = torch.rand((256, 3, 224, 224), device="cuda") * 4.8 - 2.2
    trunk = wg.WinogradTrunk(seq, 64, 4, fused64=True)
    flop = 3 * 2.0 * 256 * 224 * 224 * 9 * 64 * 64
    fn = lambda: trunk(x)  # noqa: E731
elif what in ("conv21", "conv22"):
    from torch import nn as tnn
    cin, pool = (64, False) if what == "conv21" else (128, True)
    seq = tnn.Sequential(*([tnn.Conv2d(cin, 128, 3, padding=1), tnn.ReLU()] + ([tnn.MaxPool2d(2, 2)] if pool else []))).cuda().eval()
    x = torch.relu(torch.randn((256, cin, 112, 112), device="cuda")).contiguous(memory_format=torch.channels_last)
    trunk = wg.WinogradTrunk(seq, 64, 4, fused64=True)
    flop = 3 * 2.0 * 256 * 112 * 112 * 9 * cin * 128
    fn = lambda: trunk(x)  # noqa: E731
elif what in ("input", "output"):
    hw, c = 56, 256
    B = 256
    x = torch.relu(torch.randn((B, c, hw, hw), device="cuda")).contiguous(memory_format=torch.channels_last)
    T = B * (hw // 4) * (hw // 4)
    slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    _lib.check(lib.cslam_absmax_dev(p(x), x.numel(), p(slot), st))
    V2 = torch.empty((36, T, c), device="cuda")
    flop = 0.0
    nbytes = B * hw * hw * c * 4 + 36 * T * c * 4
    fn = lambda: _lib.check(lib.cslam_wino4_input_h2_dev(p(x), B, hw, hw, c, p(slot), p(V2), st))  # noqa: E731
    if what == "output":
        raise SystemExit("output transform: not wired here")
elif what == "copy":
    nb = 1 << 30
    src = torch.empty(nb // 4, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    flop = 0.0
    nbytes = 2 * nb
    fn = lambda: _lib.check(lib.cslam_peak_copy_dev(p(src), p(dst), C.c_int64(nb), 2, st))  # noqa: E731
elif what == "peak16":
    scratch = torch.zeros(16, dtype=torch.float32, device="cuda")
    fl = C.c_double(0.0)
    fn = lambda: _lib.check(lib.cslam_peak_mfma_dev(1, 4096, 1024, 1, p(scratch), C.byref(fl), st))  # noqa: E731
    fn()
    flop = fl.value
else:
    raise SystemExit(__doc__)
fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 0
t_end = time.time() + seconds
e0.record()
while time.time() < t_end:
    for _ in range(8):
        fn()
    n += 8
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
extra = f", {nbytes / ms / 1e6:.0f} GB/s algorithmic" if what in ("input", "copy") else ""
print(f"{what}: {n} launches, {ms:.3f} ms each = {flop / ms / 1e9:.0f} TFLOP/s (fp16 flop issued; match: 2 D flop per pair){extra}")
