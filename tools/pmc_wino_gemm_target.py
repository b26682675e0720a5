"""rocprofv3 target: the pair GEMM (csrc/wino_gemm.hip) on one trunk layer's shape, default conv4_2 (28 x 28, 512 -> 512, 256 frames).
    python tools/pmc_wino_gemm_target.py [hw=28] [cin=512] [cout=512] [frames=256]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd import _lib
from cslam_amd.vpr import winograd as wg
hw, cin, cout, B = [int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 28), (2, 512), (3, 512), (4, 256))]
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
p = lambda t: C.c_void_p(t.data_ptr())
torch.manual_seed(1)
x = torch.relu(torch.randn((B, cin, hw, hw), device="cuda")).contiguous(memory_format=torch.channels_last)
w = torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5)
U2 = wg.split16_pair_weights(wg.wino_weights(w, 4).cuda())
T = B * -(-hw // 4) * -(-hw // 4)
slot = torch.zeros(1, dtype=torch.float32, device="cuda")
_lib.check(lib.cslam_absmax_dev(p(x), x.numel(), p(slot), st))
V2 = torch.empty((36, T, cin), device="cuda")
_lib.check(lib.cslam_wino4_input_h2_dev(p(x), B, hw, hw, cin, p(slot), p(V2), st))
M = torch.empty((36, T, cout), device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(5):
    e0.record()
    _lib.check(lib.cslam_wino_gemm_h2_dev(p(V2), p(U2[0]), T, cin, cout, p(M), st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"T={T} {cin}->{cout}: {ms:.3f} ms = {3 * 2.0 * 36 * T * cin * cout / ms / 1e9:.0f} TFLOP/s fp16 issued, {36.0 * T * (cin + cout) * 4 / ms / 1e6:.0f} GB/s algorithmic")
