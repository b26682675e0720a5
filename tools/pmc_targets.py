#!/usr/bin/env python
"""The launches bench.py prices, in a fixed order, for rocprofv3 --pmc passes (one counter set per run, as the MI355X
guide prescribes; tools/gpu_r2_pmc.sh).  Every target is launched REPS times in a row; tools/pmc_by_kernel.py finds
them in the counter CSV by kernel name and order of appearance:
  wino4_input_h2_kernel   conv2_2 shape      x [256,112,112,128] -> V2 [36,200704,128 pairs]
  wino_gemm_h2_kernel     conv2_2 shape      36 x [200704,128] x [128,128]        (first group of REPS launches)
  wino_gemm_h2_kernel     conv4_2 shape      36 x [12544,512] x [512,512]         (second group)
  wino4_fused_c64_h_kernel<64>  conv1_2      x [256,224,224,64] -> pooled [256,112,112,64]
  wino4_fused_c64_h_kernel<128> conv2_1      x [256,112,112,64] -> [256,112,112,128]
  wino4_fused_c64_h_kernel<64, .., STEM> conv1_1 + conv1_2   x0 [256,3,224,224] -> pooled [256,112,112,64]  (second <64 group)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cslam_amd import _lib  # noqa: E402
from cslam_amd.vpr import winograd as wg  # noqa: E402

REPS = 3
B = 256


def p(t):
    return C.c_void_p(t.data_ptr())


def main():
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    x = torch.relu(torch.randn((B, 128, 112, 112), device="cuda")).contiguous(memory_format=torch.channels_last)
    _lib.check(lib.cslam_absmax_dev(p(x), x.numel(), p(slot), st))
    T = B * 28 * 28
    V2 = torch.empty((36, T, 128), device="cuda")
    for _ in range(REPS):
        _lib.check(lib.cslam_wino4_input_h2_dev(p(x), B, 112, 112, 128, p(slot), p(V2), st))
    for cin, cout, hw in ((128, 128, 112), (512, 512, 28)):
        w = torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5)
        U2 = wg.split16_pair_weights(wg.wino_weights(w, 4).cuda())
        Tl = B * (hw // 4) ** 2
        v2 = (torch.randn((36 * Tl * 2 * cin,), device="cuda") * 100.0).to(torch.float16)
        M = torch.empty((36, Tl, cout), device="cuda")
        for _ in range(REPS):
            _lib.check(lib.cslam_wino_gemm_h2_dev(p(v2), p(U2[0]), Tl, cin, cout, p(M), st))
        del v2, M
    del x, V2
    for cout, hw, pool in ((64, 224, True), (128, 112, False)):
        xf = torch.relu(torch.randn((B, 64, hw, hw), device="cuda")).contiguous(memory_format=torch.channels_last)
        _lib.check(lib.cslam_absmax_dev(p(xf), xf.numel(), p(slot), st))
        w = torch.randn((cout, 64, 3, 3), device="cuda") / 24.0
        Uh = wg.fused64_pair_weights(wg.wino_weights(w, 4).cuda())
        b = torch.randn(cout, device="cuda")
        for _ in range(REPS):
            wg.wino_fused64_h(xf, Uh, b, True, pool, slot, None)
        del xf
    x0 = torch.rand((B, 3, 224, 224), device="cuda") * 4.8 - 2.2
    w1 = torch.randn((64, 3, 3, 3), device="cuda") / 5.0
    b1 = torch.randn(64, device="cuda")
    stem = wg.stem_pair_weights(w1)
    w = torch.randn((64, 64, 3, 3), device="cuda") / 24.0
    Uh = wg.fused64_pair_weights(wg.wino_weights(w, 4).cuda())
    b = torch.randn(64, device="cuda")
    _lib.check(lib.cslam_absmax_dev(p(x0), x0.numel(), p(slot), st))
    for _ in range(REPS):
        wg.wino_stem64_h(x0, stem, b1, Uh, b, True, slot, None)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
