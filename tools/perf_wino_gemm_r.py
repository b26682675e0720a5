#!/usr/bin/env python
"""The split-fp16 Winograd GEMM with the weight fragments in registers (`cslam_wino_gemm_h2r_dev`) against the LDS form
(`cslam_wino_gemm_h2_dev`, its default 256 x 128 ring and the 256 x 256 double buffer), interleaved rounds, median; results
compared element by element (same products, same K order: bit-identical is expected).
    python tools/perf_wino_gemm_r.py [frames=256]"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cslam_amd import _lib  # noqa: E402
from cslam_amd.vpr import winograd as wg  # noqa: E402

LAYERS = [("conv3_1", 56, 128, 256), ("conv3_2", 56, 256, 256), ("conv4_1", 28, 256, 512), ("conv4_2", 28, 512, 512),
          ("conv5_1", 14, 512, 512)]


def p(t):
    return C.c_void_p(t.data_ptr())


def timed(fn, n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for name, hw, cin, cout in LAYERS:
        torch.manual_seed(1)
        w = torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5)
        U4 = wg.wino_weights(w, 4).cuda()
        U2, U2r = wg.split16_pair_weights(U4), wg.split16_pair_weights_r(U4)
        T = B * -(-hw // 4) ** 2
        v2 = (torch.randn((36 * T * 2 * cin,), device="cuda") * 100.0).to(torch.float16)
        M1, M2 = torch.empty((36, T, cout), device="cuda"), torch.empty((36, T, cout), device="cuda")
        forms = {"lds 256x128 ring": lambda: _lib.check(lib.cslam_wino_gemm_h2_dev(p(v2), p(U2[0]), T, cin, cout, p(M1), st)),
                 "registers 256x256": lambda: _lib.check(lib.cslam_wino_gemm_h2r_dev(p(v2), p(U2r[0]), T, cin, cout, p(M2), st))}
        for f in forms.values():
            f()
        torch.cuda.synchronize()
        same = bool(torch.equal(M1, M2))
        diff = float((M1 - M2).abs().max() / M1.abs().max())
        ts = {k: [] for k in forms}
        for _ in range(5):
            for k, f in forms.items():
                ts[k].append(timed(f))
        fl16 = 3 * 2.0 * 36 * T * cin * cout
        print(f"{name:8s} T={T:6d} {cin:3d}->{cout:3d} | " + " | ".join(
            f"{k}: {statistics.median(v):.3f} ms = {fl16 / statistics.median(v) / 1e9:5.0f} TF16" for k, v in ts.items())
            + f" | identical {same} (max rel diff {diff:.1e})")
        del v2, M1, M2


if __name__ == "__main__":
    main()
