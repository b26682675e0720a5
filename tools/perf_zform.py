#!/usr/bin/env python
"""Z form vs 36-plane form of the trunk's pair products, per layer at the 256-frame chunk (GPU box), interleaved:
    plain: cslam_wino_gemm_h2_dev (M, 36 planes) + cslam_wino4_output_scaled_dev
    z:     cslam_wino_zgemm_h2_dev (Z, 24 planes) + cslam_wino4_output_z_dev
python tools/perf_zform.py [frames=256]"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cslam_amd import _lib  # noqa: E402
from cslam_amd.vpr import winograd as wg  # noqa: E402

LAYERS = [("conv2_2", 112, 128, 128, True), ("conv3_1", 56, 128, 256, False), ("conv3_2", 56, 256, 256, False),
          ("conv3_3", 56, 256, 256, True), ("conv4_1", 28, 256, 512, False), ("conv4_2", 28, 512, 512, False), ("conv5_1", 14, 512, 512, False)]


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
    tot = {"plain": 0.0, "z": 0.0}
    for name, hw, cin, cout, pool in LAYERS:
        torch.manual_seed(1)
        x = torch.relu(torch.randn((B, cin, hw, hw), device="cuda")).contiguous(memory_format=torch.channels_last)
        w = torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5)
        bias = torch.randn(cout, device="cuda") * 0.1
        U4 = wg.wino_weights(w, 4).cuda()
        T = B * -(-hw // 4) * -(-hw // 4)
        slot = torch.zeros(1, dtype=torch.float32, device="cuda")
        _lib.check(lib.cslam_absmax_dev(p(x), x.numel(), p(slot), st))
        U2 = wg.split16_pair_weights(U4)
        V2 = torch.empty((36, T, cin), device="cuda")
        M = torch.empty((36, T, cout), device="cuda")
        ho = hw // 2 if pool else hw
        y = torch.empty((B, cout, ho, ho), device="cuda").contiguous(memory_format=torch.channels_last)
        _lib.check(lib.cslam_wino4_input_h2_dev(p(x), B, hw, hw, cin, p(slot), p(V2), st))
        fns = {
            "plain_gemm": lambda: _lib.check(lib.cslam_wino_gemm_h2_dev(p(V2), p(U2[0]), T, cin, cout, p(M), st)),
            "plain_out": lambda: _lib.check(lib.cslam_wino4_output_scaled_dev(p(M), p(bias), None, B, hw, hw, cout, 1, int(pool), p(slot), float(U2[1]), None, p(y), st)),
            "z_gemm": lambda: _lib.check(lib.cslam_wino_zgemm_h2_dev(p(V2), p(U2[0]), T, cin, cout, p(M), st)),
            "z_out": lambda: _lib.check(lib.cslam_wino4_output_z_dev(p(M), p(bias), B, hw, hw, cout, 1, int(pool), p(slot), float(U2[1]), None, p(y), st)),
        }
        samples = {k: [] for k in fns}
        for rnd in range(4):
            for k in ("plain_gemm", "plain_out", "z_gemm", "z_out"):
                t = timed(fns[k])
                if rnd:
                    samples[k].append(t)
        t = {k: statistics.median(v) for k, v in samples.items()}
        tot["plain"] += t["plain_gemm"] + t["plain_out"]
        tot["z"] += t["z_gemm"] + t["z_out"]
        gb_p = (36.0 * T * (cin + cout) * 4) / 1e9
        gb_z = (36.0 * T * cin * 4 + 24.0 * T * cout * 4) / 1e9
        print(f"{name:8s} T={T:6d} {cin:3d}->{cout:3d} | plain: gemm {t['plain_gemm']:.3f} ({gb_p / t['plain_gemm'] * 1e3:.0f} GB/s) out {t['plain_out']:.3f} = "
              f"{t['plain_gemm'] + t['plain_out']:.3f} ms | z: gemm {t['z_gemm']:.3f} ({gb_z / t['z_gemm'] * 1e3:.0f} GB/s) out {t['z_out']:.3f} = {t['z_gemm'] + t['z_out']:.3f} ms", flush=True)
    print(f"sum over these layers (ms per {B} frames): plain {tot['plain']:.2f} | z {tot['z']:.2f}")


if __name__ == "__main__":
    main()
