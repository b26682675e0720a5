#!/usr/bin/env python
"""Per-layer timing of the 36 Winograd-domain products of the VGG-16 trunk at the 256-frame chunk (MI355X):
   pair  this library's split-fp16 GEMM on hi/lo pairs (csrc/wino_gemm.hip) + its input transform (wino4_input_h2)
   h3    round 1: hipBLASLt fp16 GEMM over [vh | vl | vh] + wino4_input_h3
   f32   rocBLAS sgemm + wino4_input
Prints ms per kernel, fp32-equivalent TFLOP/s (2 T Cin Cout 36 flop), fp16 TFLOP/s (x3) and the GEMM's algorithmic
HBM bytes (V in + M out) per second.      python tools/perf_wino_gemm.py [frames=256]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cslam_amd import _lib  # noqa: E402
from cslam_amd.vpr import winograd as wg  # noqa: E402

LAYERS = [("conv2_2", 112, 128, 128), ("conv3_1", 56, 128, 256), ("conv3_2", 56, 256, 256), ("conv4_1", 28, 256, 512),
          ("conv4_2", 28, 512, 512), ("conv5_1", 14, 512, 512)]


def p(t):
    return C.c_void_p(t.data_ptr())


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
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
    tot = {"pair": 0.0, "h3": 0.0, "f32": 0.0}
    print(f"B = {B} frames; CSLAM_WGEMM_DBG = {os.environ.get('CSLAM_WGEMM_DBG', '0')}; pair GEMM shapes CSLAM_WGEMM_CFG: "
          f"1 = 256x256 double buffer, 2 = 256x128 ring of 3, 3 = 128x256 ring of 3, 4 = 256x128 double buffer, 5 = 256x128 ring of 3 with 64x64 wave tiles, 6 = 256x256 on four waves of 512 registers (128x128 wave tiles)")
    for name, hw, cin, cout in LAYERS:
        torch.manual_seed(1)
        x = torch.relu(torch.randn((B, cin, hw, hw), device="cuda")).contiguous(memory_format=torch.channels_last)
        w = torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5)
        U4 = wg.wino_weights(w, 4).cuda()
        T = B * -(-hw // 4) * -(-hw // 4)
        slot = torch.zeros(1, dtype=torch.float32, device="cuda")
        _lib.check(lib.cslam_absmax_dev(p(x), x.numel(), p(slot), st))
        U2 = wg.split16_pair_weights(U4)
        U3 = wg.split16_weights(U4)
        V2 = torch.empty((36, T, cin), device="cuda")
        V3 = torch.empty((36, T, 3 * cin), dtype=torch.float16, device="cuda")
        V = torch.empty((36, T, cin), device="cuda")
        M = torch.empty((36, T, cout), device="cuda")
        t_in2 = timed(lambda: _lib.check(lib.cslam_wino4_input_h2_dev(p(x), B, hw, hw, cin, p(slot), p(V2), st)))
        # shapes of the pair GEMM: interleaved rounds (the chip's clock drifts over a run), median per shape
        import statistics
        cfgs = [c for c in (1, 2, 3, 4, 5, 6) if not (c in (1, 3, 6) and cout % 256)]
        samples = {c: [] for c in cfgs}
        for _ in range(4):
            for c in cfgs:
                os.environ["CSLAM_WGEMM_CFG"] = str(c)
                samples[c].append(timed(lambda: _lib.check(lib.cslam_wino_gemm_h2_dev(p(V2), p(U2[0]), T, cin, cout, p(M), st)), 3))
        os.environ.pop("CSLAM_WGEMM_CFG", None)
        t_cfg = {c: statistics.median(v) for c, v in samples.items()}
        t_g2 = timed(lambda: _lib.check(lib.cslam_wino_gemm_h2_dev(p(V2), p(U2[0]), T, cin, cout, p(M), st)))
        M2 = M.clone()
        t_in3 = timed(lambda: _lib.check(lib.cslam_wino4_input_h3_dev(p(x), B, hw, hw, cin, p(slot), p(V3), st)))
        t_g3 = timed(lambda: torch.bmm(V3, U3[0], out_dtype=torch.float32))
        M3 = torch.bmm(V3, U3[0], out_dtype=torch.float32)
        t_in1 = timed(lambda: _lib.check(lib.cslam_wino4_input_dev(p(x), B, hw, hw, cin, p(V), st)))
        t_g1 = timed(lambda: torch.bmm(V, U4, out=M))
        agree = float((M2 * U2[1] - M3 * U3[1]).abs().max() / (M3 * U3[1]).abs().max())
        flop = 2.0 * 36 * T * cin * cout
        gbytes = 36.0 * T * (cin + cout) * 4
        print(f"{name:8s} T={T:6d} {cin:3d}->{cout:3d} | pair: in {t_in2:.3f} gemm {t_g2:.3f} ms = {flop / t_g2 / 1e9:6.1f} TF32eq "
              f"({3 * flop / t_g2 / 1e9:6.0f} TF16) {gbytes / t_g2 / 1e6:5.0f} GB/s | h3: in {t_in3:.3f} gemm {t_g3:.3f} | "
              f"f32: in {t_in1:.3f} gemm {t_g1:.3f} | pair vs h3 rel diff {agree:.1e} | pair by shape: "
              + " ".join(f"cfg{c} {t:.3f}" for c, t in t_cfg.items()))
        rep = {"conv3_2": 2, "conv4_2": 2, "conv5_1": 3}.get(name, 1)        # conv3_3, conv4_3, conv5_2/3 have the same shape
        tot["pair"] += rep * (t_in2 + t_g2); tot["h3"] += rep * (t_in3 + t_g3); tot["f32"] += rep * (t_in1 + t_g1)
        del x, V2, V3, V, M, M2, M3
        torch.cuda.empty_cache()
    print("input transform + GEMM over the ten layers conv2_2 ... conv5_3 (ms per %d frames): pair %.2f | h3 %.2f | f32 %.2f"
          % (B, tot["pair"], tot["h3"], tot["f32"]))


if __name__ == "__main__":
    main()
