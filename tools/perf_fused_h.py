#!/usr/bin/env python
"""conv1_2 / conv2_1 of VGG-16 at the 256-frame chunk: the one-kernel F(4x4) Winograd convolution on the f32-input MFMA
(csrc/wino_fused.hip) against its fp16-pair form (csrc/wino_fused_h.hip), interleaved rounds in one process (the chip's
clock drifts by several per cent over a run, so only interleaved timings compare), median and minimum per kernel.
    python tools/perf_fused_h.py [frames=256] [rounds=5]
"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cslam_amd import _lib  # noqa: E402
from cslam_amd.vpr import winograd as wg  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def t(fn, n=3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    for name, hw, cout, pool in (("conv1_2", 224, 64, True), ("conv2_1", 112, 128, False)):
        torch.manual_seed(0)
        x = torch.relu(torch.randn((B, 64, hw, hw), device="cuda")).contiguous(memory_format=torch.channels_last)
        w = torch.randn((cout, 64, 3, 3), device="cuda") / 24.0
        b = torch.randn(cout, device="cuda")
        U4 = wg.wino_weights(w, 4).cuda()
        Up = wg.fused64_weights(U4)
        Uh = wg.fused64_pair_weights(U4)
        slot = torch.zeros(1, dtype=torch.float32, device="cuda")
        _lib.check(lib.cslam_absmax_dev(C.c_void_p(x.data_ptr()), x.numel(), C.c_void_p(slot.data_ptr()), st))
        oslot = torch.zeros(1, dtype=torch.float32, device="cuda")
        f32 = lambda: wg.wino_fused64(x, Up, b, True, pool)                      # noqa: E731
        h = lambda: wg.wino_fused64_h(x, Uh, b, True, pool, slot, oslot)          # noqa: E731
        ya, yb = f32(), h()
        torch.cuda.synchronize()
        diff = float((ya - yb).abs().max() / ya.abs().max())
        ta, tb = [], []
        for _ in range(rounds):
            ta.append(t(f32)); tb.append(t(h))
        flop = B * (hw // 4) ** 2 * 36 * 2 * 64 * cout
        nbytes = (x.numel() + ya.numel()) * 4
        for tag, ts in (("f32 MFMA", ta), ("fp16 pairs", tb)):
            med, mn = statistics.median(ts), min(ts)
            print(f"{name} {tag:10s}: median {med:.3f} ms, min {mn:.3f} ms = {flop / med / 1e9:6.1f} TFLOP/s fp32-equivalent, "
                  f"{nbytes / med / 1e6:5.0f} GB/s of algorithmic bytes (in + out)")
        print(f"{name}: fp16-pair form vs f32 form, max |diff| / max |y| = {diff:.1e}")
        del x, ya, yb


if __name__ == "__main__":
    main()
