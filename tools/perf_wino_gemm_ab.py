#!/usr/bin/env python
"""The pair GEMM of the VGG-16 trunk layers from TWO builds of the library in one process, interleaved rounds, results compared bit
for bit:     python tools/perf_wino_gemm_ab.py cslam_amd/libcslam_hip_wgold.so [frames=256]
(the first argument is the library to compare the in-tree one with; build it from another revision of csrc/wino_gemm.hip)."""
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


def main():
    other = C.CDLL(os.path.abspath(sys.argv[1]))
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    lib = _lib.load()
    fn_b = other.cslam_wino_gemm_h2_dev
    fn_b.argtypes = lib.cslam_wino_gemm_h2_dev.argtypes
    fn_b.restype = C.c_int
    fns = {"in-tree": lib.cslam_wino_gemm_h2_dev, "other": fn_b}
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    total = {k: 0.0 for k in fns}
    for name, hw, cin, cout in LAYERS:
        torch.manual_seed(1)
        x = torch.relu(torch.randn((B, cin, hw, hw), device="cuda")).contiguous(memory_format=torch.channels_last)
        w = torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5)
        U2 = wg.split16_pair_weights(wg.wino_weights(w, 4).cuda())
        T = B * -(-hw // 4) * -(-hw // 4)
        slot = torch.zeros(1, dtype=torch.float32, device="cuda")
        _lib.check(lib.cslam_absmax_dev(p(x), x.numel(), p(slot), st))
        V2 = torch.empty((36, T, cin), device="cuda")
        _lib.check(lib.cslam_wino4_input_h2_dev(p(x), B, hw, hw, cin, p(slot), p(V2), st))
        M = {k: torch.empty((36, T, cout), device="cuda") for k in fns}
        res = {k: [] for k in fns}
        for rnd in range(6):
            for k, fn in fns.items():
                e0.record()
                for _ in range(3):
                    _lib.check(fn(p(V2), p(U2[0]), T, cin, cout, p(M[k]), st))
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    res[k].append(e0.elapsed_time(e1) / 3)
        same = bool(torch.equal(M["in-tree"], M["other"]))
        flop = 3 * 2.0 * 36 * T * cin * cout
        line = f"{name:8s} T={T:7d} {cin:3d}->{cout:3d}:"
        for k in fns:
            t = statistics.median(res[k])
            total[k] += t
            line += f"  {k} {t:6.3f} ms ({flop / t / 1e9:6.0f} TF fp16)"
        print(line + f"  bit-identical {same}", flush=True)
    print("sum: " + "  ".join(f"{k} {v:.3f} ms" for k, v in total.items()))


if __name__ == "__main__":
    main()
