"""Z-form GEMM on conv2_2's shape: ring depth / workgroups per CU (CSLAM_ZGEMM_NS, CSLAM_ZGEMM_WGS), interleaved."""
import ctypes as C, os, statistics, sys
sys.path.insert(0, ".")
import torch
from cslam_amd import _lib
from cslam_amd.vpr import winograd as wg
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
p = lambda t: C.c_void_p(t.data_ptr())
B, hw, cin, cout = 256, 112, 128, 128
T = B * (hw // 4) ** 2
U2 = wg.split16_pair_weights(wg.wino_weights(torch.randn((cout, cin, 3, 3), device="cuda") / 30, 4).cuda())
V2 = (torch.randn((36 * T * 2 * cin,), device="cuda") * 100).to(torch.float16)
M = torch.empty((36, T, cout), device="cuda")
def timed(fn, n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cfgs = [("plain 36-plane gemm", None, None), ("z ns3 wg1", "3", "1"), ("z ns2 wg1", "2", "1"), ("z ns2 wg2", "2", "2")]
res = {c[0]: [] for c in cfgs}
for rnd in range(4):
    for name, ns, wgs in cfgs:
        if ns is None:
            t = timed(lambda: _lib.check(lib.cslam_wino_gemm_h2_dev(p(V2), p(U2[0]), T, cin, cout, p(M), st)))
        else:
            os.environ["CSLAM_ZGEMM_NS"], os.environ["CSLAM_ZGEMM_WGS"] = ns, wgs
            t = timed(lambda: _lib.check(lib.cslam_wino_zgemm_h2_dev(p(V2), p(U2[0]), T, cin, cout, p(M), st)))
        if rnd: res[name].append(t)
for name, _, _ in cfgs:
    print(f"{name:22s} {statistics.median(res[name]):.3f} ms", flush=True)
