# A/B timing of two or more standalone builds of csrc/wino_fused.hip (tools/_abl/libwf_*.so, each
# `hipcc --offload-arch=gfx950 -O3 -shared -fPIC wino_fused.hip stub.cpp` with stub.cpp defining cslam_set_error) in ONE
# process, interleaved, so that box-to-box and clock drift cancel.  Results: profiles/r01_exp_fused_ablation.log.
import ctypes as C, os, sys, time, glob
import torch
B, H, W = 256, 224, 224
x = torch.randn((B, H, W, 64), device="cuda"); Up = torch.randn((4, 36, 4, 4, 16, 4), device="cuda")
bias = torch.randn(64, device="cuda"); y = torch.empty((B, H // 2, W // 2, 64), device="cuda")
vp = C.c_void_p
fs = {}
for n in sorted(glob.glob("tools/_abl/libwf_*.so")):
    lib = C.CDLL(n); f = lib.cslam_wino4_fused_c64_dev; f.restype = C.c_int
    f.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]; fs[os.path.basename(n)] = f
for rep in range(4):
    for k, f in fs.items():
        ts = []
        for _ in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            assert f(x.data_ptr(), Up.data_ptr(), bias.data_ptr(), None, B, H, W, 64, 1, 1, y.data_ptr(), None) == 0
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{k}: min {min(ts[1:])*1e3:.3f} median {sorted(ts[1:])[3]*1e3:.3f} ms")
