"""Times conv1_2 (64 -> 64, 224 x 224, ReLU + pool, 256 frames) through the ablation builds of the F(4x4) fused kernel."""
import ctypes as C, os, sys, time
import torch
here = os.path.dirname(os.path.abspath(__file__))
B, H, W = 256, 224, 224
x = torch.randn((B, H, W, 64), device="cuda"); Up = torch.randn((4, 36, 4, 4, 16, 4), device="cuda")
bias = torch.randn(64, device="cuda"); y = torch.empty((B, H // 2, W // 2, 64), device="cuda")
vp = C.c_void_p
for m in (0, 32, 64, 96, 128, 160, 1024, 2048):
    lib = C.CDLL(os.path.join(here, "_abl", f"libwf_{m}.so"))
    f = lib.cslam_wino4_fused_c64_dev; f.restype = C.c_int
    f.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert f(x.data_ptr(), Up.data_ptr(), bias.data_ptr(), None, B, H, W, 64, 1, 1, y.data_ptr(), None) == 0
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"WF_ABL={m:2d}: {min(ts[1:])*1e3:.3f} ms")
