"""GPU experiment: VGG-16 trunk throughput variants (PyTorch-ROCm / MIOpen), fp32 unless noted."""
import sys, time
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr.backbones import vgg16_features_trunk

def run(tag, model, x, n=3, ctx=None):
    with torch.no_grad():
        for _ in range(2):
            (model(x) if ctx is None else ctx(model, x))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            (model(x) if ctx is None else ctx(model, x))
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{tag}: {x.shape[0] / dt:.0f} frames/s ({x.shape[0] * 30.7e9 / dt / 1e12:.1f} TFLOP/s)", flush=True)

torch.backends.cudnn.benchmark = True
m = vgg16_features_trunk().cuda().eval()
for B in (32, 64, 128):
    x = torch.randn(B, 3, 224, 224, device="cuda")
    run(f"nchw fp32 B={B}", m, x)
mcl = vgg16_features_trunk().cuda().eval().to(memory_format=torch.channels_last)
for B in (64, 128):
    x = torch.randn(B, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last)
    run(f"nhwc fp32 B={B}", mcl, x)
def ac(model, x):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return model(x)
x = torch.randn(128, 3, 224, 224, device="cuda")
run("nchw bf16-autocast B=128", m, x, ctx=ac)
x = x.to(memory_format=torch.channels_last)
run("nhwc bf16-autocast B=128", mcl, x, ctx=ac)
try:
    cm = torch.compile(m)
    x = torch.randn(64, 3, 224, 224, device="cuda")
    run("nchw fp32 torch.compile B=64", cm, x)
except Exception as e:
    print("compile failed", str(e)[:200])
