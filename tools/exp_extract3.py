"""GPU experiment: cudnn.benchmark on/off for the fp32 NHWC VGG-16 trunk (warm-up time vs throughput)."""
import sys, time
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr.backbones import vgg16_features_trunk
flag = sys.argv[1] == "1"
torch.backends.cudnn.benchmark = flag
m = vgg16_features_trunk().cuda().eval().to(memory_format=torch.channels_last)
x = torch.randn(128, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last)
with torch.no_grad():
    t0 = time.perf_counter(); m(x); torch.cuda.synchronize(); tw = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(4): m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
print(f"cudnn.benchmark={flag}: {128/dt:.0f} frames/s, first call {tw:.1f}s", flush=True)
