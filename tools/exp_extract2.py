"""GPU experiment: MIOpen find modes for the fp32 NHWC VGG-16 trunk."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr.backbones import vgg16_features_trunk
torch.backends.cudnn.benchmark = True
m = vgg16_features_trunk().cuda().eval().to(memory_format=torch.channels_last)
x = torch.randn(128, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last)
with torch.no_grad():
    t0 = time.perf_counter()
    for _ in range(2): m(x)
    torch.cuda.synchronize(); tw = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(4): m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
print(f"MIOPEN_FIND_MODE={os.environ.get('MIOPEN_FIND_MODE')} ENFORCE={os.environ.get('MIOPEN_FIND_ENFORCE')}: "
      f"{128/dt:.0f} frames/s ({128*30.7e9/dt/1e12:.1f} TFLOP/s), warmup {tw:.1f}s", flush=True)
