"""The C2 extract alone (CosPlace ResNet-18, 1000 synthetic 640x480 frames per pass, one stream) for kernel traces:
python tools/c2_extract_only.py [passes]"""
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr.cosplace import CosPlace
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cp = CosPlace({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.cosplace.descriptor_dim": 512,
               "frontend.cosplace.backbone": "resnet18"}, None)
g = torch.Generator(device="cuda").manual_seed(7)
frames = torch.randint(0, 256, (1000, 480, 640, 3), generator=g, device="cuda", dtype=torch.uint8)
for _ in range(n):
    cp.compute_embeddings_device(frames)
torch.cuda.synchronize()
