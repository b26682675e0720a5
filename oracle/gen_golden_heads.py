"""oracle/gen_golden_heads.py -- TEST INFRASTRUCTURE (build container only).

Golden vectors for the descriptor heads, produced by the REFERENCE's own modules
(NetVLADLayer from cslam/vpr/netvlad.py, GeM/L2Norm/Flatten from
cslam/vpr/cosplace_utils/layers.py), sklearn's PCA and Pillow's resize.
torchvision is not installed, so the transform chain netvlad.py:202-208 is replayed with
the calls torchvision makes for a PIL image: img.crop(box) -> img.resize((224,224), BICUBIC)
-> ToTensor (/255) -> Normalize.  Called from gen_golden.py (stubs installed there).
"""
import os

import numpy as np


def gen_heads(out_dir):
    import torch
    from PIL import Image
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import normalize
    from cslam.vpr.netvlad import NetVLADLayer, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD
    from cslam.vpr.cosplace_utils.layers import GeM, L2Norm, Flatten

    g = {}
    torch.manual_seed(0)
    # ---- G3: NetVLAD layer (vladv1: no bias), 64 clusters x 512 channels
    layer = NetVLADLayer(num_clusters=64, dim=512, vladv2=False).eval()
    with torch.no_grad():
        layer.conv.weight.mul_(8.0)      # sharper soft-assignment than the default init
        for name, shape in (("a", (2, 512, 14, 14)), ("b", (1, 512, 7, 9))):
            x = torch.randn(shape)
            y = layer(x.clone())
            g[f"vlad_{name}/x"] = x.numpy().astype(np.float16).astype(np.float32)  # stored compactly
            y = layer(torch.from_numpy(g[f"vlad_{name}/x"]))
            g[f"vlad_{name}/y"] = y.numpy()
        g["vlad/conv_w"] = layer.conv.weight.detach().numpy().reshape(64, 512)
        g["vlad/centroids"] = layer.centroids.detach().numpy()
    # ---- G4: CosPlace aggregation head, Linear 512 -> {512, 64}, GeM p in {3, 2.37}
    for tag, p, dout in (("p3", 3.0, 512), ("p237", 2.37, 64)):
        gem = GeM(p=p)
        lin = torch.nn.Linear(512, dout)
        head = torch.nn.Sequential(L2Norm(), gem, Flatten(), lin, L2Norm()).eval()
        with torch.no_grad():
            x = torch.randn(2, 512, 7, 7).abs_()        # post-ReLU feature maps are non-negative
            x = x.numpy().astype(np.float16).astype(np.float32)
            y = head(torch.from_numpy(x))
        g[f"gem_{tag}/x"] = x
        g[f"gem_{tag}/W"] = lin.weight.detach().numpy()
        g[f"gem_{tag}/b"] = lin.bias.detach().numpy()
        g[f"gem_{tag}/p"] = np.float32(p)
        g[f"gem_{tag}/y"] = y.numpy()
    # ---- G5: sklearn PCA (float32 input like netvlad.py:231-234) + normalize
    rng = np.random.default_rng(5)
    train = (rng.standard_normal((200, 256)) @ rng.standard_normal((256, 256)) * 0.1).astype(np.float32)
    xq = (rng.standard_normal((5, 256)) @ rng.standard_normal((256, 256)) * 0.1).astype(np.float32)
    for whiten in (False, True):
        pca = PCA(n_components=32, whiten=whiten, random_state=0).fit(train)
        y = normalize(pca.transform(xq))
        t = "w" if whiten else "n"
        g[f"pca_{t}/components"] = pca.components_
        g[f"pca_{t}/mean"] = pca.mean_
        g[f"pca_{t}/var"] = pca.explained_variance_
        g[f"pca_{t}/x"] = xq
        g[f"pca_{t}/y"] = y
    # ---- G6: image transform on two seeded 480x640 RGB frames (BASELINE.md recipe: seed 7 + i)
    mean = np.asarray(IMAGENET_DEFAULT_MEAN, dtype=np.float32)
    std = np.asarray(IMAGENET_DEFAULT_STD, dtype=np.float32)
    for i in range(2):
        img = np.random.default_rng(7 + i).integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
        if i == 1:   # smooth structure + hard edges exercise the negative bicubic lobes / clipping
            yy, xx = np.mgrid[0:480, 0:640]
            img = np.stack([(xx * 255 // 639), (yy * 255 // 479), ((xx // 40 + yy // 40) % 2) * 255],
                           axis=2).astype(np.uint8)
        crop = 376
        top, left = int(round((480 - crop) / 2.0)), int(round((640 - crop) / 2.0))
        pil = Image.fromarray(img).crop((left, top, left + crop, top + crop))
        pil = pil.resize((224, 224), Image.BICUBIC)
        r = np.asarray(pil)
        t = torch.from_numpy(r).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        t = (t - torch.from_numpy(mean)[:, None, None]) / torch.from_numpy(std)[:, None, None]
        g[f"prep_{i}/resized_u8"] = r
        g[f"prep_{i}/out"] = t.numpy()
        g[f"prep_{i}/seed"] = np.int64(7 + i)
    np.savez_compressed(os.path.join(out_dir, "heads_g.npz"), **g)
    print("  heads golden:", {k: v.shape for k, v in g.items() if hasattr(v, "shape") and v.ndim > 0 and "/y" in k})
