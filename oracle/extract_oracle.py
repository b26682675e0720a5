"""CPU restatement of the reference's NetVLAD extract step -- TEST INFRASTRUCTURE / cpu_baseline only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does.

It follows the reference's no-CUDA path (cslam/vpr/netvlad.py:157-160 picks torch.device("cpu") when no CUDA device is
present), one frame at a time like the reference (batch 1, netvlad.py:212-241):
    PIL CenterCrop + bicubic Resize + ToTensor + Normalize   netvlad.py:202-208,223-226  (heads_oracle.preprocess)
    VGG-16 features[:-2] on torch CPU (fp32)                 netvlad.py:163-171,227
    NetVLADLayer.forward with its 64-iteration cluster loop  netvlad.py:94-130,228
    sklearn PCA.transform + normalize                        netvlad.py:231-237          (heads_oracle.pca_transform_normalize)
The weights are DATA handed in by the caller (the same arrays the GPU extractor was given), so the two can be compared.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import heads_oracle

_VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]


def vgg16_encoder(x, conv_params):
    """torchvision vgg16().features[:-2] (netvlad.py:163-171): 13 convolutions, ReLU after all but the last,
    MaxPool2d(2, 2) after conv1_2, conv2_2, conv3_3, conv4_3.  conv_params: list of 13 (weight [Cout,Cin,3,3], bias)."""
    it = iter(conv_params)
    n_conv = 0
    for v in _VGG16_CFG:
        if v == "M":
            x = F.max_pool2d(x, kernel_size=2, stride=2)
        else:
            w, b = next(it)
            x = F.conv2d(x, w, b, padding=1)
            n_conv += 1
            if n_conv < 13:
                x = F.relu(x)
    return x


def netvlad_layer_forward(x, conv_weight, centroids, conv_bias=None):
    """NetVLADLayer.forward (netvlad.py:94-130), including the per-cluster Python loop of :115-124."""
    N, C = x.shape[:2]
    K = centroids.shape[0]
    x = F.normalize(x, p=2, dim=1)                                    # :105-106
    soft_assign = F.conv2d(x, conv_weight.view(K, C, 1, 1), conv_bias).view(N, K, -1)   # :109
    soft_assign = F.softmax(soft_assign, dim=1)                       # :110
    x_flatten = x.view(N, C, -1)
    vlad = torch.zeros([N, K, C], dtype=x.dtype, layout=x.layout, device=x.device)       # :115
    for Cidx in range(K):                                             # :119-124
        residual = x_flatten.unsqueeze(0).permute(1, 0, 2, 3) - \
            centroids[Cidx:Cidx + 1, :].expand(x_flatten.size(-1), -1, -1).permute(1, 2, 0).unsqueeze(0)
        residual *= soft_assign[:, Cidx:Cidx + 1, :].unsqueeze(2)
        vlad[:, Cidx:Cidx + 1, :] = residual.sum(dim=-1)
    vlad = F.normalize(vlad, p=2, dim=2)                              # :126
    vlad = vlad.view(x.size(0), -1)
    return F.normalize(vlad, p=2, dim=1)                              # :127-128


def netvlad_embed(frame_u8, crop, conv_params, vlad_conv_weight, vlad_centroids, pca_components, pca_mean,
                  mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """NetVLAD.compute_embedding (netvlad.py:212-241) for ONE [H,W,3] uint8 frame -> [Dout] float32."""
    t = heads_oracle.preprocess(np.asarray(frame_u8), crop, 224, mean, std)
    with torch.no_grad():
        x = torch.from_numpy(t).unsqueeze(0)
        enc = vgg16_encoder(x, conv_params)
        v = netvlad_layer_forward(enc, vlad_conv_weight, vlad_centroids)
    out = heads_oracle.pca_transform_normalize(v.numpy(), pca_components, pca_mean)
    return out[0]
