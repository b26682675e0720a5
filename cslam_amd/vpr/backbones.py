"""CNN backbones of the two extractors in plain PyTorch-ROCm (MIOpen / hipBLASLt).

north_star keeps the conv backbone on PyTorch; torchvision is not required: the layer
lists below reproduce torchvision's module order and parameter names so that the
reference's checkpoints load unchanged:
    VGG-16  `features[:-2]`           cslam/vpr/netvlad.py:163-171, cosplace_utils/network.py:59-63
    ResNet-18/50/101/152 `children()[:-2]`             cosplace_utils/network.py:39-56
"""
import torch
from torch import nn

_VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


def vgg16_features_trunk():
    """torchvision vgg16().features with the last ReLU and MaxPool removed ([:-2])."""
    layers, c_in = [], 3
    for v in _VGG16_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c_in, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            c_in = v
    return nn.Sequential(*layers[:-2])


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, c_in, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(c_in, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, c_in, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(c_in, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


def _make_layer(block, c_in, planes, blocks, stride):
    down = None
    if stride != 1 or c_in != planes * block.expansion:
        down = nn.Sequential(nn.Conv2d(c_in, planes * block.expansion, 1, stride, bias=False),
                             nn.BatchNorm2d(planes * block.expansion))
    layers = [block(c_in, planes, stride, down)]
    c_in = planes * block.expansion
    layers += [block(c_in, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers), c_in


_RESNETS = {"resnet18": (BasicBlock, [2, 2, 2, 2]), "resnet50": (Bottleneck, [3, 4, 6, 3]),
            "resnet101": (Bottleneck, [3, 4, 23, 3]), "resnet152": (Bottleneck, [3, 8, 36, 3])}

CHANNELS_NUM_IN_LAST_CONV = {"resnet18": 512, "resnet50": 2048, "resnet101": 2048, "resnet152": 2048,
                             "vgg16": 512}


def resnet_trunk(name):
    """torchvision resnetXX children()[:-2]: conv1, bn1, relu, maxpool, layer1..layer4."""
    block, cfg = _RESNETS[name]
    mods = [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, 2, 1)]
    c = 64
    for planes, n, stride in zip((64, 128, 256, 512), cfg, (1, 2, 2, 2)):
        layer, c = _make_layer(block, c, planes, n, stride)
        mods.append(layer)
    return nn.Sequential(*mods)


def get_backbone(backbone_name):
    """(trunk, channels of the last conv) -- reference cosplace_utils/network.py:38-68."""
    if backbone_name.startswith("resnet"):
        trunk = resnet_trunk(backbone_name)
    elif backbone_name == "vgg16":
        trunk = vgg16_features_trunk()
    else:
        raise ValueError(f"unknown backbone {backbone_name}")
    return trunk, CHANNELS_NUM_IN_LAST_CONV[backbone_name]
