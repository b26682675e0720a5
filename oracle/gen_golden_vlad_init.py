#!/usr/bin/env python
"""oracle/gen_golden_vlad_init.py -- TEST INFRASTRUCTURE (build container only).

G12: `NetVLADLayer.init_params` (cslam/vpr/netvlad.py:63-92) run by the REFERENCE's own class on
seeded clusters / training descriptors, both branches (vladv1: alpha from the two best cluster dots per
descriptor; vladv2: sklearn NearestNeighbors -- the reference squares the neighbour INDICES, `kneighbors(...)[1]`,
and the fixture records exactly that), plus one forward pass of each initialised layer.

    python oracle/gen_golden_vlad_init.py      -> tests/golden/vlad_init_g12.npz
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import OUT, _install_stubs  # noqa: E402


def main():
    _install_stubs()
    import torch
    from cslam.vpr.netvlad import NetVLADLayer
    g = {}
    rng = np.random.default_rng(12)
    K, C = 64, 32
    clsts = rng.standard_normal((K, C)).astype(np.float32)
    train = rng.standard_normal((500, C)).astype(np.float32)
    train /= np.linalg.norm(train, axis=1, keepdims=True)
    x = rng.standard_normal((2, C, 5, 6)).astype(np.float32)
    g["clsts"], g["train"], g["x"] = clsts, train, x
    for tag, v2 in (("v1", False), ("v2", True)):
        torch.manual_seed(0)
        layer = NetVLADLayer(num_clusters=K, dim=C, vladv2=v2)
        layer.init_params(clsts.copy(), train.copy())
        g[tag + "/alpha"] = np.float64(layer.alpha)
        g[tag + "/conv_w"] = layer.conv.weight.detach().numpy().reshape(K, C)
        g[tag + "/centroids"] = layer.centroids.detach().numpy()
        if layer.conv.bias is not None:
            g[tag + "/conv_b"] = layer.conv.bias.detach().numpy()
        with torch.no_grad():
            g[tag + "/y"] = layer.eval()(torch.from_numpy(x.copy())).numpy()
    np.savez_compressed(os.path.join(OUT, "vlad_init_g12.npz"), **g)
    print({k: (v.shape, v.dtype) for k, v in g.items()})


if __name__ == "__main__":
    main()
