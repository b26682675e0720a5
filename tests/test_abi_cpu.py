"""CPU suite: the C-ABI library loads, exports every symbol include/cslam_hip.h declares, and
the product path fails loudly (no CPU fallback) when no GPU is visible."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from cslam_amd import _lib


def declared_symbols(header="cslam_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cslam_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 23
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/cslam_hip.h but not exported"
    assert set(names) == set(_lib.EXPORTED_SYMBOLS), "ctypes signature table out of sync with the header"
    # the experimental header (A/B partners, profiling hooks, peak micro-benchmarks) is kept apart from the stable ABI
    exp = declared_symbols("cslam_hip_experimental.h")
    assert set(exp) == set(_lib.EXPERIMENTAL_SYMBOLS) and not (set(exp) & set(names))
    for n in exp:
        assert hasattr(lib, n), f"{n} declared in include/cslam_hip_experimental.h but not exported"
    assert not any(("debug" in n or "peak" in n) for n in names), "diagnostics belong in the experimental header"
    assert _lib.load().cslam_version() >= 100


def _no_gpu():
    n = ctypes.c_int(0)
    return _lib.load().cslam_device_count(ctypes.byref(n)) != 0 or n.value == 0


@pytest.mark.skipif(not _no_gpu(), reason="GPU present: covered by the -m gpu suite")
def test_product_path_fails_loudly_without_gpu():
    from cslam_amd.nns_matching import NearestNeighborsMatching
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    nn = NearestNeighborsMatching()
    assert nn.search(np.zeros(4, dtype=np.float32), 3) == ([], [])      # never-populated bank (reference :52-53)
    assert nn.search_best(np.zeros(4)) == (None, None)
    with pytest.raises(_lib.CslamHipError):
        nn.add_item(np.ones(8, dtype=np.float32), 0)
    with pytest.raises(_lib.CslamHipError):
        NearestNeighborsMatching(dim=16)
    p = {"robot_id": 0, "max_nb_robots": 2, "frontend.sensor_type": "stereo", "frontend.similarity_threshold": 0.0}
    lcsm = LoopClosureSparseMatching(p)
    with pytest.raises(_lib.CslamHipError):
        lcsm.add_local_global_descriptor(np.ones(8), 0)
    h = ctypes.c_void_p()
    rc = _lib.load().cslam_bank_create(0, 16, 0, ctypes.byref(h))
    assert rc == -2 and b"hip" in _lib.load().cslam_last_error().lower()
    assert _lib.load().cslam_bank_create(0, 0, 0, ctypes.byref(h)) == -1  # argument check precedes HIP


def test_extractor_disabled_mode_needs_no_gpu():
    from cslam_amd.vpr.netvlad import NetVLAD
    from cslam_amd.vpr.cosplace import CosPlace
    assert NetVLAD({"frontend.nn_checkpoint": "disable"}, None).compute_embedding(None).shape == (128,)
    cp = CosPlace({"frontend.nn_checkpoint": "Disable", "frontend.cosplace.descriptor_dim": 64}, None)
    assert cp.compute_embedding(None).shape == (64,)


def test_backbone_state_dict_names_match_torchvision_layout():
    """Checkpoints of the reference address parameters by torchvision's names."""
    from cslam_amd.vpr.backbones import get_backbone
    vgg, c = get_backbone("vgg16")
    keys = list(vgg.state_dict().keys())
    assert c == 512 and keys[0] == "0.weight" and keys[-1] == "28.bias" and len(keys) == 26
    r18, c = get_backbone("resnet18")
    keys = set(r18.state_dict().keys())
    assert c == 512 and {"0.weight", "1.running_mean", "4.0.conv1.weight", "5.0.downsample.0.weight",
                         "7.1.bn2.bias"} <= keys
    import torch
    assert r18(torch.zeros(1, 3, 224, 224)).shape == (1, 512, 7, 7)
    assert vgg(torch.zeros(1, 3, 224, 224)).shape == (1, 512, 14, 14)
