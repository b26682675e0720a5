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


def _ring_schedule(lib, nqt, n_rows, n_xcd=8, wpx=32):
    import ctypes as C
    info = (C.c_int32 * 6)()
    _lib.check(lib.cslam_ring_schedule_describe(nqt, n_rows, n_xcd, wpx, C.byref(info), None, 0, None, None, None))
    sq, sb, ntask, lists, _, words = list(info)
    assert words == 8
    tasks = np.zeros((max(ntask, 1), 8), dtype=np.int32)
    off = np.zeros(n_xcd * wpx + 1, dtype=np.int32)
    qn = np.zeros(nqt, dtype=np.int32)
    qo = np.zeros(nqt, dtype=np.int32)
    _lib.check(lib.cslam_ring_schedule_describe(nqt, n_rows, n_xcd, wpx, C.byref(info), tasks.ctypes.data_as(C.c_void_p), tasks.size,
                                                off.ctypes.data_as(C.c_void_p), qn.ctypes.data_as(C.c_void_p),
                                                qo.ctypes.data_as(C.c_void_p)))
    return sq, sb, tasks[:ntask], off, qn, qo, lists


@pytest.mark.parametrize("nqt,n_rows", [(391, 100_000), (4, 100_000), (1, 100_000), (2, 1200), (3, 10_000), (7, 50_000), (64, 100_000),
                                        (391, 700), (33, 4321), (5, 256), (128, 50_000), (8, 125_000), (4, 31), (2, 1_000_000),
                                        (32, 300_000)])
def test_ring_schedule_covers_every_pair_once_and_balances_the_xcds(nqt, n_rows):
    """The static schedule of the persistent candidate stage (csrc/sim_topk_ring.hip): every (query tile, bank row) pair belongs to
    exactly one task, a query tile's lists are numbered 0 .. qt_nseg - 1 without gaps, the per-tile list offsets are the running
    sum, and the workgroups of the launch carry the same load: no workgroup computes more than one 256-row tile (plus the 32-row
    rounding of the cuts) beyond the mean of the busy ones."""
    lib = _lib.load()
    n_xcd, wpx = 8, 32
    sq, sb, tasks, off, qn, qo, lists = _ring_schedule(lib, nqt, n_rows, n_xcd, wpx)
    assert sq * sb <= wpx and sq >= 1 and sb >= 1
    seen = np.zeros((nqt, n_rows), dtype=np.int32)
    segs = [set() for _ in range(nqt)]
    load = np.zeros(n_xcd * wpx, dtype=np.int64)           # 32-row blocks of matrix work per workgroup
    for w in range(n_xcd * wpx):
        for qt, row0, ntiles, seg, run, stride, row_end, _ in tasks[off[w]:off[w + 1]]:
            assert 0 <= qt < nqt and ntiles >= 0 and run >= 0 and row_end <= n_rows and stride >= 256 and stride % 256 == 0
            assert ntiles <= 128, "the packed candidate lists address a task's rows with 13 bits (csrc/sim_topk_pair_dev.h)"
            for i in range(ntiles):
                a = row0 + i * stride
                b = min(a + 256, row_end)
                assert a < b, "an empty tile"
                seen[qt, a:b] += 1
                load[w] += -(-(b - a) // 32)
            if ntiles:
                assert row0 + (ntiles - 1) * stride + 256 >= row_end, "a walk stops short of its end row"
            assert seg not in segs[qt]
            segs[qt].add(seg)
    assert np.array_equal(seen, np.ones_like(seen))
    for qt in range(nqt):
        assert segs[qt] == set(range(qn[qt])), (qt, sorted(segs[qt]), qn[qt])
    assert np.array_equal(qo, np.concatenate([[0], np.cumsum(qn)[:-1]])) and lists == int(qn.sum())
    busy = load[load > 0]
    # chunks of a run differ by the 32-row rounding of its cuts (the last column of a patch takes what is left: up to sb - 1 blocks
    # less per run), and a query group that is not full leaves its missing rows' workgroups out of one run
    nruns = max(off[w + 1] - off[w] for w in range(n_xcd * wpx))
    assert busy.max() <= busy.mean() * 1.01 + 3 * nruns, (busy.max(), busy.mean(), nruns)


def test_ring_schedule_of_the_bench_step_is_even():
    """1024 queries against 100 000 rows (the bench's in-step launch): 1564 tile pairs on 256 workgroups were 6 or 7 whole tiles
    each -- a seventh round with 28 of 256 workgroups busy; by rows every workgroup gets 6 tiles + 32 rows (or less)."""
    lib = _lib.load()
    sq, sb, tasks, off, qn, qo, lists = _ring_schedule(lib, 4, 100_000)
    rows = np.zeros(256, dtype=np.int64)
    for w in range(256):
        for qt, row0, ntiles, seg, run, stride, row_end, _ in tasks[off[w]:off[w + 1]]:
            assert stride == 256
            rows[w] += row_end - row0
    assert rows.min() > 0 and rows.max() <= 6 * 256 + 32 and rows.sum() == 4 * 100_000
