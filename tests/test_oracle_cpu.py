"""CPU suite: the oracle (oracle/nns_oracle.c) against the reference-generated golden vectors.

tests/golden/nns_g1.npz was produced by oracle/gen_golden.py importing the REAL reference
(cslam/nns_matching.py) in the build container.  Indices must be identical; scores within
3e-7 (the reference accumulates float32 dots for float32 queries -- see nns_oracle.c header).
"""
import numpy as np
import pytest

from helpers import GOLDEN, assert_topk_equal, nns_case_inputs, nns_case_names
from oracle import pyoracle


@pytest.fixture(scope="module")
def g1():
    return np.load(GOLDEN + "/nns_g1.npz")


def test_oracle_matches_reference_golden(g1):
    names = nns_case_names(g1)
    assert len(names) >= 17
    for name in names:
        bank, q = nns_case_inputs(g1, name)
        k = int(g1[name + "/k"])
        idx, sims, cnt = pyoracle.nns_search(bank, q, k)
        assert_topk_equal(idx, sims, cnt, g1[name + "/idx"], g1[name + "/sims"], g1[name + "/cnt"], 3e-7)


def test_oracle_tie_rule_and_nan():
    # duplicates tie exactly -> larger row first; zero row -> NaN ranks first (argsort()[::-1])
    rng = np.random.default_rng(0)
    bank = rng.standard_normal((6, 16)).astype(np.float32)
    bank[4] = bank[1]
    bank[2] = 0.0
    q = bank[1:2].copy()
    idx, sims, cnt = pyoracle.nns_search(bank, q, 4)
    assert idx[0, 0] == 2 and np.isnan(sims[0, 0])
    assert list(idx[0, 1:3]) == [4, 1]
    assert sims[0, 1] == sims[0, 2]


def test_oracle_row_limit_is_causal_mask():
    rng = np.random.default_rng(1)
    bank = rng.standard_normal((50, 32)).astype(np.float32)
    q = bank[:10].copy()
    lim = np.arange(10, dtype=np.int64)
    idx, sims, cnt = pyoracle.nns_search(bank, q, 3, row_limit=lim)
    assert list(cnt) == [min(3, i) for i in range(10)]
    for j in range(10):
        assert np.all(idx[j, :cnt[j]] < j)
        ref = pyoracle.nns_search(bank[:j], q[j:j + 1], 3)[0][0] if j else np.full(3, -1)
        assert np.array_equal(idx[j], ref)


def test_oracle_cosine_equals_euclidean_order():
    """The reference's own numerical pin (tests/test_sparse_matching.py:51-81): on unit vectors
    cosine order == Euclidean order (ties within 1e-6 tolerated)."""
    rng = np.random.default_rng(5)
    bank = rng.random((100, 100))
    bank /= np.linalg.norm(bank, axis=1, keepdims=True)
    bank32 = bank.astype(np.float32)
    for _ in range(20):
        q = rng.random(100)
        q /= np.linalg.norm(q)
        ds = np.linalg.norm(q[None, :] - bank32, axis=1)
        order = np.argsort(ds)
        idx, sims, _ = pyoracle.nns_search(bank32, q[None, :], 100)
        assert np.all(sims[0, :-1] >= sims[0, 1:])
        for j in range(100):
            if order[j] != idx[0, j]:
                assert abs(ds[order[j]] - ds[idx[0, j]]) < 1e-6


def test_vlad_init_params_host_logic_matches_reference():
    """SURVEY 8 row a5 (host arithmetic, no GPU): NetVLADLayer.init_params against the reference's own outputs
    (G12, oracle/gen_golden_vlad_init.py), and the numpy oracle's forward pass of the initialised layers."""
    from cslam_amd.vpr.netvlad import NetVLADLayer
    from oracle import heads_oracle as ho
    g = np.load(GOLDEN + "/vlad_init_g12.npz")
    for tag, v2 in (("v1", False), ("v2", True)):
        layer = NetVLADLayer(64, 32, device="cpu", vladv2=v2).init_params(g["clsts"], g["train"])
        assert abs(layer.alpha - float(g[tag + "/alpha"])) <= 1e-6 * abs(float(g[tag + "/alpha"]))
        assert np.allclose(layer.conv_weight.numpy(), g[tag + "/conv_w"], rtol=1e-6, atol=0)
        assert np.array_equal(layer.centroids.numpy(), g[tag + "/centroids"])
        b = None if not v2 else layer.conv_bias.numpy()
        assert (b is None) == (tag + "/conv_b" not in g.files)
        y = ho.vlad_forward(g["x"], layer.conv_weight.numpy(), b, layer.centroids.numpy())
        assert np.max(np.abs(y - g[tag + "/y"])) < 1e-6


def test_extract_oracle_vlad_loop_matches_reference_golden():
    """oracle/extract_oracle.py (the cpu_baseline of the extract leg) restates NetVLADLayer.forward with the
    reference's per-cluster loop; pinned by G3, the outputs of the real reference layer (oracle/gen_golden_heads.py)."""
    import torch
    from oracle import extract_oracle
    g = np.load(GOLDEN + "/heads_g.npz")
    w, c = torch.from_numpy(g["vlad/conv_w"]), torch.from_numpy(g["vlad/centroids"])
    for case in ("vlad_a", "vlad_b"):
        y = extract_oracle.netvlad_layer_forward(torch.from_numpy(g[case + "/x"]), w, c).numpy()
        assert np.abs(y - g[case + "/y"]).max() <= 2e-7


def test_extract_oracle_vgg16_is_the_reference_layer_list():
    """features[:-2] of torchvision's VGG-16 (netvlad.py:163-171): 13 convolutions, the last one without ReLU / pool,
    four 2x2 poolings -> [1, 512, 14, 14] for a 224 px input; equal to the product's trunk module on the same weights
    (the module whose parameter names carry the reference's checkpoints)."""
    import torch
    from oracle import extract_oracle
    from cslam_amd.vpr.backbones import vgg16_features_trunk
    torch.manual_seed(0)
    trunk = vgg16_features_trunk().eval()
    convs = [m for m in trunk if isinstance(m, torch.nn.Conv2d)]
    assert len(convs) == 13
    params = [(m.weight.detach(), m.bias.detach()) for m in convs]
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        a, b = extract_oracle.vgg16_encoder(x, params), trunk(x)
    assert a.shape == (1, 512, 4, 4) and torch.equal(a, b)
    assert (a < 0).any()                                   # no ReLU after conv5_3


def _prep_image(i):
    img = np.random.default_rng(7 + i).integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    if i == 1:
        yy, xx = np.mgrid[0:480, 0:640]
        img = np.stack([(xx * 255 // 639), (yy * 255 // 479), ((xx // 40 + yy // 40) % 2) * 255], axis=2).astype(np.uint8)
    return img


def test_preprocess_oracle_matches_the_pillow_golden():
    """oracle/heads_oracle.py `preprocess` (the checker of cslam_preprocess_dev) against tests/golden/heads_g.npz `prep_*`: Pillow's own
    crop + bicubic resize of the two 480 x 640 frames oracle/gen_golden_heads.py fed it (netvlad.py:202-208,223-226)."""
    from oracle import heads_oracle as ho
    g = np.load(GOLDEN + "/heads_g.npz")
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    for i in range(2):
        img = _prep_image(i)
        top, left = int(round((480 - 376) / 2.0)), int(round((640 - 376) / 2.0))
        assert np.array_equal(ho.pil_bicubic_resize_u8(img[top:top + 376, left:left + 376], 224), g["prep_%d/resized_u8" % i])
        assert np.max(np.abs(ho.preprocess(img, 376, 224, mean, std) - g["prep_%d/out" % i])) <= 2.4e-7


@pytest.mark.parametrize("size,out", [(100, 224), (300, 224), (600, 224), (301, 111), (700, 224), (376, 225), (224, 224), (256, 112)])
def test_preprocess_oracle_resize_equals_pillow_at_other_geometries(size, out):
    """The geometries tests/test_heads_gpu.py::test_preprocess_other_geometries_bit_exact checks the kernels on (5 .. 15 taps,
    upsampling, odd sizes): the numpy restatement of ImagingResample's two 8-bit passes against Pillow itself, where Pillow is
    installed (it is in the build container; nothing here travels)."""
    Image = pytest.importorskip("PIL.Image")
    from oracle import heads_oracle as ho
    img = np.random.default_rng(size + out).integers(0, 256, size=(size, size, 3), dtype=np.uint8)
    img[: size // 2, : size // 3] //= 8                                 # flat dark region + edges: negative lobes, clipping
    want = np.asarray(Image.fromarray(img).resize((out, out), Image.BICUBIC))
    assert np.array_equal(ho.pil_bicubic_resize_u8(img, out), want)
