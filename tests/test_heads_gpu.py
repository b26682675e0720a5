"""GPU parity tests of the descriptor heads and the image transform (-m gpu).

Golden vectors come from the REFERENCE's own torch modules (NetVLADLayer, GeM/L2Norm/Flatten),
sklearn's PCA and Pillow (oracle/gen_golden_heads.py); the numpy oracle is checked alongside.
Tolerance: float32 kernels vs float32 reference modules, 1e-5 absolute on unit-norm outputs
(north_star's fp32 gate); the integer resize is bit-exact.
"""
import os
import numpy as np
import pytest

from helpers import GOLDEN
from oracle import heads_oracle as ho

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN + "/heads_g.npz")


@pytest.fixture(scope="module")
def T():
    import torch
    from cslam_amd.vpr import heads
    return torch, heads


def dev(T, a):
    torch, _ = T
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_vlad_matches_reference_layer(g, T):
    torch, heads = T
    for name in ("a", "b"):
        x = g[f"vlad_{name}/x"]
        y = heads.vlad_aggregate(dev(T, x), dev(T, g["vlad/conv_w"]), None, dev(T, g["vlad/centroids"])).cpu().numpy()
        assert y.shape == g[f"vlad_{name}/y"].shape
        assert np.max(np.abs(y - g[f"vlad_{name}/y"])) < 2e-7          # outputs are ~1e-2: tight absolute
        o = ho.vlad_forward(x, g["vlad/conv_w"], None, g["vlad/centroids"])
        assert np.max(np.abs(y - o)) < 2e-7
        assert np.allclose(np.linalg.norm(y, axis=1), 1.0, atol=1e-5)


def test_vlad_with_bias_and_odd_shapes(T):
    torch, heads = T
    rng = np.random.default_rng(0)
    for (B, C, H, W) in [(3, 512, 5, 3), (1, 96, 20, 13), (2, 64, 1, 1)]:
        x = rng.standard_normal((B, C, H, W)).astype(np.float32)
        w = rng.standard_normal((64, C)).astype(np.float32)
        b = rng.standard_normal(64).astype(np.float32)
        c = rng.random((64, C)).astype(np.float32)
        y = heads.vlad_aggregate(dev(T, x), dev(T, w), dev(T, b), dev(T, c)).cpu().numpy()
        o = ho.vlad_forward(x, w, b, c)
        assert np.max(np.abs(y - o)) < 1e-6


def test_vlad_small_batch_path_equals_batch_kernel(T):
    """B <= 8 runs the three-kernel split (online path), larger batches the one-workgroup-per-image kernel."""
    torch, heads = T
    rng = np.random.default_rng(4)
    for (C, H, W) in [(512, 14, 14), (96, 7, 9), (512, 16, 16)]:
        x = rng.standard_normal((12, C, H, W)).astype(np.float32)
        w = rng.standard_normal((64, C)).astype(np.float32) * 0.3
        c = rng.random((64, C)).astype(np.float32)
        big = heads.vlad_aggregate(dev(T, x), dev(T, w), None, dev(T, c)).cpu().numpy()
        o = ho.vlad_forward(x, w, None, c)
        assert np.max(np.abs(big - o)) < 1e-6
        for b in (1, 3, 8):
            small = heads.vlad_aggregate(dev(T, x[:b]), dev(T, w), None, dev(T, c)).cpu().numpy()
            assert np.max(np.abs(small - o[:b])) < 1e-6 and np.max(np.abs(small - big[:b])) < 1e-6


def test_cosplace_head_matches_reference_modules(g, T):
    torch, heads = T
    for tag in ("p3", "p237"):
        x, W, b, p = g[f"gem_{tag}/x"], g[f"gem_{tag}/W"], g[f"gem_{tag}/b"], float(g[f"gem_{tag}/p"])
        y = heads.gem_fc_head(dev(T, x), p, 1e-6, dev(T, W), dev(T, b)).cpu().numpy()
        assert np.max(np.abs(y - g[f"gem_{tag}/y"])) < 1e-5
        assert np.max(np.abs(y - ho.cosplace_head(x, p, 1e-6, W, b))) < 1e-5
        assert np.allclose(np.linalg.norm(y, axis=1), 1.0, atol=1e-5)
        if x.ndim == 4 and x.shape[1] % 4 == 0:
            # the same map in channels_last storage (what the trunks write) goes through the NHWC kernel
            xc = dev(T, x).contiguous(memory_format=torch.channels_last)
            assert not xc.is_contiguous() or x.shape[2] * x.shape[3] == 1
            yc = heads.gem_fc_head(xc, p, 1e-6, dev(T, W), dev(T, b)).cpu().numpy()
            assert np.max(np.abs(yc - g[f"gem_{tag}/y"])) < 1e-5 and np.max(np.abs(yc - y)) < 2e-6


def test_preprocess_channels_last_output_equals_planar(T):
    """cslam_preprocess_nhwc_dev writes the same values as cslam_preprocess_dev, in channels_last storage."""
    torch, heads = T
    gen = torch.Generator(device="cuda").manual_seed(3)
    for (H, W, crop, out_hw) in [(480, 640, 376, 224), (300, 280, 260, 224), (100, 120, 376, 96)]:
        fr = torch.randint(0, 256, (3, H, W, 3), generator=gen, device="cuda", dtype=torch.uint8)
        a = heads.preprocess(fr, crop, out_hw)
        b = heads.preprocess(fr, crop, out_hw, channels_last=True)
        assert b.shape == a.shape and b.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(a, b)


def test_pca_project_matches_sklearn(g, T):
    torch, heads = T
    for t in ("n", "w"):
        comp, mean, var, x = g[f"pca_{t}/components"], g[f"pca_{t}/mean"], g[f"pca_{t}/var"], g[f"pca_{t}/x"]
        mean_proj = (mean.reshape(1, -1) @ comp.T).reshape(-1).astype(np.float32)
        inv = None
        if t == "w":
            inv = dev(T, (1.0 / np.sqrt(var)).astype(np.float32))
        y = heads.pca_project(dev(T, x), dev(T, comp), dev(T, mean_proj), inv).cpu().numpy()
        assert np.max(np.abs(y - g[f"pca_{t}/y"])) < 1e-5
        assert np.max(np.abs(y - ho.pca_transform_normalize(x, comp, mean, var, t == "w"))) < 1e-5


def test_pca_gemm_full_shape_split_k(T):
    """NetVLAD's real projection shape (32768 -> 4096), batch 130 (ragged tile), vs float64 numpy."""
    torch, heads = T
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((130, 32768), generator=gen, device="cuda")
    comp = torch.randn((4096, 32768), generator=gen, device="cuda") / 181.0
    y = heads.pca_project(x, comp, None, None)
    ref = (x[:8].double() @ comp.double().T)
    ref = ref / ref.norm(dim=1, keepdim=True)
    assert float((y[:8].double() - ref).abs().max()) < 1e-5
    yb = heads.pca_project(x[122:].contiguous(), comp, None, None)        # batch-size independence
    assert float((yb - y[122:]).abs().max()) < 1e-6
    for b in (1, 2, 3, 4):                                                # small batches: the GEMV kernel
        ys = heads.pca_project(x[:b].contiguous(), comp, None, None)
        assert float((ys.double() - ref[:b]).abs().max()) < 1e-5
        assert float((ys - y[:b]).abs().max()) < 1e-6
    xs = torch.randn((3, 288), generator=gen, device="cuda")              # Din not a multiple of the 256-float stride
    cs = torch.randn((50, 288), generator=gen, device="cuda")
    mp = torch.randn(50, generator=gen, device="cuda")
    rs = xs.double() @ cs.double().T - mp.double()
    rs = rs / rs.norm(dim=1, keepdim=True)
    assert float((heads.pca_project(xs, cs, mp, None).double() - rs).abs().max()) < 1e-5


def test_l2_normalize_variants(T):
    torch, heads = T
    x = torch.randn((37, 1000), device="cuda") * 3
    x[5] = 0
    a = heads.l2_normalize_(x.clone())
    assert torch.allclose(a, torch.nn.functional.normalize(x, dim=1), atol=1e-6)
    b = heads.l2_normalize_(x.clone(), zero_norm_to_one=True)
    assert float(b[5].abs().max()) == 0.0 and torch.allclose(b[6], x[6] / x[6].norm(), atol=1e-6)


def test_preprocess_bit_exact_vs_pillow(g, T):
    torch, heads = T
    for i in range(2):
        img = np.random.default_rng(7 + i).integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
        if i == 1:
            yy, xx = np.mgrid[0:480, 0:640]
            img = np.stack([(xx * 255 // 639), (yy * 255 // 479), ((xx // 40 + yy // 40) % 2) * 255],
                           axis=2).astype(np.uint8)
        out = heads.preprocess(dev(T, img[None]), 376).cpu().numpy()[0]
        gold = g[f"prep_{i}/out"]
        # recover the uint8 resize result: must be identical to Pillow's
        mean = np.array(heads.IMAGENET_DEFAULT_MEAN, dtype=np.float32)[:, None, None]
        std = np.array(heads.IMAGENET_DEFAULT_STD, dtype=np.float32)[:, None, None]
        u8 = np.rint((out * std + mean) * 255.0).astype(np.int64).transpose(1, 2, 0)
        assert np.array_equal(u8, g[f"prep_{i}/resized_u8"].astype(np.int64))
        assert np.max(np.abs(out - gold)) <= 2.4e-7           # <= 1 float32 ulp at |x| <= 2.7
    # batch of frames, other crop size
    imgs = np.random.default_rng(3).integers(0, 256, size=(3, 300, 400, 3), dtype=np.uint8)
    out = heads.preprocess(dev(T, imgs), 256, 112).cpu().numpy()
    for b in range(3):
        o = ho.preprocess(imgs[b], 256, 112, heads.IMAGENET_DEFAULT_MEAN, heads.IMAGENET_DEFAULT_STD)
        assert np.max(np.abs(out[b] - o)) <= 2.4e-7


@pytest.mark.parametrize("H,W,crop,out", [(120, 160, 100, 224), (301, 403, 300, 224), (480, 640, 600, 224), (301, 403, 301, 111),
                                          (480, 640, 700, 224), (64, 48, 376, 224), (480, 640, 376, 225), (480, 641, 224, 224)])
def test_preprocess_other_geometries_bit_exact(T, H, W, crop, out):
    """cslam_preprocess_dev over the tap counts and layouts its two kernels meet: 5 taps (upsampling), 7, 9, 13, 15 taps (the
    last beyond preprocess_tile_kernel's instantiations: the table kernel), frames smaller than the crop (zero padding like
    torchvision's CenterCrop), rows that are not dword-aligned, an output width that is not a multiple of four, crop == output
    (5 taps, a copy) -- against the numpy restatement of Pillow's two-pass 8-bit resize (oracle/heads_oracle.py, pinned to
    Pillow's own output by tests/golden): the uint8 result identical, the floats within an ulp."""
    torch, heads = T
    imgs = np.random.default_rng(H + W + crop).integers(0, 256, size=(2, H, W, 3), dtype=np.uint8)
    got = heads.preprocess(dev(T, imgs), crop, out).cpu().numpy()
    mean = np.array(heads.IMAGENET_DEFAULT_MEAN, dtype=np.float32)[:, None, None]
    std = np.array(heads.IMAGENET_DEFAULT_STD, dtype=np.float32)[:, None, None]
    for b in range(2):
        want = ho.preprocess(imgs[b], crop, out, heads.IMAGENET_DEFAULT_MEAN, heads.IMAGENET_DEFAULT_STD)
        assert got[b].shape == want.shape == (3, out, out)
        assert np.array_equal(np.rint((got[b] * std + mean) * 255.0), np.rint((want * std + mean) * 255.0))
        assert np.max(np.abs(got[b] - want)) <= 2.4e-7


@pytest.mark.parametrize("chunk", [12, 4])
def test_netvlad_batch_extraction_over_two_lanes_equals_one_stream(T, chunk):
    """NetVLAD.compute_embeddings_batch_device: chunks alternating over two HIP streams (each with its own trunk workspaces) give
    the descriptors of the same chunks on one stream, bit for bit -- chunk 12: the batch kernels (stem, pair products,
    channels-last VLAD, pair PCA); chunk 4: the small-batch kernels, whose scratch (VLAD partials, split-K sums) is per stream."""
    torch, heads = T
    from cslam_amd.vpr.netvlad import NetVLAD
    nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 256}, None)
    gen = torch.Generator(device="cuda").manual_seed(chunk)
    frames = torch.randint(0, 256, (5 * chunk + 3, 240, 320, 3), generator=gen, device="cuda", dtype=torch.uint8)
    one = nv.compute_embeddings_batch_device(frames, chunk, 1)
    for _ in range(3):
        two = nv.compute_embeddings_batch_device(frames, chunk, 2)
        torch.cuda.synchronize()
        assert two.shape == one.shape == (frames.shape[0], 256) and torch.equal(one, two)
    three = nv.compute_embeddings_batch_device(frames, chunk, 3)
    assert torch.equal(one, three)
    assert torch.equal(one[:chunk], nv.compute_embeddings_device(frames[:chunk]))


def test_projection_scratch_is_per_stream(T):
    """cslam_pca_project_pairs_dev / cslam_pca_project_dev keep operand pairs and split-K sums in scratch between their kernels:
    two streams in flight at once (two extraction lanes, two host threads) must not share it."""
    torch, heads = T
    gen = torch.Generator(device="cuda").manual_seed(5)
    comp = torch.randn((512, 4096), generator=gen, device="cuda") / 64.0
    mp = torch.randn(512, generator=gen, device="cuda") * 0.1
    pairs = heads.pca_pair_weights(comp)
    xs = [torch.randn((96, 4096), generator=gen, device="cuda") for _ in range(2)]
    small = [x[:3].contiguous() for x in xs]
    want = [heads.pca_project(x, comp, mp, None, pairs) for x in xs] + [heads.pca_project(x, comp, mp, None) for x in small]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    for _ in range(10):
        got = [None] * 4
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                for _ in range(4):
                    got[i] = heads.pca_project(xs[i], comp, mp, None, pairs)
                    got[2 + i] = heads.pca_project(small[i], comp, mp, None)
        torch.cuda.synchronize()
        assert all(torch.equal(g, w) for g, w in zip(got, want))


def test_extractors_end_to_end_structure(T):
    """NetVLAD / CosPlace drop-in classes with seeded random weights (no checkpoints ship with
    the reference): HIP pipeline == the same pipeline restated with torch + the numpy oracle."""
    torch, heads = T
    from cslam_amd.vpr.netvlad import NetVLAD
    from cslam_amd.vpr.cosplace import CosPlace
    frames = np.random.default_rng(1).integers(0, 256, size=(2, 480, 640, 3), dtype=np.uint8)
    nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376,
                  "frontend.netvlad.pca_dim": 128}, None)
    e = nv.compute_embedding(frames[0])
    assert e.shape == (128,) and e.dtype == np.float32 and abs(np.linalg.norm(e) - 1) < 1e-5
    x = torch.from_numpy(ho.preprocess(frames[0], 376, 224, heads.IMAGENET_DEFAULT_MEAN,
                                       heads.IMAGENET_DEFAULT_STD))[None].cuda()
    with torch.no_grad():
        f = nv.encoder(x)
    v = ho.vlad_forward(f.cpu().numpy(), nv.pool.conv_weight.cpu().numpy(), None, nv.pool.centroids.cpu().numpy())
    ref = ho.sk_normalize(v @ nv.pca_components.cpu().numpy().T)
    assert np.max(np.abs(e - ref[0])) < 1e-5
    cp = CosPlace({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376,
                   "frontend.cosplace.descriptor_dim": 512, "frontend.cosplace.backbone": "resnet18"}, None)
    e = cp.compute_embeddings_device(torch.from_numpy(frames).cuda()).cpu().numpy()
    assert e.shape == (2, 512) and np.allclose(np.linalg.norm(e, axis=1), 1, atol=1e-5)
    xs = np.stack([ho.preprocess(fr, 376, 224, heads.IMAGENET_DEFAULT_MEAN, heads.IMAGENET_DEFAULT_STD)
                   for fr in frames])
    with torch.no_grad():
        f = cp.model.backbone(torch.from_numpy(xs).cuda())
    ref = ho.cosplace_head(f.cpu().numpy(), 3.0, 1e-6, cp.model.fc_weight.cpu().numpy(),
                           cp.model.fc_bias.cpu().numpy())
    assert np.max(np.abs(e - ref)) < 1e-5
    assert NetVLAD({"frontend.nn_checkpoint": "disable"}, None).compute_embedding(frames[0]).shape == (128,)


# ---- Winograd F(2x2,3x3) execution of the wide backbone convolutions (vpr/winograd.py) ----
@pytest.mark.parametrize("B,cin,cout,H,W,relu,pool,bias", [(2, 128, 256, 28, 28, True, False, True),
                                                          (3, 256, 64, 14, 6, True, True, True),
                                                          (1, 512, 512, 14, 14, False, False, True),
                                                          (40, 64, 128, 16, 24, True, True, True),
                                                          (40, 64, 96, 14, 14, True, False, True),     # ragged F(4x4) tiles
                                                          (40, 128, 64, 14, 18, True, True, True),
                                                          (2, 128, 64, 7, 9, True, True, True),        # odd maps, floor pooling
                                                          (3, 64, 64, 5, 1, False, False, True),
                                                          (2, 64, 128, 8, 10, False, True, False)])
def test_winograd_conv_equals_direct_float64(T, B, cin, cout, H, W, relu, pool, bias):
    """The two HIP transforms around 16 GEMMs == conv2d (+bias, ReLU, MaxPool) evaluated in float64."""
    torch, _ = T
    from torch import nn
    from cslam_amd.vpr.winograd import WinogradTrunk
    torch.manual_seed(B * 1000 + cin)
    mods = [nn.Conv2d(cin, cout, 3, padding=1, bias=bias)]
    if relu:
        mods.append(nn.ReLU())
        if pool:
            mods.append(nn.MaxPool2d(2, 2))
    seq = nn.Sequential(*mods).cuda().eval()
    x = torch.randn((B, cin, H, W), device="cuda")
    with torch.no_grad():
        ref = seq.double()(x.double())
    seq.float()
    scale = ref.abs().max().item()
    for tile, tol in ((2, 5e-6), (4, 2e-5)):
        wt = WinogradTrunk(seq, min_in_channels=64, tile=tile)
        assert [s.kind for s in wt.steps][0] == "wino" and wt.steps[0].relu == relu and wt.steps[0].pool == (relu and pool)
        y = wt(x)
        assert y.shape == ref.shape
        assert (y.double() - ref).abs().max().item() <= tol * scale, tile


def test_winograd_trunk_equals_direct_trunk_including_odd_maps(T):
    torch, _ = T
    from cslam_amd.vpr.backbones import vgg16_features_trunk
    from cslam_amd.vpr.winograd import WinogradTrunk
    torch.manual_seed(3)
    enc = vgg16_features_trunk().cuda().eval()
    for tile, minc, nw in ((2, 128, 10), (4, 64, 12)):
        wt = WinogradTrunk(enc, min_in_channels=minc, tile=tile)
        kinds = [s.kind for s in wt.steps]
        assert kinds.count("wino") == nw and kinds.count("c3") == 1 and kinds.count("direct") == 12 - nw \
            and kinds.count("torch") == 0
        assert sum(s.pool for s in wt.steps) == 4
        for B, hw in ((2, 224), (24, 112), (2, 72), (40, 200)):   # 72 -> ... -> 9 and 200 -> ... -> 25 -> 12: odd maps
            x = torch.randn((B, 3, hw, hw), device="cuda")
            with torch.no_grad():
                r = enc.double()(x.double())
                enc.float()
                a = enc(x)
            b = wt(x)
            assert a.shape == b.shape
            scale = r.abs().max().item()
            e_direct = (a.double() - r).abs().max().item() / scale
            e_wino = (b.double() - r).abs().max().item() / scale
            assert e_wino <= 1e-5 and e_wino <= 5 * e_direct + 1e-6, (tile, hw, e_wino, e_direct)


def test_netvlad_descriptors_winograd_vs_direct(T):
    torch, _ = T
    from cslam_amd.vpr.netvlad import NetVLAD
    frames = torch.from_numpy(np.random.default_rng(5).integers(0, 256, size=(4, 480, 640, 3), dtype=np.uint8)).cuda()
    base = {"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 512}
    frames = frames.repeat(16, 1, 1, 1)                        # 64 frames: enough tiles for the F(4x4) path
    b = NetVLAD(dict(base, **{"frontend.backbone_conv": "direct"}), None).compute_embeddings_device(frames)
    for mode in ("winograd", "winograd2"):
        a = NetVLAD(dict(base, **{"frontend.backbone_conv": mode}), None).compute_embeddings_device(frames)
        assert (a - b).abs().max().item() <= 1e-5              # unit-norm descriptors, north_star's fp32 gate
        assert torch.all((a * b).sum(1) > 1 - 1e-6)
        assert torch.equal(a[:4], a[4:8])                      # batch position does not change a descriptor


def test_first_layer_conv_c3_matches_float64(T):
    """conv1_1 in hand-written HIP (planar input -> NHWC, bias + ReLU fused) against float64 conv2d."""
    torch, _ = T
    from torch import nn
    from cslam_amd.vpr.winograd import WinogradTrunk
    torch.manual_seed(11)
    for cout, relu, hw in ((64, True, (37, 50)), (64, False, (224, 224)), (64, True, (1, 1)), (16, False, (8, 8)),
                           (32, True, (33, 65))):     # Cout = 64: tiled kernel (ragged tiles in both directions)
        mods = [nn.Conv2d(3, cout, 3, padding=1)] + ([nn.ReLU()] if relu else [])
        seq = nn.Sequential(*mods).cuda().eval()
        x = torch.randn((3, 3) + hw, device="cuda")
        wt = WinogradTrunk(seq)
        assert [s.kind for s in wt.steps] == ["c3"]
        y = wt(x)
        with torch.no_grad():
            ref = seq.double()(x.double())
        seq.float()
        assert y.shape == ref.shape and (y.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("name,B,hw", [("resnet18", 24, 224), ("resnet18", 2, 96), ("resnet50", 8, 224),
                                       ("resnet18", 130, 224)])          # 130 frames: layer4's 7x7 maps on ragged F(4x4)
def test_winograd_resnet_equals_torch_trunk(T, name, B, hw):
    """BatchNorm folding + Winograd 3x3 + fused shortcut/ReLU == the torch eval trunk, both judged against float64."""
    torch, _ = T
    from torch import nn
    from cslam_amd.vpr.backbones import resnet_trunk
    from cslam_amd.vpr.winograd import WinogradResNet
    torch.manual_seed(17)
    trunk = resnet_trunk(name).cuda().eval()
    with torch.no_grad():
        for m in trunk.modules():
            if isinstance(m, nn.BatchNorm2d):                      # non-trivial statistics to fold
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.6, 1.2)
                m.bias.normal_(0, 0.1)
    x = torch.randn((B, 3, hw, hw), device="cuda")
    with torch.no_grad():
        r = trunk.double()(x.double())
        trunk.float()
        a = trunk(x.contiguous(memory_format=torch.channels_last))
    for tile, direct in ((4, True), (4, False), (2, False)):
        # direct: every convolution through the implicit GEMM on fp16 pairs (csrc/conv_igemm.hip, the default); otherwise the 3x3 /
        # stride 1 layers through the fp32 Winograd pipeline (F(4x4) / F(2x2)), the rest through the implicit GEMM
        run = WinogradResNet(trunk, 64, tile, direct=direct)
        if direct:
            assert run.stem.Wg is not None and all(b["c1"].Wg is not None and b["c2"].Wg is not None for b in run.blocks)
        else:
            assert any(b["c1"].U is not None or b["c2"].U is not None for b in run.blocks)
        b = run(x)
        assert b.shape == r.shape
        scale = r.abs().max().item()
        e_direct = (a.double() - r).abs().max().item() / scale
        e_wino = (b.double() - r).abs().max().item() / scale
        assert e_wino <= 2e-5 and e_wino <= 6 * e_direct + 1e-6, (tile, e_wino, e_direct)


def test_cosplace_descriptors_winograd_vs_direct(T):
    torch, _ = T
    from cslam_amd.vpr.cosplace import CosPlace
    frames = torch.from_numpy(np.random.default_rng(6).integers(0, 256, size=(4, 480, 640, 3), dtype=np.uint8)).cuda()
    frames = frames.repeat(8, 1, 1, 1)
    base = {"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.cosplace.descriptor_dim": 512}
    for bb in ("resnet18", "vgg16"):
        p = dict(base, **{"frontend.cosplace.backbone": bb})
        d = CosPlace(dict(p, **{"frontend.backbone_conv": "direct"}), None).compute_embeddings_device(frames)
        for mode in ("winograd", "winograd2"):
            w = CosPlace(dict(p, **{"frontend.backbone_conv": mode}), None).compute_embeddings_device(frames)
            assert (w - d).abs().max().item() <= 1e-5, (bb, mode)


def test_cosplace_extraction_lanes_give_the_single_stream_descriptors(T):
    """CosPlace.compute_embeddings_batch_device: chunks alternating over two streams with a runner each -- the same kernels on the same
    frames, so bit-equal to the single-stream passes (and the caller's stream continues behind the lanes without a host wait)."""
    torch, _ = T
    from cslam_amd.vpr.cosplace import CosPlace
    frames = torch.from_numpy(np.random.default_rng(16).integers(0, 256, size=(22, 300, 320, 3), dtype=np.uint8)).cuda()
    cp = CosPlace({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 280, "frontend.cosplace.descriptor_dim": 128,
                   "frontend.cosplace.backbone": "resnet18"}, None)
    one = torch.cat([cp.compute_embeddings_device(frames[s:s + 8]) for s in range(0, 22, 8)])
    for lanes in (2, 3):
        got = cp.compute_embeddings_batch_device(frames, chunk=8, lanes=lanes)
        again = cp.compute_embeddings_batch_device(frames, chunk=8, lanes=lanes)
        assert got.shape == one.shape and torch.equal(got, one) and torch.equal(again, one)


def test_online_hip_graph_replay_equals_plain_launches(T):
    """With frontend.hip_graph compute_embedding (one keyframe) replays a captured HIP graph; it must return what the
    same kernels give when launched one by one (the library GEMMs may pick another solution under capture, hence
    1e-6 rather than bit equality), also after batched calls grew the shared workspaces in between."""
    torch, _ = T
    from cslam_amd.vpr.netvlad import NetVLAD
    from cslam_amd.vpr.cosplace import CosPlace
    rng = np.random.default_rng(8)
    frames = rng.integers(0, 256, size=(3, 480, 640, 3), dtype=np.uint8)
    small = rng.integers(0, 256, size=(400, 500, 3), dtype=np.uint8)         # a second frame shape -> second graph
    base = {"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 256,
            "frontend.cosplace.descriptor_dim": 128, "frontend.cosplace.backbone": "resnet18"}
    for cls in (NetVLAD, CosPlace):
        g = cls(dict(base, **{"frontend.hip_graph": True}), None)
        p = cls(dict(base), None)
        assert g.use_graph and not p.use_graph
        for rnd in range(2):
            for f in (frames[0], frames[1], small, frames[2]):
                a, b = g.compute_embedding(f), p.compute_embedding(f)
                assert a.dtype == np.float32 and a.shape == b.shape and np.abs(a - b).max() <= 1e-6, (cls.__name__, rnd)
            big = torch.from_numpy(np.repeat(frames, 11, axis=0)).cuda()     # 33 frames: grows V / M / split-K buffers
            g.compute_embeddings_device(big)
        assert g._online is not None and not g._online.failed and len(g._online.entries) == 2, cls.__name__


def test_tuned_gemm_table_is_loaded_and_harmless(T):
    """The TunableOp selections shipped for the VGG-16 trunk load on this box (same library build) and a GEMM whose
    shape is not in the table still runs; tuning itself stays off."""
    torch, _ = T
    import torch.cuda.tunable as tunable
    from cslam_amd.vpr import winograd
    winograd.use_tuned_gemms()
    if not tunable.is_enabled():
        pytest.skip("TunableOp was disabled by the environment (CSLAM_TUNED_GEMM=0)")
    assert not tunable.tuning_is_enabled()
    a, b = torch.randn((16, 300, 72), device="cuda"), torch.randn((16, 72, 40), device="cuda")
    assert torch.allclose(torch.bmm(a, b), torch.einsum("bij,bjk->bik", a.double(), b.double()).float(), atol=1e-3)
    names = [r[1] for r in tunable.get_results()]
    assert any(n.startswith("nn_512_12544_512_B_36") for n in names), names[:5]


def test_vlad_init_params_matches_reference(T):
    """NetVLADLayer.init_params (netvlad.py:63-92) on the reference's own outputs (G12): alpha,
    assignment weights / bias, centroids, and a forward pass of the initialised layer through the HIP kernel --
    vladv1 (no bias) and vladv2 (bias; the reference squares the neighbour indices, reproduced)."""
    from cslam_amd.vpr.netvlad import NetVLADLayer
    g = np.load(GOLDEN + "/vlad_init_g12.npz")
    for tag, v2 in (("v1", False), ("v2", True)):
        layer = NetVLADLayer(64, 32, device="cuda", vladv2=v2).init_params(g["clsts"], g["train"])
        assert abs(layer.alpha - float(g[tag + "/alpha"])) <= 1e-6 * abs(float(g[tag + "/alpha"]))
        assert np.allclose(layer.conv_weight.cpu().numpy(), g[tag + "/conv_w"], rtol=1e-6, atol=0)
        assert np.array_equal(layer.centroids.cpu().numpy(), g[tag + "/centroids"])
        assert (layer.conv_bias is None) == (not v2)
        if v2:
            assert np.allclose(layer.conv_bias.cpu().numpy(), g[tag + "/conv_b"], rtol=1e-6, atol=0)
        y = layer(dev(T, g["x"])).cpu().numpy()
        assert y.shape == g[tag + "/y"].shape and np.max(np.abs(y - g[tag + "/y"])) < 1e-6


@pytest.mark.parametrize("waves", [0])
@pytest.mark.parametrize("B,H,W,relu,pool,bias", [(2, 112, 112, True, False, True), (3, 37, 50, True, True, True),
                                                  (1, 8, 16, False, True, False), (2, 1, 1, False, False, True),
                                                  (300, 16, 16, True, False, True)])
def test_fused_winograd_64_to_128_equals_float64_and_unfused(T, B, H, W, relu, pool, bias, waves):
    """The same for 64 -> 128 channels (VGG-16 conv2_1), in the three forms of the kernel: 0 = persistent producer /
    consumer workgroups (the default; 600 virtual blocks > the compute units in the last case, so the block loop and its
    odd remainders run), 4 / 8 = one tile block per workgroup of 4 / 8 waves."""
    _fused_case(T, 128, B, H, W, relu, pool, bias, waves)


@pytest.mark.parametrize("B,H,W,relu,pool,bias", [(2, 224, 224, True, True, True), (3, 37, 50, True, False, True),
                                                  (1, 8, 16, False, True, False), (5, 16, 8, True, True, True),
                                                  (2, 1, 1, False, False, True), (1, 40, 70, True, True, True),
                                                  (7, 100, 36, True, True, True)])
@pytest.mark.parametrize("waves", [0])
def test_fused_winograd_64_to_64_equals_float64_and_unfused(T, B, H, W, relu, pool, bias, waves):
    _fused_case(T, 64, B, H, W, relu, pool, bias, waves)


def _fused_case(T, cout, B, H, W, relu, pool, bias, waves):
    """The single-kernel F(2x2,3x3) form of a 64 -> 64 channel layer (VGG-16 conv1_2: csrc/wino_fused.hip) against a
    float64 conv2d (+ ReLU + MaxPool2d) and against the transform / rocBLAS / transform pipeline it replaces; ragged
    tile blocks in both directions, with and without bias / ReLU / pooling."""
    torch, _ = T
    from torch import nn
    from cslam_amd.vpr.winograd import WinogradTrunk
    torch.manual_seed(23)
    mods = [nn.Conv2d(64, cout, 3, padding=1, bias=bias)] + ([nn.ReLU()] if relu else []) + \
        ([nn.MaxPool2d(2, 2)] if pool and relu else [])
    seq = nn.Sequential(*mods).cuda().eval()
    x = torch.randn((B, 64, H, W), device="cuda")
    fused = WinogradTrunk(seq, 64, 2, fused64=True)
    fused.fused_min_blocks = 0                       # (the trunk's own rule: fused from one tile block per CU on)
    plain = WinogradTrunk(seq, 64, 2, fused64=False)
    assert fused.steps[0].Up is not None and plain.steps[0].Up is None
    yf, yp = fused(x), plain(x)
    with torch.no_grad():
        ref = seq.double()(x.double())
    seq.float()
    scale = ref.abs().max().item()
    assert yf.shape == ref.shape == yp.shape
    ef = (yf.double() - ref).abs().max().item() / scale
    ep = (yp.double() - ref).abs().max().item() / scale
    assert ef <= 3e-6 and ef <= 3 * ep + 1e-7, (ef, ep)


@pytest.mark.parametrize("B,H,W,cout", [(3, 56, 56, 64), (300, 14, 10, 64), (2, 9, 7, 128)])
def test_fused_winograd_shortcut_add(T, B, H, W, cout):
    """The fused kernel with a BasicBlock shortcut (ResNet-18 layer1: conv2 + bn2 folded, + identity, ReLU) against
    float64; also that a pooled output refuses a shortcut."""
    torch, _ = T
    from cslam_amd import _lib
    from cslam_amd.vpr.winograd import fused64_weights, wino_fused64, wino_weights
    torch.manual_seed(5)
    w = torch.randn((cout, 64, 3, 3), device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    x = torch.randn((B, 64, H, W), device="cuda").contiguous(memory_format=torch.channels_last)
    idt = torch.randn((B, cout, H, W), device="cuda").contiguous(memory_format=torch.channels_last)
    Up = fused64_weights(wino_weights(w).cuda())
    y = wino_fused64(x, Up, b, True, False, idt)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1) + idt.double())
    assert y.shape == ref.shape
    assert (y.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
    if H % 2 == 0 and W % 2 == 0:
        with pytest.raises(_lib.CslamHipError):
            wino_fused64(x, Up, b, True, True, idt[:, :, ::2, ::2])


@pytest.mark.parametrize("form", ["h", "f32"])
@pytest.mark.parametrize("cout", [64, 128])
@pytest.mark.parametrize("B,H,W,relu,pool,bias", [(2, 224, 224, True, True, True), (3, 37, 50, True, False, True),
                                                  (300, 16, 16, False, False, False), (1, 8, 90, True, True, False)])
def test_fused_winograd_f4_equals_float64(T, form, cout, B, H, W, relu, pool, bias, monkeypatch):
    """The F(4x4,3x3) one-kernel forms -- "f32": f32-input MFMA (csrc/wino_fused.hip `wino4_fused_c64_pipe_kernel`); "h":
    exact fp16 pairs on the fp16 matrix pipe (csrc/wino_fused_h.hip, the default) -- against a float64 conv2d (+ ReLU +
    MaxPool2d) at the F(4x4) tolerance of the three-kernel form (2e-5 of the largest activation, 5e-6 in the relative
    2-norm); ragged 16 x 16 blocks, odd block counts per row (the fp16 form takes two blocks per step at 64 output
    channels), block counts below and above the compute-unit count, both output widths."""
    torch, _ = T
    from torch import nn
    from cslam_amd.vpr.winograd import WinogradTrunk
    from cslam_amd.vpr import winograd as wgm
    monkeypatch.setitem(wgm.TRUNK_FORMS, "fused_h", form == "h")
    torch.manual_seed(29)
    mods = [nn.Conv2d(64, cout, 3, padding=1, bias=bias)] + ([nn.ReLU()] if relu else []) + \
        ([nn.MaxPool2d(2, 2)] if pool and relu else [])
    seq = nn.Sequential(*mods).cuda().eval()
    x = torch.randn((B, 64, H, W), device="cuda")
    fused = WinogradTrunk(seq, 64, 4, fused64=True)
    fused.fused_min_blocks = 0
    assert fused.steps[0].Up is not None and fused.steps[0].Up.shape[1] == 36
    assert (fused.steps[0].Uph is not None) == (form == "h")
    yf = fused(x)
    with torch.no_grad():
        ref = seq.double()(x.double())
    seq.float()
    assert yf.shape == ref.shape
    ef = (yf.double() - ref).abs().max().item() / ref.abs().max().item()
    rf = float(((yf.double() - ref) ** 2).sum().sqrt() / (ref ** 2).sum().sqrt())
    assert ef <= 2e-5 and rf <= 5e-6, (ef, rf)


@pytest.mark.parametrize("cin", [64, 128])
@pytest.mark.parametrize("B,H,W,relu,pool,bias,amp", [(2, 112, 112, True, True, True, 1.0), (3, 37, 50, True, False, True, 1e3),
                                                      (300, 16, 16, False, False, False, 1.0), (1, 8, 90, True, True, False, 1e-3),
                                                      (5, 20, 34, False, True, True, 1.0)])
def test_direct_conv_on_fp16_pairs_equals_float64(T, cin, B, H, W, relu, pool, bias, amp):
    """The direct one-kernel convolution of the 128-output-channel layers (csrc/conv_direct_h.hip: VGG-16 conv2_1, conv2_2) against
    a float64 conv2d (+ ReLU + MaxPool2d): fp32-grade -- no Winograd transform in it, so the bar is that of an fp32 convolution,
    3 x tighter than the F(4x4) forms' -- on ragged 16 x 16 blocks, block counts below and above the compute-unit count,
    activations six decades apart (power-of-two scale from the max |x| slot), with the max |y| it leaves for the next layer."""
    torch, _ = T
    from cslam_amd import _lib
    from cslam_amd.vpr import winograd as wg
    lib = _lib.load()
    torch.manual_seed(101 + cin)
    x = (torch.randn(B, cin, H, W, device="cuda") * amp).contiguous(memory_format=torch.channels_last)
    w = torch.randn(128, cin, 3, 3, device="cuda") / (3.0 * cin ** 0.5)
    b = torch.randn(128, device="cuda") * amp if bias else None
    Wd = wg.direct_pair_weights(w)
    assert tuple(Wd[0].shape) == (9, 128, cin // 32, 2, 32) and Wd[0].dtype == torch.float16
    slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    _lib.check(lib.cslam_absmax_dev(x.data_ptr(), x.numel(), slot.data_ptr(), torch.cuda.current_stream().cuda_stream))
    out_slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    y = wg.conv3x3_direct_h(x, Wd, b, relu, pool, slot, out_slot)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2)             # max and ReLU commute
    if relu:
        ref = torch.relu(ref)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    ef = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    rf = float(((y.double() - ref) ** 2).sum().sqrt() / (ref ** 2).sum().sqrt())
    assert ef <= 5e-6 and rf <= 2e-6, (ef, rf)
    assert out_slot.item() == y.abs().max().item()


@pytest.mark.parametrize("cin,pool", [(128, True), (64, False)])
def test_direct_conv_beside_a_stream_that_thrashes_the_l2_is_bit_identical(T, cin, pool):
    """The direct kernel's stage barriers count outstanding requests (`s_waitcnt vmcnt(N)`), which is only right if every wave
    issues the same requests per stage.  Round 4's first form loaded the patch under `if (inside)`: waves whose lanes were all
    outside skipped the load, waited for one request too few, and -- once the weights were slow to arrive because another stream
    streamed 256 MB through the L2 -- multiplied a ring slot that had not landed (errors up to 0.7 in whole channel groups; alone on
    the GPU the weights are L2 hits and the race never showed).  Same launch alone and beside such a stream: bit-identical."""
    torch, _ = T
    from cslam_amd.vpr import winograd as wg
    torch.manual_seed(0)
    B, H, W = 12, 188, 188                                        # a 12-frame chunk at conv2's resolution: 1728 blocks, ragged edges
    x = torch.relu(torch.randn(B, cin, H, W, device="cuda")).contiguous(memory_format=torch.channels_last)
    w = torch.randn(128, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    b = torch.randn(128, device="cuda")
    Wd = wg.direct_pair_weights(w)
    slot = x.abs().max().reshape(1).clone()
    ref = wg.conv3x3_direct_h(x, Wd, b, True, pool, slot)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.zeros(64 << 20, device="cuda")
    for _ in range(6):
        with torch.cuda.stream(s2):
            for _ in range(4):
                big.add_(1.0)
        with torch.cuda.stream(s1):
            ys = [wg.conv3x3_direct_h(x, Wd, b, True, pool, slot) for _ in range(3)]
        torch.cuda.synchronize()
        for y in ys:
            assert torch.equal(y, ref)


def test_pair_convolution_entry_points_reject_what_they_do_not_serve(T):
    """Error behaviour of the round-6 entry points: shapes outside the kernels' range, a missing bound slot, a float32 input with a
    shortcut -- CslamHipError (CSLAM_E_INVALID) with a message, nothing launched."""
    torch, _ = T
    import ctypes as C
    from cslam_amd import _lib
    lib = _lib.load()
    z = torch.zeros(64, device="cuda")
    pz = C.c_void_p(z.data_ptr())
    with pytest.raises(_lib.CslamHipError, match="Cin and Cout must be 64"):
        _lib.check(lib.cslam_conv3x3_direct_p_dev(pz, 1, pz, pz, None, None, 0, None, 1, 8, 8, 32, 64, 1, pz, 1.0, 1.0, 0.0, None, 1, pz, pz, None))
    with pytest.raises(_lib.CslamHipError, match="bound slot"):
        _lib.check(lib.cslam_conv3x3_direct_p_dev(pz, 1, pz, pz, None, None, 0, None, 1, 8, 8, 64, 64, 1, pz, 1.0, 1.0, 0.0, None, 1, None, pz, None))
    with pytest.raises(_lib.CslamHipError, match="shortcut"):
        _lib.check(lib.cslam_conv3x3_direct_p_dev(pz, 0, None, pz, None, pz, 0, pz, 1, 8, 8, 64, 64, 1, pz, 1.0, 1.0, 0.0, None, 1, pz, pz, None))
    with pytest.raises(_lib.CslamHipError, match="NULL"):
        _lib.check(lib.cslam_conv3x3_direct_p_dev(None, 1, pz, pz, None, None, 0, None, 1, 8, 8, 64, 64, 1, pz, 1.0, 1.0, 0.0, None, 1, pz, pz, None))
    with pytest.raises(_lib.CslamHipError, match="pair-format output"):
        _lib.check(lib.cslam_conv3x3_direct_r_pairs_dev(pz, pz, None, 1, 8, 8, 64, 128, pz, 1.0, 1.0, 0.0, None, None, pz, None))
    with pytest.raises(_lib.CslamHipError, match="Cout must be 128"):
        _lib.check(lib.cslam_conv3x3_direct_hp_dev(pz, pz, pz, None, 1, 8, 8, 128, 64, 1, 0, 1.0, None, pz, None))


def test_cosplace_extraction_beside_a_stream_that_thrashes_the_l2_is_bit_identical(T):
    """CosPlace's whole extract pass (patch-form stem, the register-resident direct kernel of layer1 with its LDS-DMA patches and
    deferred epilogue, the implicit GEMM's DMA rings, pair-format maps, GeM head) alone and beside a stream that keeps HBM and the L2
    busy: the same descriptors bit for bit (every kernel of the pass either drains its requests with vmcnt(0) or counts requests that
    every wave issues)."""
    torch, heads = T
    from cslam_amd.vpr.cosplace import CosPlace
    cp = CosPlace({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.cosplace.descriptor_dim": 512,
                   "frontend.cosplace.backbone": "resnet18"}, None)
    gen = torch.Generator(device="cuda").manual_seed(9)
    frames = torch.randint(0, 256, (300, 240, 320, 3), generator=gen, device="cuda", dtype=torch.uint8)
    ref = cp.compute_embeddings_device(frames)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.zeros(64 << 20, device="cuda")
    for _ in range(4):
        with torch.cuda.stream(s2):
            for _ in range(6):
                big.add_(1.0)
        with torch.cuda.stream(s1):
            got = cp.compute_embeddings_device(frames)
        torch.cuda.synchronize()
        assert torch.equal(got, ref)


def test_netvlad_extraction_beside_a_stream_that_thrashes_the_l2_is_bit_identical(T):
    """The whole extract pass (stem kernel, direct kernels, pair products, transforms, heads: every kernel with counted waits or
    LDS-DMA rings) beside a stream that keeps HBM and the L2 busy: the descriptors of the pass alone, bit for bit."""
    torch, heads = T
    from cslam_amd.vpr.netvlad import NetVLAD
    nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 256}, None)
    gen = torch.Generator(device="cuda").manual_seed(5)
    frames = torch.randint(0, 256, (24, 240, 320, 3), generator=gen, device="cuda", dtype=torch.uint8)
    ref = nv.compute_embeddings_device(frames)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.zeros(64 << 20, device="cuda")
    for _ in range(4):
        with torch.cuda.stream(s2):
            for _ in range(12):
                big.add_(1.0)
        with torch.cuda.stream(s1):
            got = nv.compute_embeddings_device(frames)
        torch.cuda.synchronize()
        assert torch.equal(got, ref)


def test_trunk_runs_conv2_1_and_conv2_2_through_the_direct_kernel_and_matches_the_winograd_forms(T, monkeypatch):
    """VGG-16's first two blocks through the trunk runner: by default the stem and conv2_1 take the register-resident direct kernels
    and conv2_2 the register-resident kernel on output-channel halves; forms conv_direct = 2 leaves conv2_1 on the one-kernel F(4x4)
    form, = 0 keeps round 3's F(4x4) forms for both; conv_direct_r / conv_direct_r2 / stem_direct = False select the A/B partners of
    the register-resident kernels (conv2_2: the weights through an LDS ring).  All
    against float64, and against each other at the F(4x4) forms' tolerance."""
    torch, _ = T
    from torch import nn
    from cslam_amd.vpr.winograd import WinogradTrunk
    torch.manual_seed(7)
    seq = nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(), nn.MaxPool2d(2, 2),
                        nn.Conv2d(64, 128, 3, padding=1), nn.ReLU(), nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(),
                        nn.MaxPool2d(2, 2)).cuda().eval()
    x = torch.randn((36, 3, 64, 80), device="cuda")
    outs = {}
    for tag, env, direct in (("direct", 1, (True, True)), ("conv2_2", 2, (False, True)), ("wino", 0, (False, False))):
        t = WinogradTrunk(seq, 64, 4, fused64=True, forms={"conv_direct": env})
        assert ((t.steps[2].Wd is not None), (t.steps[3].Wd is not None)) == direct, tag
        assert (t.steps[2].Wdr is not None) == direct[0] and t.steps[3].Wdr is None and t.steps[0].Wr is not None
        outs[tag] = t(x)
    # the A/B partners of the register-resident kernels: conv2_1 on the streaming direct kernel, the F(4x4) stem kernel
    for tag, env in (("conv2_1 streaming", "conv_direct_r"), ("wino stem", "stem_direct"), ("conv2_2 lds ring", "conv_direct_r2")):
        t = WinogradTrunk(seq, 64, 4, fused64=True, forms={env: False})
        assert (t.steps[2].Wdr is None) == (env == "conv_direct_r") and (t.steps[0].Wr is None) == (env == "stem_direct"), tag
        assert (t.steps[3].Wdr2 is None) == (env == "conv_direct_r2")
        assert t.steps[2].Wd is not None and t.steps[0].stem is not None
        outs[tag] = t(x)
    with torch.no_grad():
        ref = seq.double()(x.double())
    seq.float()
    for tag in outs:
        assert _rel_rms(outs[tag], ref) <= 1e-5, tag
    for tag in ("direct", "conv2_2", "conv2_1 streaming", "wino stem", "conv2_2 lds ring"):
        assert (outs[tag] - outs["wino"]).abs().max().item() <= 2e-5 * ref.abs().max().item(), tag


@pytest.mark.parametrize("B,H,W,pool,amp", [(20, 32, 48, True, 1.0), (3, 112, 112, True, 30.0), (17, 22, 38, False, 1e-3), (33, 16, 16, True, 1.0)])
def test_conv2_1_writes_pairs_and_conv2_2_stages_them_without_conversion(T, B, H, W, pool, amp):
    """`cslam_conv3x3_direct_r_pairs_dev` -> `cslam_conv3x3_direct_hp_dev` (VGG-16 conv2_1 -> conv2_2 with the map between them in pair
    format: the second kernel's patch goes to LDS as it is) against the float32 chain of the same two kernels and against float64 --
    no further from it than 4 x torch's float32 chain; the pair tensor itself decodes to conv2_1's output; the bound slot is
    max|x| wl1 + bmax; through `WinogradTrunk` with `VGG_PAIRS` (off by default: measured no faster) and without; repeated calls are bit-identical."""
    torch, _ = T
    from torch import nn
    from cslam_amd.vpr import winograd as wg
    torch.manual_seed(B + H)
    x = (torch.relu(torch.randn((B, 64, H, W), device="cuda")) * amp).contiguous(memory_format=torch.channels_last)
    c1, c2 = nn.Conv2d(64, 128, 3, padding=1).cuda(), nn.Conv2d(128, 128, 3, padding=1).cuda()
    with torch.no_grad():
        c1.bias.mul_(amp * 0.5)
        c2.bias.mul_(amp * 0.5)
        r1 = torch.relu(torch.nn.functional.conv2d(x.double(), c1.weight.double(), c1.bias.double(), padding=1))
        r2 = torch.relu(torch.nn.functional.conv2d(r1, c2.weight.double(), c2.bias.double(), padding=1))
        f2 = torch.relu(c2(torch.relu(c1(x))))
        if pool:
            r2, f2 = torch.nn.functional.max_pool2d(r2, 2, 2), torch.nn.functional.max_pool2d(f2, 2, 2)
    slots = torch.zeros(6, device="cuda")
    slots[0] = x.abs().max()
    Wr, Wd = wg.direct_r_pair_weights(c1.weight), wg.direct_pair_weights(c2.weight)
    wl1, bmax = float(c1.weight.detach().abs().sum(dim=(1, 2, 3)).max()), float(c1.bias.detach().abs().max())
    xp = wg.conv3x3_direct_r_pairs(x, Wr, c1.bias.detach(), wl1, bmax, slots[0:1], slots[1:2], slots[2:3])
    act = wg.PairAct(xp, True, (B, 128, H, W), slots[2:3], slots[1:2])
    y1 = wg.pairs_to_float(act)
    sc1 = r1.abs().max().item()
    assert (y1.double() - r1).abs().max().item() / sc1 <= 3e-6
    assert abs(slots[2].item() - sc1) <= 1e-5 * sc1 and slots[2].item() <= slots[1].item() <= (slots[0].item() * wl1 + bmax) * 1.002
    y2 = wg.conv3x3_direct_hp(xp, (B, 128, H, W), slots[1:2], Wd, c2.bias.detach(), True, pool, slots[3:4])
    sc2 = r2.abs().max().item()
    e32 = (f2.double() - r2).abs().max().item() / sc2
    assert tuple(y2.shape) == tuple(r2.shape) and y2.is_contiguous(memory_format=torch.channels_last)
    assert (y2.double() - r2).abs().max().item() / sc2 <= 4 * e32 + 4e-7
    assert slots[3].item() >= y2.abs().max().item() * (1 - 1e-6)
    # the float32 chain of the same kernels
    y1f = wg.conv3x3_direct_r(x, Wr, c1.bias.detach(), True, False, slots[0:1], slots[4:5])
    y2f = wg.conv3x3_direct_h(y1f, Wd, c2.bias.detach(), True, pool, slots[4:5], None)
    assert (y2 - y2f).abs().max().item() / sc2 <= 2e-6
    assert torch.equal(y2, wg.conv3x3_direct_hp(wg.conv3x3_direct_r_pairs(x, Wr, c1.bias.detach(), wl1, bmax, slots[0:1], slots[5:6]),
                                                (B, 128, H, W), slots[5:6], Wd, c2.bias.detach(), True, pool, None))
    # through the trunk runner
    mods = [c1, nn.ReLU(), c2, nn.ReLU()] + ([nn.MaxPool2d(2, 2)] if pool else [])
    seq = nn.Sequential(*mods).eval()
    t = wg.WinogradTrunk(seq, 64, 4, forms={"conv_direct_r2": False})          # conv2_2 with its weights through the LDS ring: the kernels above
    t.fused_min_blocks = 0
    assert t.steps[0].Wdr is not None and t.steps[1].Wd is not None and t.steps[1].Wdr is None and t.steps[1].Wdr2 is None
    gotf = t(x)
    try:
        wg.VGG_PAIRS = True
        got = t(x)
    finally:
        wg.VGG_PAIRS = False
    assert torch.equal(got, y2) and torch.equal(gotf, y2f)
    t2 = wg.WinogradTrunk(seq, 64, 4)                                           # the default: conv2_2 register-resident on output-channel halves
    t2.fused_min_blocks = 0
    assert t2.steps[1].Wdr2 is not None
    assert (t2(x) - y2f).abs().max().item() / sc2 <= 2e-6


@pytest.mark.parametrize("cout,B,H,W,pool,amp", [(64, 4, 64, 48, True, 1.0), (128, 3, 40, 56, False, 1e3), (64, 2, 30, 22, False, 1e-3)])
def test_fused_winograd_h_scales_shortcut_and_amax(T, cout, B, H, W, pool, amp):
    """cslam_wino4_fused_c64_h_dev called directly: activations six decades apart (the power-of-two scale from the max |x|
    slot), the max |y| slot it writes (a bound of the pre-pool maximum, as the output transform's), the shortcut add of the
    non-pooled form, and max |y| out of the first-layer kernel feeding it."""
    torch, _ = T
    from cslam_amd import _lib
    from cslam_amd.vpr import winograd as wg
    lib = _lib.load()
    import ctypes as C
    torch.manual_seed(41)
    x = (torch.relu(torch.randn(B, 64, H, W, device="cuda")) * amp).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, 64, 3, 3, device="cuda") / 24.0
    b = torch.randn(cout, device="cuda") * amp
    res = None if pool else (torch.randn(B, cout, H, W, device="cuda") * amp).contiguous(memory_format=torch.channels_last)
    U4 = wg.wino_weights(w, 4).cuda()
    Uh = wg.fused64_pair_weights(U4)
    Uh = (Uh[0].cuda(), Uh[1])
    st = torch.cuda.current_stream().cuda_stream
    slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    _lib.check(lib.cslam_absmax_dev(C.c_void_p(x.data_ptr()), x.numel(), C.c_void_p(slot.data_ptr()), st))
    out_slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    y = wg.wino_fused64_h(x, Uh, b, True, pool, slot, out_slot, res)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    if res is not None:
        ref = ref + res.double()
    ref = torch.relu(ref)
    prepool = ref
    ref = torch.nn.functional.max_pool2d(ref, 2, 2) if pool else ref
    s = ref.abs().max().item()
    assert (y.double() - ref).abs().max().item() <= 2e-5 * s
    assert float(((y.double() - ref) ** 2).sum().sqrt() / (ref ** 2).sum().sqrt()) <= 5e-6
    m = prepool.abs().max().item()
    assert m * (1 - 1e-4) <= out_slot.item() <= m * (1 + 1e-4)
    # first-layer kernel: max |y| delivered with the convolution
    x3 = torch.randn(B, 3, H, W, device="cuda") * amp
    w3 = torch.randn(64, 3, 3, 3, device="cuda") / 5.0
    b3 = torch.randn(64, device="cuda") * amp
    wt = w3.permute(1, 2, 3, 0).reshape(27, 64).contiguous()
    y3 = torch.empty((B, 64, H, W), device="cuda", memory_format=torch.channels_last)
    s3 = torch.zeros(1, dtype=torch.float32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
    _lib.check(lib.cslam_conv3x3_c3_amax_dev(p(x3), p(wt), p(b3), B, H, W, 64, 1, p(y3), p(s3), st))
    r3 = torch.relu(torch.nn.functional.conv2d(x3.double(), w3.double(), b3.double(), padding=1))
    assert (y3.double() - r3).abs().max().item() <= 1e-5 * r3.abs().max().item()
    assert s3.item() == y3.abs().max().item()


def _rel_rms(y, ref):
    """||y - ref||_2 / ||ref||_2 in float64: unlike the max-norm bar it does not let small activations hide."""
    return float(((y.double() - ref) ** 2).sum().sqrt() / (ref ** 2).sum().sqrt())


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["pair", "h3"])
@pytest.mark.parametrize("B,H,W,cin,cout,relu,pool,amp", [
    (8, 56, 56, 128, 256, True, False, 1.0), (8, 56, 56, 128, 256, True, True, 1e3), (32, 14, 14, 512, 512, True, False, 1e-3),
    (16, 28, 28, 256, 512, False, False, 1.0), (40, 13, 15, 256, 256, True, False, 1.0),
    (57, 12, 12, 256, 384, True, False, 1.0)])     # 513 tiles x 192 channel pairs: the last workgroup of the output transform is part empty
def test_split16_winograd_layer_is_fp32_grade(T, form, B, H, W, cin, cout, relu, pool, amp):
    """Split-fp16 forms of the 36 GEMMs -- "pair": this library's GEMM over exact hi/lo pairs (csrc/wino_gemm.hip,
    `split16_pair_weights`, the default); "h3": round 1's library GEMM over [vh | vl | vh] (`split16_weights`) --
    against a float64 convolution: the error is that of the plain-fp32 three-kernel form (within 1.5x, in the max norm
    AND in the relative 2-norm over all activations, and inside the F(4x4) tolerance of 2e-5 of the largest activation /
    5e-6 relative 2-norm), at activation magnitudes six decades apart (the power-of-two scale), with the ragged last
    tile row / column, and the max |y| slot written by the output transform is a bound."""
    torch, _ = T
    from cslam_amd.vpr import winograd as wg
    torch.manual_seed(31)
    x = (torch.relu(torch.randn(B, cin, H, W, device="cuda")) * amp).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    b = torch.randn(cout, device="cuda") * amp
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = torch.relu(ref) if relu else ref
    prepool = ref
    ref = torch.nn.functional.max_pool2d(ref, 2, 2) if pool else ref
    ws = wg._Workspace()
    U, U4 = wg.wino_weights(w).cuda(), wg.wino_weights(w, 4).cuda()
    if form == "h3":
        U3 = wg.split16_weights(U4)
        rec = (U3[0][:, :cin].double() + U3[0][:, 2 * cin:].double()) * U3[1]
        assert (rec - U4.double()).abs().max().item() <= 2.0 ** -21 * U4.abs().max().item()
        assert torch.equal(U3[0][:, :cin], U3[0][:, cin:2 * cin])
        kw = {"U3": U3}
    else:
        kw = {"U2": wg.split16_pair_weights(U4)}
    slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    y32 = wg.wino_conv3x3(ws, x, U, U4, b, relu, pool)
    y16 = wg.wino_conv3x3(ws, x, U, U4, b, relu, pool, amax_out=slot, **kw)
    assert ws.amax_written
    s = ref.abs().max().item()
    e32 = (y32.double() - ref).abs().max().item() / s
    e16 = (y16.double() - ref).abs().max().item() / s
    assert e16 <= 2e-5 and e16 <= 1.5 * e32 + 1e-7, (e16, e32)
    r32, r16 = _rel_rms(y32, ref), _rel_rms(y16, ref)
    assert r16 <= 5e-6 and r16 <= 1.5 * r32 + 1e-8, (r16, r32)
    bound = slot.item()
    assert prepool.abs().max().item() * (1 - 1e-4) <= bound <= prepool.abs().max().item() * (1 + 1e-4)
    # the slot as the next call's amax_in gives the same result as the separate pass over x (same scale either way
    # whenever the bound and the exact maximum fall into the same power-of-two bucket; here they are the same number)
    if not pool:
        xs = y16
        w2 = torch.randn(cout, cout, 3, 3, device="cuda") / (3 * cout ** 0.5)
        V2, V4 = wg.wino_weights(w2).cuda(), wg.wino_weights(w2, 4).cuda()
        kw2 = {"U3": wg.split16_weights(V4)} if form == "h3" else {"U2": wg.split16_pair_weights(V4)}
        ya = wg.wino_conv3x3(ws, xs, V2, V4, None, True, False, **kw2)
        yb = wg.wino_conv3x3(ws, xs, V2, V4, None, True, False, amax_in=slot, **kw2)
        r2 = torch.relu(torch.nn.functional.conv2d(xs.double(), w2.double(), None, padding=1))
        s2 = r2.abs().max().item()
        assert (ya.double() - r2).abs().max().item() <= 2e-5 * s2
        assert (yb.double() - r2).abs().max().item() <= 2e-5 * s2
        assert _rel_rms(ya, r2) <= 5e-6 and _rel_rms(yb, r2) <= 5e-6


@pytest.mark.gpu
def test_split16_trunk_equals_fp32_gemm_trunk(T):
    """VGG-16 trunk with the split-fp16 GEMMs -- the default pair form from 128 input channels on (conv2_2 ... conv5_3,
    this library's GEMM) and round 1's h3 form from 256 on (split16_h3=True) -- against the same trunk on plain fp32
    GEMMs (forms split16_min_cin = 0) and against a float64 evaluation: neither split form is less accurate than the fp32 one
    by more than 1.5x (max norm and relative 2-norm), and all sit inside the trunk tolerance used for the fp32 form."""
    torch, _ = T
    from cslam_amd.vpr.backbones import vgg16_features_trunk
    from cslam_amd.vpr.winograd import WinogradTrunk
    torch.manual_seed(37)
    enc = vgg16_features_trunk().cuda().eval()
    x = torch.randn((32, 3, 224, 224), device="cuda")     # 32 frames: conv5_x (4 x 4 tiles per frame) reaches the 512-tile F(4x4) floor
    t32 = WinogradTrunk(enc, 64, 4, forms={"split16_min_cin": 0})
    t2 = WinogradTrunk(enc, 64, 4)
    t3 = WinogradTrunk(enc, 64, 4, split16_h3=True)
    assert all(st.U3 is None and st.U2 is None for st in t32.steps)
    assert sum(st.U2 is not None for st in t2.steps) == 10 and all(st.U3 is None for st in t2.steps)    # conv2_2 ... conv5_3
    assert sum(st.U3 is not None for st in t3.steps) == 8 and all(st.U2 is None for st in t3.steps)     # conv3_2 ... conv5_3
    y32, y2, y3 = t32(x), t2(x), t3(x)
    with torch.no_grad():
        ref = enc.double()(x.double())
    enc.float()
    s = ref.abs().max().item()
    e32, e2, e3 = ((y.double() - ref).abs().max().item() / s for y in (y32, y2, y3))
    assert e32 <= 2e-5 and e2 <= 2e-5 and e3 <= 2e-5 and max(e2, e3) <= 1.5 * e32 + 1e-7, (e2, e3, e32)
    r32, r2, r3 = _rel_rms(y32, ref), _rel_rms(y2, ref), _rel_rms(y3, ref)
    assert r32 <= 1e-5 and max(r2, r3) <= 1.5 * r32 + 1e-8, (r2, r3, r32)


@pytest.mark.gpu
@pytest.mark.parametrize("direct", [True, False])
@pytest.mark.parametrize("B,H,W,pool,amp", [(2, 224, 224, True, 1.0), (3, 37, 50, False, 1.0), (5, 64, 96, True, 1e-3),
                                            (300, 16, 32, True, 1.0), (2, 30, 22, False, 40.0), (1, 18, 34, True, 1.0),
                                            (7, 8, 16, True, 1.0), (1, 2, 2, True, 1.0), (40, 226, 230, True, 0.05)])
def test_stem_kernel_first_two_convolutions_equal_float64_and_separate_kernels(T, B, H, W, pool, amp, direct, monkeypatch):
    """cslam_conv_stem_direct_h_dev (direct = True: conv 3 -> 64 + ReLU + conv 64 -> 64 + ReLU (+ MaxPool2d) as one DIRECT kernel
    whose second-layer weights stay in registers, csrc/conv_stem_direct_h.hip) and cslam_wino4_stem_c64_h_dev (the one-kernel
    F(4x4) form of the same pair); VGG-16 conv1_1 / conv1_2, cslam/vpr/netvlad.py:163-171: against a float64 evaluation of the
    two layers at the tolerance of the one-kernel form, and against the two separate kernels (fp32-grade: within 2e-6 of the
    largest activation).  Ragged blocks, maps narrower / lower than one block, single-block maps, block counts below / above the
    compute-unit count (several blocks per persistent workgroup), image scales."""
    from cslam_amd.vpr import winograd as wgm
    monkeypatch.setitem(wgm.TRUNK_FORMS, "stem_direct", bool(direct))
    torch, _ = T
    from torch import nn
    from cslam_amd.vpr.winograd import WinogradTrunk
    torch.manual_seed(57)
    mods = [nn.Conv2d(3, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 64, 3, padding=1), nn.ReLU()] + \
        ([nn.MaxPool2d(2, 2)] if pool else [])
    seq = nn.Sequential(*mods).cuda().eval()
    x = (torch.rand((B, 3, H, W), device="cuda") * 4.8 - 2.2) * amp       # the range of a normalised image
    monkeypatch.setitem(wgm.TRUNK_FORMS, "wino_stem", True)
    stem = WinogradTrunk(seq, 64, 4, fused64=True)
    stem.fused_min_blocks = 0
    assert stem.steps[0].stem is not None and stem.steps[1].Uph is not None and (stem.steps[0].Wr is not None) == direct
    ys = stem(x)
    monkeypatch.setitem(wgm.TRUNK_FORMS, "wino_stem", False)
    apart = WinogradTrunk(seq, 64, 4, fused64=True)
    apart.fused_min_blocks = 0
    assert apart.steps[0].stem is None
    ya = apart(x)
    with torch.no_grad():
        ref = seq.double()(x.double())
    seq.float()
    assert ys.shape == ref.shape == ya.shape
    top = ref.abs().max().item()
    es = (ys.double() - ref).abs().max().item() / top
    rs = float(((ys.double() - ref) ** 2).sum().sqrt() / (ref ** 2).sum().sqrt())
    ea = (ya.double() - ref).abs().max().item() / top
    ra = float(((ya.double() - ref) ** 2).sum().sqrt() / (ref ** 2).sum().sqrt())
    assert es <= 2e-5 and rs <= 5e-6, (es, rs)
    assert es <= max(2.0 * ea, 2e-6) and rs <= max(2.0 * ra, 5e-7), (es, ea, rs, ra)
    assert (ys - ya).abs().max().item() <= 1e-5 * top


@pytest.mark.gpu
@pytest.mark.parametrize("B,din,dout,whiten", [(130, 32768, 4096, False), (256, 32768, 4096, True), (33, 2048, 256, False),
                                               (700, 8192, 128, True)])
def test_pca_project_on_fp16_pairs_is_fp32_grade(T, B, din, dout, whiten):
    """cslam_pca_project_pairs_dev (the batched projection on the fp16 matrix pipe: x and the components as exact hi / lo
    pairs, the pair GEMM of the trunk with the K splits in place of its 36 frequencies) against a float64 evaluation and
    against the f32-MFMA form: within 1e-5 absolute on unit-norm outputs (actual ~1e-7), no further from float64 than 2x
    the f32 form; ragged batches, mean / whitening terms, inputs two decades apart (the power-of-two input scale)."""
    torch, heads = T
    gen = torch.Generator(device="cuda").manual_seed(B + dout)
    x = torch.randn((B, din), generator=gen, device="cuda")
    x[B // 2:] *= 0.01
    comp = torch.randn((dout, din), generator=gen, device="cuda") / din ** 0.5
    mp = torch.randn(dout, generator=gen, device="cuda") * 0.1
    inv = (torch.rand(dout, generator=gen, device="cuda") + 0.5) if whiten else None
    pairs = heads.pca_pair_weights(comp)
    assert pairs is not None and pairs[0].shape[0] == heads.PCA_PAIR_SPLITS
    yp = heads.pca_project(x, comp, mp, inv, pairs)
    yf = heads.pca_project(x, comp, mp, inv)
    ref = x.double() @ comp.double().T - mp.double()
    if inv is not None:
        ref = ref * inv.double()
    ref = ref / ref.norm(dim=1, keepdim=True)
    ep, ef = float((yp.double() - ref).abs().max()), float((yf.double() - ref).abs().max())
    assert ep < 1e-5 and ep <= max(2.0 * ef, 5e-7), (ep, ef)
    assert float((yp.norm(dim=1) - 1.0).abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W", [(12, 512, 14, 14), (9, 96, 16, 16), (40, 500, 5, 3), (33, 64, 1, 1), (11, 512, 7, 5),
                                     (10, 256, 14, 14), (9, 128, 8, 4), (17, 384, 10, 20), (9, 512, 15, 14)])
def test_vlad_channels_last_feature_map_equals_nchw(T, B, C, H, W):
    """cslam_vlad_aggregate_nhwc_dev: the batch kernels on the channels_last map the Winograd trunk writes, against the NCHW
    kernel on the converted map and the numpy restatement of NetVLADLayer.forward (netvlad.py:94-130).  C a multiple of 128 (up
    to 512) with 32 <= P <= 200 -- NetVLAD's own 512 x 14 x 14 among them -- runs the two contractions on the f32 matrix pipe
    (vlad_mfma_kernel: another summation order, so a tolerance; P odd, P = 200, one to four slabs); the other shapes run the VALU
    kernel in both layouts (same arithmetic in the same order: bit-identical): ragged last slab (C not a multiple of 32), P = 1,
    P above the matrix form's limit."""
    torch, heads = T
    rng = np.random.default_rng(B + C)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = rng.standard_normal((64, C)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    c = rng.random((64, C)).astype(np.float32)
    xd = dev(T, x)
    xl = xd.contiguous(memory_format=torch.channels_last)
    assert not xl.is_contiguous() or H * W == 1
    y_nchw = heads.vlad_aggregate(xd, dev(T, w), dev(T, b), dev(T, c))
    y_nhwc = heads.vlad_aggregate(xl, dev(T, w), dev(T, b), dev(T, c))
    matrix_form = C % 128 == 0 and C <= 512 and 32 <= H * W <= 200
    if matrix_form:
        assert float((y_nchw - y_nhwc).abs().max()) < 2e-7
    else:
        assert torch.equal(y_nchw, y_nhwc)
    assert np.max(np.abs(y_nhwc.cpu().numpy() - ho.vlad_forward(x, w, b, c))) < 1e-6


@pytest.mark.gpu
def test_vlad_matrix_form_against_float64_and_its_valu_partner(T, monkeypatch):
    """vlad_mfma_kernel on trunk-like input (non-negative, a few dead channels, one all-zero pixel, one all-zero frame) with the
    trained layer's scales (assignment weights 2 alpha centroids, netvlad.py:62-71): no further from the float64 restatement than
    the VALU kernel is, unit norm, finite on the zero frame."""
    torch, heads = T
    rng = np.random.default_rng(3)
    B, C, H, W = 64, 512, 14, 14
    x = np.maximum(rng.standard_normal((B, C, H, W)), 0).astype(np.float32) * rng.random((1, C, 1, 1)).astype(np.float32) * 30
    x[:, ::37] = 0
    x[3, :, 5, 5] = 0
    x[7] = 0
    cent = rng.random((64, C)).astype(np.float32)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    w = (2.0 * 100.0 * cent).astype(np.float32)
    b = (-100.0 * np.linalg.norm(cent, axis=1)).astype(np.float32)
    xl = dev(T, x).contiguous(memory_format=torch.channels_last)
    y = heads.vlad_aggregate(xl, dev(T, w), dev(T, b), dev(T, cent)).cpu().numpy()
    y_valu = heads.vlad_aggregate(dev(T, x), dev(T, w), dev(T, b), dev(T, cent)).cpu().numpy()
    # NetVLADLayer.forward (netvlad.py:94-130) in float64
    xf = x.astype(np.float64).reshape(B, C, -1)
    xf = xf / np.maximum(np.linalg.norm(xf, axis=1, keepdims=True), 1e-12)
    sa = np.einsum("kc,ncp->nkp", w.astype(np.float64), xf) + b.astype(np.float64)[None, :, None]
    a = np.exp(sa - sa.max(axis=1, keepdims=True))
    a /= a.sum(axis=1, keepdims=True)
    v = np.einsum("nkp,ncp->nkc", a, xf) - a.sum(axis=2)[:, :, None] * cent.astype(np.float64)[None]
    v = v / np.maximum(np.linalg.norm(v, axis=2, keepdims=True), 1e-12)
    v = v.reshape(B, -1)
    ref = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-12)
    assert np.all(np.isfinite(y))
    e, ev = np.max(np.abs(y - ref)), np.max(np.abs(y_valu - ref))
    print("vlad matrix form: max error against float64 %.2e (VALU kernel %.2e)" % (e, ev))
    assert e < 1e-5 and e <= max(2.0 * ev, 2e-7), (e, ev)
    keep = [i for i in range(B) if i != 7]
    assert np.max(np.abs(np.linalg.norm(y[keep], axis=1) - 1.0)) < 1e-6


@pytest.mark.gpu
def test_netvlad_batch_path_descriptors_against_float64_model_on_distinct_frames(T):
    """Descriptor-level gate on the path the bench times: config 3's extractor (4096-D projection) on 96 DISTINCT frames
    -- block patterns of every scale, contrast, brightness and noise level, so small activations are covered -- through the batch forms (stem kernel, fp16-pair
    trunk GEMMs, one-kernel convolutions, channels-last VLAD, pair PCA), against the SAME weights evaluated in float64
    (torch, double precision convolutions) from the same preprocessed input.  north_star's gate: |component| error <= 1e-5 on
    unit-norm descriptors."""
    torch, _ = T
    import copy
    import torch.nn.functional as Fn
    from cslam_amd.vpr import heads
    from cslam_amd.vpr.netvlad import NetVLAD
    rng = np.random.default_rng(11)
    imgs = []
    for i in range(96):                                        # block patterns of every scale, contrast, brightness and noise level
        bsz = (6, 10, 16, 24, 40, 64, 96, 160)[i % 8]
        low = rng.random((-(-480 // bsz), -(-640 // bsz), 3)).astype(np.float32)
        img = np.kron(low, np.ones((bsz, bsz, 1), dtype=np.float32))[:480, :640]
        contrast, bright, sigma = rng.uniform(0.05, 1.0), rng.uniform(20.0, 230.0), rng.uniform(0.0, 40.0) * (i % 3 == 0)
        img = bright + contrast * (img - 0.5) * 255.0 * rng.uniform(0.3, 1.0, size=(1, 1, 3)) + rng.normal(0.0, 1.0, size=img.shape) * sigma
        imgs.append(np.clip(img, 0, 255).astype(np.uint8))
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096,
                  "frontend.random_seed": 3}, None)
    got = nv.compute_embeddings_device(frames).double()
    assert nv.trunk is not None and any(getattr(st, "stem", False) for st in nv.trunk.steps), "the stem kernel did not run"
    assert nv.pca_pairs is not None
    x = heads.preprocess(frames.contiguous(), nv.crop)
    enc = copy.deepcopy(nv.encoder).double()
    W, b, cent = nv.pool.conv_weight.double(), nv.pool.conv_bias, nv.pool.centroids.double()
    comp = nv.pca_components[:4096].double()
    ref = []
    with torch.no_grad():
        for s in range(0, x.shape[0], 16):
            f = enc(x[s:s + 16].double())
            xf = Fn.normalize(f, p=2.0, dim=1).flatten(2)
            sa = torch.einsum("kc,ncp->nkp", W.reshape(W.shape[0], -1), xf)
            if b is not None:
                sa = sa + b.double().reshape(1, -1, 1)
            a = torch.softmax(sa, dim=1)
            v = torch.einsum("nkp,ncp->nkc", a, xf) - a.sum(2)[:, :, None] * cent[None]
            v = Fn.normalize(Fn.normalize(v, p=2.0, dim=2).flatten(1), p=2.0, dim=1)
            y = v @ comp.T - nv.pca_mean_proj.double()[None]
            ref.append(Fn.normalize(y, p=2.0, dim=1))
    ref = torch.cat(ref)
    err = (got - ref).abs()
    worst = err.max().item()
    cos = (got * ref).sum(1)
    # With random weights the descriptors of different images are close (cosine > 0.999: He-initialised ReLU features are
    # dominated by their mean), so the gate above says little about the image-DEPENDENT part.  Two checks that do:
    # (i) the error against the distance of a descriptor from the mean descriptor, (ii) the neighbour ranking among the 96
    # frames -- what the matcher consumes -- is the float64 model's wherever its similarities are 2e-5 apart.
    centred = ref - ref.mean(0, keepdim=True)
    rel = ((got - ref).norm(dim=1) / centred.norm(dim=1)).max().item()
    g_ref = ref @ ref.T
    g_got = got @ got.T
    eye = torch.eye(ref.shape[0], device=ref.device, dtype=torch.bool)
    g_ref.masked_fill_(eye, -2.0); g_got.masked_fill_(eye, -2.0)
    top_ref = torch.topk(g_ref, 4, dim=1)
    top_got = torch.topk(g_got, 4, dim=1)
    clear = (top_ref.values[:, :3] - top_ref.values[:, 1:4]) > 2e-5          # rank r is separated from rank r + 1
    same = top_ref.indices[:, :3] == top_got.indices[:, :3]
    prefix_clear = torch.cumprod(clear.to(torch.int32), dim=1).bool()
    print("descriptor error vs float64: max %.2e, max 1-cos %.1e; pair cosines %.6f .. %.6f, "
          "error / distance from the mean descriptor %.2e; %d of %d top-3 ranks clearly separated, all equal: %s" % (
              worst, float((1.0 - cos).max()), g_ref[~eye].min().item(), g_ref.max().item(), rel, int(prefix_clear.sum()),
              prefix_clear.numel(), bool(same[prefix_clear].all())), flush=True)
    # the same extractor through torch's own fp32 convolutions (MIOpen), the yardstick for "fp32-grade"
    nd = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096,
                  "frontend.random_seed": 3, "frontend.backbone_conv": "direct"}, None)
    e_direct = (nd.compute_embeddings_device(frames).double() - ref).abs().max().item()
    print("   torch fp32 (direct convolutions) against the same float64 model: max %.2e" % e_direct, flush=True)
    assert worst <= 1e-5, worst
    assert worst <= 1.5 * e_direct + 1e-6, (worst, e_direct)
    assert float((1.0 - cos).max()) <= 2e-7
    assert rel <= 2e-2, rel
    assert int(prefix_clear.sum()) >= 150 and bool(same[prefix_clear].all())
    assert float((g_got - g_ref)[~eye].abs().max()) <= 1e-6                  # similarities themselves: well inside the 1e-5 gate


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,relu,pool,bias,amp", [(2, 112, 112, True, False, True, 1.0), (3, 37, 50, True, False, True, 1e3),
                                                      (300, 16, 16, False, False, False, 1.0), (1, 8, 90, True, True, False, 1e-3),
                                                      (5, 20, 34, False, True, True, 1.0), (1, 2, 2, True, True, True, 1.0),
                                                      (40, 114, 118, True, False, True, 0.05)])
def test_register_resident_direct_conv_equals_float64(T, B, H, W, relu, pool, bias, amp):
    """cslam_conv3x3_direct_r_dev (csrc/conv_direct_r.hip: VGG-16 conv2_1, 64 -> 128 channels, the weights register-resident, four waves
    of 32 output channels each) against a float64 conv2d (+ ReLU + MaxPool2d) at the direct kernel's fp32-grade bar, and against the
    streaming direct kernel; ragged 8 x 16 blocks, single-block maps, block counts below / above the compute-unit count (several blocks
    per persistent workgroup: the double-buffered patch), activations six decades apart, the max |y| slot."""
    torch, _ = T
    from cslam_amd import _lib
    from cslam_amd.vpr import winograd as wg
    lib = _lib.load()
    torch.manual_seed(211)
    x = (torch.randn(B, 64, H, W, device="cuda") * amp).contiguous(memory_format=torch.channels_last)
    w = torch.randn(128, 64, 3, 3, device="cuda") / 24.0
    b = torch.randn(128, device="cuda") * amp if bias else None
    Wr = wg.direct_r_pair_weights(w)
    assert tuple(Wr[0].shape) == (4, 9, 2, 2, 2, 64, 8) and Wr[0].dtype == torch.float16
    slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    _lib.check(lib.cslam_absmax_dev(x.data_ptr(), x.numel(), slot.data_ptr(), torch.cuda.current_stream().cuda_stream))
    out_slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    y = wg.conv3x3_direct_r(x, Wr, b, relu, pool, slot, out_slot)
    yh = wg.conv3x3_direct_h(x, wg.direct_pair_weights(w), b, relu, pool, slot, None)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2)             # max and ReLU commute
    if relu:
        ref = torch.relu(ref)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    top = ref.abs().max().item()
    ef = (y.double() - ref).abs().max().item() / top
    rf = float(((y.double() - ref) ** 2).sum().sqrt() / (ref ** 2).sum().sqrt())
    assert ef <= 5e-6 and rf <= 2e-6, (ef, rf)
    assert (y - yh).abs().max().item() <= 5e-6 * top
    assert out_slot.item() == y.abs().max().item()
    assert torch.equal(y, wg.conv3x3_direct_r(x, Wr, b, relu, pool, slot, None))      # run to run bit-identical


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,relu,pool,bias,amp", [(2, 112, 112, True, True, True, 1.0), (3, 37, 50, True, False, True, 1e3),
                                                      (300, 16, 16, False, False, False, 1.0), (1, 8, 90, True, True, False, 1e-3),
                                                      (5, 20, 34, False, True, True, 1.0), (1, 2, 2, True, True, True, 1.0),
                                                      (40, 114, 118, True, True, True, 0.05), (1, 8, 16, True, False, True, 1.0)])
def test_register_resident_conv_128_to_128_on_output_channel_halves_equals_float64(T, B, H, W, relu, pool, bias, amp):
    """cslam_conv3x3_direct_r2_dev (csrc/conv_direct_r.hip: VGG-16 conv2_2, 128 -> 128 channels; half the output channels' weights
    register-resident per workgroup, the two workgroups of a pair walking the same blocks, a block = two 64-channel passes) against a
    float64 conv2d (+ ReLU + MaxPool2d) at the direct kernels' fp32-grade bar and against the kernel with the weights through an LDS
    ring; ragged 8 x 16 blocks, a single block (one workgroup pair), fewer / more blocks than workgroup pairs, activations six decades
    apart, the max |y| slot, run-to-run bit identity."""
    torch, _ = T
    from cslam_amd import _lib
    from cslam_amd.vpr import winograd as wg
    lib = _lib.load()
    torch.manual_seed(223)
    x = (torch.randn(B, 128, H, W, device="cuda") * amp).contiguous(memory_format=torch.channels_last)
    w = torch.randn(128, 128, 3, 3, device="cuda") / 34.0
    w[5] *= 40.0                                                 # output channels of very different weight
    b = torch.randn(128, device="cuda") * amp if bias else None
    Wr2 = wg.direct_r2_pair_weights(w)
    assert tuple(Wr2[0].shape) == (2, 4, 9, 2, 2, 2, 64, 8) and Wr2[0].dtype == torch.float16
    slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    _lib.check(lib.cslam_absmax_dev(x.data_ptr(), x.numel(), slot.data_ptr(), torch.cuda.current_stream().cuda_stream))
    out_slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    y = wg.conv3x3_direct_r2(x, Wr2, b, relu, pool, slot, out_slot)
    yh = wg.conv3x3_direct_h(x, wg.direct_pair_weights(w), b, relu, pool, slot, None)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2)             # max and ReLU commute
    if relu:
        ref = torch.relu(ref)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    top = ref.abs().max().item()
    ef = (y.double() - ref).abs().max().item() / top
    rf = float(((y.double() - ref) ** 2).sum().sqrt() / (ref ** 2).sum().sqrt())
    assert ef <= 5e-6 and rf <= 2e-6, (ef, rf)
    assert (y - yh).abs().max().item() <= 5e-6 * top
    assert out_slot.item() == y.abs().max().item()
    assert torch.equal(y, wg.conv3x3_direct_r2(x, Wr2, b, relu, pool, slot, None))     # run to run bit-identical
    with pytest.raises(_lib.CslamHipError):
        _lib.check(lib.cslam_conv3x3_direct_r2_dev(x.data_ptr(), Wr2[0].data_ptr(), None, B, H, W, 64, 128, 1, 0, slot.data_ptr(), float(Wr2[1]),
                                                   None, y.data_ptr(), torch.cuda.current_stream().cuda_stream))


@pytest.mark.gpu
def test_netvlad_batch_path_descriptors_with_heavy_tailed_weights(T):
    """The same descriptor-level gate with weights that are NOT He-initialised Gaussians (no trained checkpoint ships; this is the nearest
    stand-in): every convolution's output channels rescaled log-normally (sigma 0.7: a factor 16 between the 2.5 % tails), 0.5 % of the
    weights blown up 8 x, biases spread -- activations whose channels differ by orders of magnitude, which is what the power-of-two
    scales of the fp16-pair operands (one scale per layer, from max |x|) have to survive.  Batch path (direct stem, register-resident
    conv2_1, direct conv2_2, pair GEMMs, pair PCA) against the SAME weights in float64, with torch's fp32 convolutions as yardstick."""
    torch, _ = T
    import copy
    import torch.nn.functional as Fn
    from cslam_amd.vpr import heads
    from cslam_amd.vpr.netvlad import NetVLAD
    rng = np.random.default_rng(23)
    imgs = []
    for i in range(48):
        bsz = (6, 10, 16, 24, 40, 64, 96, 160)[i % 8]
        low = rng.random((-(-480 // bsz), -(-640 // bsz), 3)).astype(np.float32)
        img = np.kron(low, np.ones((bsz, bsz, 1), dtype=np.float32))[:480, :640]
        contrast, bright, sigma = rng.uniform(0.05, 1.0), rng.uniform(20.0, 230.0), rng.uniform(0.0, 40.0) * (i % 3 == 0)
        img = bright + contrast * (img - 0.5) * 255.0 * rng.uniform(0.3, 1.0, size=(1, 1, 3)) + rng.normal(0.0, 1.0, size=img.shape) * sigma
        imgs.append(np.clip(img, 0, 255).astype(np.uint8))
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    cfg = {"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096, "frontend.random_seed": 5}
    nv = NetVLAD(cfg, None)
    gen = torch.Generator(device="cpu").manual_seed(77)
    with torch.no_grad():
        for m in nv.encoder.modules():
            if isinstance(m, torch.nn.Conv2d):
                co = m.weight.shape[0]
                scale = torch.exp(0.7 * torch.randn(co, generator=gen))
                scale = scale / scale.pow(2).mean().sqrt()                      # the layer's overall gain stays He's
                spikes = 1.0 + 7.0 * (torch.rand(m.weight.shape, generator=gen) < 0.005).float()
                m.weight.mul_((scale.reshape(-1, 1, 1, 1) * spikes).to(m.weight.device))
                if m.bias is not None:
                    m.bias.copy_((0.3 * torch.randn(co, generator=gen) * m.weight.detach().abs().mean().cpu() * 27.0).to(m.bias.device))
    nv.trunk = None                                            # transformed weights are rebuilt on the next forward
    got = nv.compute_embeddings_device(frames).double()
    assert nv.trunk is not None and nv.trunk.steps[0].Wr is not None and any(st.Wdr is not None for st in nv.trunk.steps), "the register-resident kernels did not run"
    x = heads.preprocess(frames.contiguous(), nv.crop)
    enc = copy.deepcopy(nv.encoder).double()
    W, b, cent = nv.pool.conv_weight.double(), nv.pool.conv_bias, nv.pool.centroids.double()
    comp = nv.pca_components[:4096].double()

    def head(f):
        xf = Fn.normalize(f, p=2.0, dim=1).flatten(2)
        sa = torch.einsum("kc,ncp->nkp", W.reshape(W.shape[0], -1), xf)
        if b is not None:
            sa = sa + b.double().reshape(1, -1, 1)
        a = torch.softmax(sa, dim=1)
        v = torch.einsum("nkp,ncp->nkc", a, xf) - a.sum(2)[:, :, None] * cent[None]
        v = Fn.normalize(Fn.normalize(v, p=2.0, dim=2).flatten(1), p=2.0, dim=1)
        return Fn.normalize(v @ comp.T - nv.pca_mean_proj.double()[None], p=2.0, dim=1)
    ref, ref32, act = [], [], []
    with torch.no_grad():
        for s in range(0, x.shape[0], 16):
            f = enc(x[s:s + 16].double())
            act.append(f.abs().amax(dim=(0, 2, 3)))
            ref.append(head(f))
            ref32.append(head(nv.encoder(x[s:s + 16]).double()))                # torch's own fp32 convolutions (MIOpen), float64 head
    ref, ref32 = torch.cat(ref), torch.cat(ref32)
    chan = torch.stack(act).amax(0)
    worst = (got - ref).abs().max().item()
    e_direct = (ref32 - ref).abs().max().item()
    cos = (got * ref).sum(1)
    print("heavy-tailed weights: channel maxima of the last map span %.1e .. %.1e; descriptor error vs float64 %.2e (torch fp32 convolutions "
          "%.2e), max 1-cos %.1e" % (chan[chan > 0].min().item(), chan.max().item(), worst, e_direct, float((1.0 - cos).max())), flush=True)
    assert torch.isfinite(got).all()
    assert worst <= 1e-5, worst
    assert worst <= 2.0 * e_direct + 1e-6, (worst, e_direct)
    assert float((1.0 - cos).max()) <= 2e-7


@pytest.mark.gpu
@pytest.mark.parametrize("B,cin,cout,hw,k,stride,pad,relu,res", [
    (3, 3, 64, (64, 80), 7, 2, 3, True, False),        # ResNet stem: 7x7 / 2, three input channels (row-packed K blocks)
    (2, 3, 64, (37, 53), 7, 2, 3, False, False),       # ... odd map, ragged last tile
    (4, 64, 128, (28, 28), 3, 2, 1, True, False),      # layer2.0.conv1: 3x3 / 2
    (4, 64, 128, (28, 28), 1, 2, 0, False, False),     # layer2.0.downsample: 1x1 / 2
    (2, 128, 256, (15, 13), 3, 2, 1, True, False),     # odd sides
    (2, 256, 512, (14, 14), 1, 2, 0, False, False),
    (3, 128, 128, (12, 20), 3, 1, 1, True, True),      # stride 1 with the shortcut fused
    (2, 64, 64, (9, 9), 3, 1, 1, True, True),          # 64 output channels: the 64-column tile
    (1, 512, 512, (7, 7), 3, 1, 1, False, True),       # 144 K blocks, 49 pixels: one ragged tile row
])
def test_implicit_gemm_convolution_on_fp16_pairs_equals_float64(T, B, cin, cout, hw, k, stride, pad, relu, res):
    """csrc/conv_igemm.hip (the strided / 1x1 / 7x7 layers of the ResNet trunks, cosplace_utils/network.py:38-68): against the
    same convolution in float64, no further from it than 4 x torch's own float32 convolution (+ 2e-7 of the output's scale), with
    inputs whose magnitudes span several binades; max |y| comes back in the slot; zero padding, ragged tiles, shortcut, ReLU."""
    torch, _ = T
    from cslam_amd.vpr import winograd as wg
    torch.manual_seed(1000 * cin + cout + k)
    H, W = hw
    x = torch.randn((B, cin, H, W), device="cuda") * torch.exp2(torch.randint(-6, 3, (B, cin, 1, 1), device="cuda").float())
    x = x.contiguous(memory_format=torch.channels_last)
    w = torch.randn((cout, cin, k, k), device="cuda") / (k * cin ** 0.5)
    w *= torch.exp2(torch.randint(-3, 2, (cout, 1, 1, 1), device="cuda").float())
    bias = torch.randn(cout, device="cuda") * 0.3
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), stride=stride, padding=pad)
    r = None
    if res:
        r = torch.randn(ref.shape, device="cuda").contiguous(memory_format=torch.channels_last)
        ref = ref + r.double()
    if relu:
        ref = ref.relu()
    a32 = torch.nn.functional.conv2d(x, w, bias, stride=stride, padding=pad)
    if res:
        a32 = a32 + r
    if relu:
        a32 = a32.relu()
    ws = wg._Workspace()
    slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    y = wg.conv_igemm(ws, x, wg.igemm_pair_weights(w), bias, (k, k), stride, pad, relu, r, amax_out=slot)
    torch.cuda.synchronize()
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    scale = ref.abs().max().item()
    e = (y.double() - ref).abs().max().item() / scale
    e32 = (a32.double() - ref).abs().max().item() / scale
    assert e <= 4 * e32 + 2e-7, (e, e32)
    assert slot.item() == y.abs().max().item()
    # a bound instead of the measured maximum (what a caller with a known input range passes): same result to rounding
    bound = torch.full((1,), float(x.abs().max().item()) * 1.7, device="cuda")
    y2 = wg.conv_igemm(ws, x, wg.igemm_pair_weights(w), bias, (k, k), stride, pad, relu, r, amax_in=bound)
    assert (y2.double() - ref).abs().max().item() / scale <= 4 * e32 + 4e-7


@pytest.mark.gpu
@pytest.mark.parametrize("B,c0,c1,hw,k2,stride2", [(3, 64, 64, (20, 24), 3, 1), (2, 64, 128, (19, 13), 3, 2), (5, 128, 256, (14, 14), 1, 2),
                                                   (1, 256, 512, (7, 7), 3, 1), (3, 128, 128, (13, 9), 3, 1), (2, 128, 256, (5, 130), 3, 1),
                                                   (24, 128, 128, (28, 28), 3, 1), (21, 128, 256, (29, 27), 3, 1), (350, 256, 256, (7, 7), 3, 1)])
def test_pair_format_activations_between_implicit_gemm_layers(T, B, c0, c1, hw, k2, stride2):
    """`cslam_conv_igemm_h2p_dev`: float32 map -> conv (pair-format output) -> conv reading the pairs by LDS-DMA, with a pair-format and
    with a float32 shortcut, pair-format and float32 output.  Every result against the same chain in float64, no further from it than 4 x
    torch's float32 chain (+ 4e-7 of the scale): the bound-derived scale costs nothing visible; the measured max |y| and the bound slots
    are right; ragged pixel tiles, zero padding, strides.  The 3x3 / stride-1 second layers with 128-channel tiles run the shared-row form
    (one activation block for the three taps of a kernel row, row-end pixels masked in the fragments): maps of 7 x 7, 13 x 9 (rows
    shorter than a tile, tiles across images) and 5 x 130 (a row longer than a tile), with and without the shortcut.  From 16 384 output
    pixels on those layers take 256-pixel tiles (one workgroup per CU, 258-row blocks): 24 x 28 x 28, 21 x 29 x 27 (ragged last tile, rows
    that straddle tiles) and 350 x 7 x 7 (five images per tile)."""
    torch, _ = T
    from cslam_amd.vpr import winograd as wg
    torch.manual_seed(c0 * 7 + c1 + k2)
    H, W = hw
    x = torch.randn((B, c0, H, W), device="cuda") * torch.exp2(torch.randint(-5, 2, (B, c0, 1, 1), device="cuda").float())
    x = x.contiguous(memory_format=torch.channels_last)
    w1 = torch.randn((c0, c0, 3, 3), device="cuda") / (3 * c0 ** 0.5)
    b1 = torch.randn(c0, device="cuda") * 0.2
    w2 = torch.randn((c1, c0, k2, k2), device="cuda") / (k2 * c0 ** 0.5)
    b2 = torch.randn(c1, device="cuda") * 0.2
    p2 = k2 // 2
    ws = wg._Workspace()
    slots = torch.zeros(12, dtype=torch.float32, device="cuda")
    slots[0] = x.abs().max()
    a0 = wg.PairAct(x, False, x.shape, slots[0:1], slots[0:1])
    wl1 = lambda w: float(w.abs().sum(dim=(1, 2, 3)).max())        # noqa: E731
    # layer 1: float32 in, pairs out (ReLU)
    a1 = wg.conv_igemm_p(ws, a0, wg.igemm_pair_weights(w1), b1, (3, 3), 1, 1, True, None, wl1(w1), float(b1.abs().max()),
                         slots[1:2], slots[2:3], True)
    r1 = torch.nn.functional.conv2d(x.double(), w1.double(), b1.double(), padding=1).relu()
    f1 = torch.nn.functional.conv2d(x, w1, b1, padding=1).relu()
    y1 = wg.pairs_to_float(a1)
    sc1 = r1.abs().max().item()
    e32 = (f1.double() - r1).abs().max().item() / sc1
    assert (y1.double() - r1).abs().max().item() / sc1 <= 4 * e32 + 4e-7
    assert a1.pairs and a1.t.dtype == torch.float16 and a1.t.shape == (B, H, W, c0 // 32, 2, 32)
    assert abs(slots[1].item() - r1.abs().max().item()) <= 1e-5 * sc1 and slots[2].item() >= slots[1].item()
    assert slots[2].item() <= (slots[0].item() * wl1(w1) + float(b1.abs().max())) * 1.002
    # layer 2: pairs in; shortcut (same geometry only) in pair format, then as float32; output float32 and pairs
    same = c1 == c0 and stride2 == 1
    r2 = torch.nn.functional.conv2d(r1, w2.double(), b2.double(), stride=stride2, padding=p2)
    f2 = torch.nn.functional.conv2d(f1, w2, b2, stride=stride2, padding=p2)
    if same:
        r2, f2 = r2 + r1, f2 + f1
    r2, f2 = r2.relu(), f2.relu()
    sc2 = r2.abs().max().item()
    e32 = (f2.double() - r2).abs().max().item() / sc2
    Wg2 = wg.igemm_pair_weights(w2)
    res_forms = [a1, wg.PairAct(y1.contiguous(memory_format=torch.channels_last), False, y1.shape, slots[1:2], slots[1:2])] if same else [None]
    for res in res_forms:
        for out_pairs in (False, True):
            slots[3:].zero_()
            a2 = wg.conv_igemm_p(ws, a1, Wg2, b2, (k2, k2), stride2, p2, True, res, wl1(w2), float(b2.abs().max()), slots[3:4],
                                 slots[4:5], out_pairs)
            y2 = wg.pairs_to_float(a2) if out_pairs else a2.t
            assert tuple(y2.shape) == tuple(r2.shape)
            assert (y2.double() - r2).abs().max().item() / sc2 <= 4 * e32 + 4e-7, (res is not None and res.pairs, out_pairs)
            assert abs(slots[3].item() - sc2) <= 1e-5 * sc2


@pytest.mark.gpu
@pytest.mark.parametrize("B,hw", [(3, (56, 56)), (5, (7, 7)), (3, (13, 9)), (2, (5, 130)), (1, (8, 16)), (2, (112, 112)), (300, (56, 56)),
                                  (1, (1, 1)), (4, (9, 23))])
def test_register_resident_pair_convolution_equals_float64(T, B, hw):
    """`cslam_conv3x3_direct_p_dev` (csrc/conv_direct_p.hip: ResNet layer1's 64 -> 64 convolutions between pair-format maps, weights
    register-resident, patch by LDS-DMA, blocks of two 8 x 8 halves that may straddle block rows and images): a float32 map -> the
    implicit GEMM (pair-format output) -> this kernel, with no shortcut, a pair-format one and a float32 one, pair-format and float32
    output, with and without ReLU.  Every result against the same chain in float64, no further from it than 4 x torch's float32 chain
    (+ 4e-7 of the scale), and next to the implicit GEMM's result on the same operands; the measured max |y| and the bound slots are
    right; repeated launches are bit-identical.  Maps: ResNet's 56 x 56 (seven halves per block row: blocks straddle rows and images;
    300 images: every workgroup walks several blocks), ragged ones, a row longer than a block, a single pixel."""
    torch, _ = T
    from cslam_amd.vpr import winograd as wg
    torch.manual_seed(B * 131 + hw[0] * 7 + hw[1])
    H, W = hw
    x = torch.randn((B, 64, H, W), device="cuda") * torch.exp2(torch.randint(-5, 2, (B, 64, 1, 1), device="cuda").float())
    x = x.contiguous(memory_format=torch.channels_last)
    w1 = torch.randn((64, 64, 3, 3), device="cuda") / 24
    b1 = torch.randn(64, device="cuda") * 0.2
    w2 = torch.randn((64, 64, 3, 3), device="cuda") / 24 * torch.exp2(torch.randint(-3, 1, (64, 1, 1, 1), device="cuda").float())
    b2 = torch.randn(64, device="cuda") * 0.2
    ws = wg._Workspace()
    slots = torch.zeros(12, dtype=torch.float32, device="cuda")
    slots[0] = x.abs().max()
    a0 = wg.PairAct(x, False, x.shape, slots[0:1], slots[0:1])
    wl1 = lambda w: float(w.abs().sum(dim=(1, 2, 3)).max())        # noqa: E731
    a1 = wg.conv_igemm_p(ws, a0, wg.igemm_pair_weights(w1), b1, (3, 3), 1, 1, True, None, wl1(w1), float(b1.abs().max()),
                         slots[1:2], slots[2:3], True)
    r1 = torch.nn.functional.conv2d(x.double(), w1.double(), b1.double(), padding=1).relu()
    f1 = torch.nn.functional.conv2d(x, w1, b1, padding=1).relu()
    y1 = wg.pairs_to_float(a1)
    Wp, Wg2 = wg.stem_direct_pair_weights(w2), wg.igemm_pair_weights(w2)
    assert wg.direct_p_fits(w2.shape, (3, 3), 1, 1, H, W)
    # the float32-input form (the layer that opens the pair-format chain: its patch is split while it is staged through registers)
    sc1 = r1.abs().max().item()
    e32 = (f1.double() - r1).abs().max().item() / sc1
    for out_pairs in (True, False):
        slots[9:].zero_()
        a1d = wg.conv3x3_direct_p(a0, wg.stem_direct_pair_weights(w1), b1, True, None, wl1(w1), float(b1.abs().max()), slots[9:10], slots[10:11], out_pairs)
        y1d = wg.pairs_to_float(a1d) if out_pairs else a1d.t
        assert (y1d.double() - r1).abs().max().item() / sc1 <= 4 * e32 + 4e-7
        assert (y1d - y1).abs().max().item() / sc1 <= 2e-6 and abs(slots[9].item() - sc1) <= 1e-5 * sc1
        if out_pairs:
            assert slots[10].item() == slots[2].item() and torch.equal(a1d.t, wg.conv3x3_direct_p(
                a0, wg.stem_direct_pair_weights(w1), b1, True, None, wl1(w1), float(b1.abs().max()), slots[11:12], slots[10:11], True).t)
    res_forms = [None, a1, wg.PairAct(y1.contiguous(memory_format=torch.channels_last), False, y1.shape, slots[1:2], slots[1:2])]
    for res in res_forms:
        for relu in (True, False):
            r2 = torch.nn.functional.conv2d(r1, w2.double(), b2.double(), padding=1)
            f2 = torch.nn.functional.conv2d(f1, w2, b2, padding=1)
            if res is not None:
                r2, f2 = r2 + r1, f2 + f1
            if relu:
                r2, f2 = r2.relu(), f2.relu()
            sc2 = r2.abs().max().item()
            e32 = (f2.double() - r2).abs().max().item() / sc2
            for out_pairs in (True, False):
                slots[3:].zero_()
                a2 = wg.conv3x3_direct_p(a1, Wp, b2, relu, res, wl1(w2), float(b2.abs().max()), slots[3:4], slots[4:5], out_pairs)
                y2 = wg.pairs_to_float(a2) if out_pairs else a2.t
                assert tuple(y2.shape) == tuple(r2.shape)
                err = (y2.double() - r2).abs().max().item() / sc2
                assert err <= 4 * e32 + 4e-7, (res is not None and res.pairs, relu, out_pairs, err, e32)
                assert abs(slots[3].item() - sc2) <= 1e-5 * sc2
                if out_pairs:
                    rb = 0.0 if res is None else res.bound.item()
                    assert slots[4].item() >= slots[3].item()
                    assert slots[4].item() <= (slots[1].item() * wl1(w2) + float(b2.abs().max()) + rb) * 1.002
                # the implicit GEMM on the same operands: the same arithmetic in another summation order
                slots[5:].zero_()
                g2 = wg.conv_igemm_p(ws, a1, Wg2, b2, (3, 3), 1, 1, relu, res, wl1(w2), float(b2.abs().max()), slots[5:6], slots[6:7], out_pairs)
                yg = wg.pairs_to_float(g2) if out_pairs else g2.t
                assert (y2 - yg).abs().max().item() / sc2 <= 2e-6
                if out_pairs:
                    assert slots[6].item() == slots[4].item()
                # bit-stable
                slots[7:].zero_()
                a3 = wg.conv3x3_direct_p(a1, Wp, b2, relu, res, wl1(w2), float(b2.abs().max()), slots[7:8], slots[8:9], out_pairs)
                assert torch.equal(a3.t, a2.t) and slots[7].item() == slots[3].item()


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,cout,k,stride,pad", [(3, 224, 224, 64, 7, 2, 3), (2, 64, 96, 64, 7, 2, 3), (5, 32, 32, 128, 3, 2, 1),
                                                    (2, 16, 32, 64, 3, 1, 1), (1, 32, 64, 64, 7, 2, 3), (7, 224, 224, 64, 7, 2, 3)])
def test_stem_with_fused_maxpool_equals_pooling_the_unfused_output(T, B, H, W, cout, k, stride, pad):
    """`cslam_conv_stem_pool_igemm_h2_dev` (conv1 + bn1 + relu + maxpool of the ResNet trunks as one kernel, 8 x 16-pixel tiles, windows
    across tiles completed with atomic maxima; the 7 x 7 / 2 / 64-channel cases run the persistent patch-form kernel -- 4 tiles on two
    workgroups, 24, 294 and 686 tiles over the XCDs --, the others the implicit-GEMM form): EXACTLY MaxPool2d(3, 2, 1) of the un-fused
    kernel's output (the same products in the same order), which the test above holds against float64; the slot carries max of the un-pooled map; repeated calls agree (the
    atomics have no order dependence)."""
    torch, _ = T
    from cslam_amd.vpr import winograd as wg
    torch.manual_seed(B * 1000 + H + cout)
    x = (torch.rand((B, 3, H, W), device="cuda") * 4.6 - 2.2).contiguous(memory_format=torch.channels_last)
    w = torch.randn((cout, 3, k, k), device="cuda") / (k * 3 ** 0.5)
    bias = torch.randn(cout, device="cuda") * 0.3
    Wg = wg.igemm_pair_weights(w)
    ws = wg._Workspace()
    s0 = torch.zeros(1, dtype=torch.float32, device="cuda")
    s1 = torch.zeros(1, dtype=torch.float32, device="cuda")
    y = wg.conv_igemm(ws, x, Wg, bias, (k, k), stride, pad, True, amax_out=s0)
    assert wg.stem_pool_fits(y.shape[2], y.shape[3])
    want = torch.nn.functional.max_pool2d(y, 3, 2, 1)
    for _ in range(2):
        s1.zero_()
        got = wg.conv_igemm(ws, x, Wg, bias, (k, k), stride, pad, True, amax_out=s1, pool=True)
        torch.cuda.synchronize()
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(got, want)
        assert s1.item() == s0.item() == y.max().item()
    ref = torch.nn.functional.max_pool2d(torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), stride=stride,
                                                                    padding=pad).relu(), 3, 2, 1)
    assert (got.double() - ref).abs().max().item() / ref.abs().max().item() < 1e-6
