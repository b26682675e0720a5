"""GPU tests at BASELINE.json's configuration sizes (-m gpu): properties that do not need the
CPU oracle at full size, plus oracle spot checks.
  C2: CosPlace ResNet-18 512-D extract + causal intra NNS over the growing bank.
  C4: 8 robots x 50k x 4096 banks: best-1 of every robot's new keyframes against every other
      robot's bank (what 8 ranks do after the all-gather; here bank-major on one GPU)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c2_cosplace_extract_and_causal_matching():
    import torch
    from cslam_amd import nns_matching as nnm
    from cslam_amd.vpr.cosplace import CosPlace
    from oracle import pyoracle
    n = 10_000                                  # config 2 at its size: 10k synthetic 640x480 frames (~1 s of extraction)
    cp = CosPlace({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376,
                   "frontend.cosplace.descriptor_dim": 512, "frontend.cosplace.backbone": "resnet18"}, None)
    gen = torch.Generator(device="cuda").manual_seed(7)
    descs = []
    for s in range(0, n, 250):
        frames = torch.randint(0, 256, (250, 480, 640, 3), generator=gen, device="cuda", dtype=torch.uint8)
        descs.append(cp.compute_embeddings_device(frames))
    d = torch.cat(descs)
    assert d.shape == (n, 512) and torch.allclose(d.norm(dim=1), torch.ones(n, device="cuda"), atol=1e-5)
    nn = nnm.NearestNeighborsMatching()
    nn.add_items_device(d)
    lim = torch.arange(n, device="cuda", dtype=torch.int64)
    rows, sims, cnt = nn.search_device(d, 5, row_limit=lim, mode=nnm.MODE_MFMA)
    assert torch.equal(cnt.cpu(), torch.clamp(torch.arange(n), max=5).int())
    valid = rows >= 0
    assert torch.all(rows[valid] < lim[:, None].expand(-1, 5)[valid])       # causal: only earlier keyframes
    assert torch.all(sims[:, :-1][valid[:, 1:]] >= sims[:, 1:][valid[:, 1:]])
    hd = d.cpu().numpy()
    sel = np.arange(0, n, 197)                  # 51 keyframes spread over the run, the last ones against ~10k rows
    oi, os_, oc = pyoracle.nns_search(hd, hd[sel], 5, row_limit=sel.astype(np.int64))
    assert np.array_equal(rows.cpu().numpy()[sel], oi) and np.array_equal(cnt.cpu().numpy()[sel], oc)
    assert np.nanmax(np.abs(sims.cpu().numpy()[sel] - os_)) < 1e-12


def test_c4_eight_robot_banks_best1():
    import torch
    from cslam_amd import nns_matching as nnm
    R, N, D, Q = 8, 50_000, 4096, 512
    gen = torch.Generator(device="cuda").manual_seed(1)
    banks, nns = [], []
    for r in range(R):
        b = torch.randn((N, D), generator=gen, device="cuda")
        b /= b.norm(dim=1, keepdim=True)
        nn = nnm.NearestNeighborsMatching()
        nn.add_items_device(b)
        banks.append(b); nns.append(nn)
    # robot r's new keyframes = noisy copies of rows of robot (r+1)%R's bank: the best match in
    # that bank is known by construction, in every other bank it must equal the scan's answer
    for r in range(R):
        o = (r + 1) % R
        idx = torch.randint(0, N, (Q,), generator=gen, device="cuda")
        q = banks[o][idx] + 0.002 * torch.randn((Q, D), generator=gen, device="cuda")
        rows, sims, cnt = nns[o].search_device(q.contiguous(), 1, mode=nnm.MODE_MFMA)
        assert torch.equal(rows[:, 0], idx) and float(sims.min()) > 0.98
        other = (r + 3) % R
        r1, s1, _ = nns[other].search_device(q[:16].contiguous(), 1, mode=nnm.MODE_MFMA)
        r2, s2, _ = nns[other].search_device(q[:16].contiguous(), 1, mode=nnm.MODE_SCAN)
        assert torch.equal(r1, r2) and float((s1 - s2).abs().max()) < 1e-12
