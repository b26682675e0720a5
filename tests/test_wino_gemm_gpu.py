"""GPU parity tests of this library's split-fp16 Winograd GEMM (csrc/wino_gemm.hip; -m gpu): the 36 per-frequency
products of the F(4x4,3x3) form of the trunk's wide convolutions (cslam/vpr/netvlad.py:163-171,227), computed on the
fp16 matrix pipe from exact hi/lo pairs.  Compared with float64 evaluations of the same sums (tolerances stated per
test) and, at layer / trunk level, with the plain-fp32 forms beside it."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    from cslam_amd import _lib
    return torch, _lib


def _p(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr())


def _pairs_rows(v):
    """float32 [36, R, K] (|v| < 2^15) -> fp16 [36, R, K/32, 2, 32]: the operand layout of the GEMM, hi + lo == v to 2^-22."""
    import torch
    hi = v.to(torch.float16)
    lo = (v - hi.float()).to(torch.float16)
    n, r, k = v.shape
    return torch.stack((hi.view(n, r, k // 32, 32), lo.view(n, r, k // 32, 32)), dim=3).contiguous(), hi, lo


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("rows,cin,cout", [(300, 128, 128), (1024, 256, 256), (257, 512, 512), (5000, 128, 256),
                                           (2049, 256, 512), (64, 32, 128), (70000, 64, 256)])
def test_pair_gemm_equals_float64(T, rows, cin, cout, cfg, monkeypatch):
    """M = (vh + vl)(uh + ul) - vl ul, every frequency, ragged row counts, both tile widths (Cout 128 -> 128-column
    tiles, multiples of 256 -> 256-column tiles), one to sixteen K stages.  The three fp16 x fp16 products are exact in
    fp32, so the only error is the fp32 accumulation: <= 2e-6 of sum |terms| (K <= 512 terms x 3), checked per element
    against a float64 evaluation of the same three products; and the dropped vl ul is <= 2^-21 of the full product."""
    torch, _lib = T
    from cslam_amd.vpr import winograd as wg
    lib = _lib.load()
    if cfg:
        monkeypatch.setenv("CSLAM_WGEMM_CFG", str(cfg))
    else:
        monkeypatch.delenv("CSLAM_WGEMM_CFG", raising=False)
    g = torch.Generator(device="cuda").manual_seed(rows + cin)
    v = (torch.randn((36, rows, cin), generator=g, device="cuda") * 3000.0).clamp_(-30000, 30000)
    U4 = torch.randn((36, cin, cout), generator=g, device="cuda") / cin ** 0.5
    V2, vh, vl = _pairs_rows(v)
    U2, inv_su = wg.split16_pair_weights(U4)
    M = torch.full((36, rows, cout), float("nan"), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.cslam_wino_gemm_h2_dev(_p(V2), _p(U2), rows, cin, cout, _p(M), st))
    torch.cuda.synchronize()
    assert torch.isfinite(M).all()
    uh = U2[:, :, :, 0, :].reshape(36, cout, cin).transpose(1, 2).double()
    ul = U2[:, :, :, 1, :].reshape(36, cout, cin).transpose(1, 2).double()
    rec = (uh + ul) * inv_su                                            # the pair layout carries U to 22 bits
    assert (rec - U4.double()).abs().max().item() <= 2.0 ** -21 * U4.abs().max().item()
    vhd, vld = vh.double(), vl.double()
    want = vhd @ uh + vld @ uh + vhd @ ul
    mag = vhd.abs() @ uh.abs() + vld.abs() @ uh.abs() + vhd.abs() @ ul.abs()
    err = (M.double() - want).abs()
    assert (err <= 2e-6 * mag + 1e-30).all(), float((err / mag).max())
    full = (vhd + vld) @ (uh + ul)
    assert ((want - full).abs() <= 2.0 ** -21 * mag).all()


@pytest.mark.parametrize("B,H,W,C", [(3, 56, 56, 128), (2, 13, 15, 256), (1, 4, 4, 32)])
def test_input_transform_pair_layout(T, B, H, W, C):
    """cslam_wino4_input_h2_dev: hi = fp16(sV V) and hi + lo = sV V to 2^-22 (sV V = the fp32 transform of the scaled
    input, i.e. cslam_wino4_input_dev on sV x: the scale is a power of two and commutes with the fp32 arithmetic),
    stored as [36][T][C/32][hi 32 | lo 32]; ragged maps (tiles hanging over the border)."""
    torch, _lib = T
    lib = _lib.load()
    torch.manual_seed(B * H + C)
    x = (torch.randn((B, C, H, W), device="cuda") * 7.0).contiguous(memory_format=torch.channels_last)
    tiles = B * -(-H // 4) * -(-W // 4)
    st = torch.cuda.current_stream().cuda_stream
    slot = torch.zeros(1, dtype=torch.float32, device="cuda")
    _lib.check(lib.cslam_absmax_dev(_p(x), x.numel(), _p(slot), st))
    amax = slot.item()
    assert amax == x.abs().max().item()
    sc = 2.0 ** np.floor(np.log2(327.68 / amax))
    V2 = torch.zeros((36, tiles, C // 32, 2, 32), dtype=torch.float16, device="cuda")
    _lib.check(lib.cslam_wino4_input_h2_dev(_p(x), B, H, W, C, _p(slot), _p(V2), st))
    V = torch.empty((36, tiles, C), device="cuda")
    xs = (x * sc).contiguous(memory_format=torch.channels_last)
    _lib.check(lib.cslam_wino4_input_dev(_p(xs), B, H, W, C, _p(V), st))
    torch.cuda.synchronize()
    hi = V2[:, :, :, 0, :].reshape(36, tiles, C)
    lo = V2[:, :, :, 1, :].reshape(36, tiles, C)
    assert V.abs().max().item() < 32768.0
    assert torch.equal(hi, V.to(torch.float16))
    rec = hi.double() + lo.double()
    tol = 2.0 ** -22 * V.double().abs() + 2.0 ** -25           # fp16 subnormal spacing 2^-24 at the bottom of lo's range
    assert ((rec - V.double()).abs() <= tol).all()



@pytest.mark.parametrize("rows,cin,cout", [(300, 128, 128), (1000, 256, 256), (129, 32, 128), (5000, 128, 256), (70000, 64, 128)])
def test_z_form_gemm_equals_the_column_transform_of_the_plain_products(T, rows, cin, cout):
    """cslam_wino_zgemm_h2_dev: Z[4 i + q] = sum_j M[6 i + j] A^T[q][j] with M from cslam_wino_gemm_h2_dev on the same operands --
    the same fp32 products, folded with four exact small-integer coefficients per plane: equal to a float64 fold of M within
    fp32 rounding of the fold (<= 4e-7 of sum |terms|).  Ragged row counts (partial 128-row blocks), one and two column blocks."""
    torch, _lib = T
    from cslam_amd.vpr import winograd as wg
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(rows + cin)
    v = (torch.randn((36, rows, cin), generator=g, device="cuda") * 3000.0).clamp_(-30000, 30000)
    U4 = torch.randn((36, cin, cout), generator=g, device="cuda") / cin ** 0.5
    V2, _, _ = _pairs_rows(v)
    U2, _ = wg.split16_pair_weights(U4)
    M = torch.empty((36, rows, cout), device="cuda")
    Z = torch.full((24, rows, cout), float("nan"), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.cslam_wino_gemm_h2_dev(_p(V2), _p(U2), rows, cin, cout, _p(M), st))
    _lib.check(lib.cslam_wino_zgemm_h2_dev(_p(V2), _p(U2), rows, cin, cout, _p(Z), st))
    torch.cuda.synchronize()
    assert torch.isfinite(Z).all()
    At = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64, device="cuda")
    Md = M.double().view(6, 6, rows, cout)
    want = torch.einsum("qj,ijrc->iqrc", At, Md).reshape(24, rows, cout)
    mag = torch.einsum("qj,ijrc->iqrc", At.abs(), Md.abs()).reshape(24, rows, cout)
    err = (Z.double() - want).abs()
    assert (err <= 4e-7 * mag + 1e-30).all(), float((err / (mag + 1e-30)).max())


@pytest.mark.parametrize("B,H,W,cin,cout,pool", [(3, 28, 28, 128, 128, True), (2, 13, 15, 256, 256, False), (5, 56, 56, 128, 256, False)])
def test_z_form_layer_equals_plain_form_layer_and_float64(T, B, H, W, cin, cout, pool, monkeypatch):
    """A whole layer (input transform, products, output transform + bias + ReLU (+ MaxPool2d)) through the Z form against the
    36-plane form and a float64 convolution: the two forms agree to fp32 rounding, both inside the layer tolerance of
    tests/test_heads_gpu.py (2e-5 of the largest activation); max |y| delivered for the next layer is identical in kind."""
    torch, _lib = T
    from cslam_amd.vpr import winograd as wg
    torch.manual_seed(B * H + cin)
    x = torch.relu(torch.randn((B, cin, H, W), device="cuda")).contiguous(memory_format=torch.channels_last)
    w = torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5)
    bias = torch.randn(cout, device="cuda") * 0.1
    U, U4 = wg.wino_weights(w, 2).cuda(), wg.wino_weights(w, 4).cuda()
    U2 = wg.split16_pair_weights(U4)
    ws = wg._Workspace()
    outs, slots = {}, {}
    for tag, z in (("z", 262144), ("plain", 0)):
        monkeypatch.setattr(wg, "Z_FORM_MAX", z)
        slot = torch.zeros(1, dtype=torch.float32, device="cuda")
        big = x.repeat(-(-512 // (B * -(-H // 4) * -(-W // 4))), 1, 1, 1) if B * -(-H // 4) * -(-W // 4) < 512 else x
        y = wg.wino_conv3x3(ws, big.contiguous(memory_format=torch.channels_last), U, U4, bias, True, pool=pool, U2=U2, amax_out=slot)
        torch.cuda.synchronize()
        outs[tag], slots[tag] = y[:B].clone(), slot.item()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=1).relu()
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2)
    scale = ref.abs().max().item()
    for tag in ("z", "plain"):
        assert (outs[tag].double() - ref).abs().max().item() <= 2e-5 * scale, tag
    assert (outs["z"] - outs["plain"]).abs().max().item() <= 2e-6 * scale
    assert abs(slots["z"] - slots["plain"]) <= 2e-6 * scale and slots["z"] > 0
