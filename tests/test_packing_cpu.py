"""CPU suite: the host-side operand packers of the HIP kernels (pure tensor arithmetic, no GPU): the exact fp16 hi / lo pair
split and the memory orders the kernels read -- checked by re-assembling the original float32 values from the packed form."""
import numpy as np
import torch

from cslam_amd.vpr import heads
from cslam_amd.vpr import winograd as wg


def _halves(dwords):
    """int32 tensor of packed dwords -> (low half, high half) as float16 tensors"""
    u = dwords.to(torch.int64) & 0xFFFFFFFF
    lo = torch.from_numpy((u.numpy() & 0xFFFF).astype(np.uint16).view(np.float16))
    hi = torch.from_numpy(((u.numpy() >> 16) & 0xFFFF).astype(np.uint16).view(np.float16))
    return lo, hi


def test_pca_pair_weights_reassemble_to_22_bits_in_the_gemm_row_order():
    torch.manual_seed(0)
    w = torch.randn(256, 2048) / 45.0
    pairs, inv_sw = heads.pca_pair_weights(w, splits=4)
    assert pairs.shape == (4, 256, 16, 2, 32) and pairs.dtype == torch.float16
    rec = (pairs[:, :, :, 0, :].double() + pairs[:, :, :, 1, :].double()) * inv_sw          # [S, Dout, kb, 32]
    rec = rec.permute(1, 0, 2, 3).reshape(256, 2048)
    assert float((rec - w.double()).abs().max()) <= 2.0 ** -21 * float(w.abs().max())
    assert float(pairs.abs().max()) < 2.0 ** 15 and float(pairs[:, :, :, 0, :].abs().max()) >= 2.0 ** 13
    assert heads.pca_pair_weights(torch.randn(100, 2048)) is None and heads.pca_pair_weights(torch.randn(128, 1000)) is None


def test_stem_pair_weights_slots_taps_and_bound():
    torch.manual_seed(1)
    w = torch.randn(64, 3, 3, 3) / 5.0
    W1, inv_sw, sumw = wg.stem_pair_weights(w)
    assert W1.shape == (4, 2, 64, 4) and W1.dtype == torch.int32 and sumw.shape == (64,)
    sw = 1.0 / inv_sw
    for hl in (0, 1):
        lo16, hi16 = _halves(W1[:, hl].reshape(-1))
        vals = torch.stack((lo16, hi16), dim=1).reshape(4, 64, 4, 2).reshape(4, 64, 8)       # [kq][16 g + n][slot j]
        if hl == 0:
            hi_part = vals.double()
        else:
            lo_part = vals.double()
    rec = (hi_part + lo_part) * inv_sw
    for kq in range(4):
        for g in range(4):
            for n in range(16):
                co = 16 * kq + n
                for j in range(8):
                    if g < 3:
                        want = float(w[co, j % 3, g, j // 3])          # tap (ky = g, kx = j // 3, ci = j % 3)
                    elif j < 3:
                        want = float(w[co, 2, j, 2])                   # the ninth tap (kx = 2, ci = 2) of row ky = j
                    else:
                        want = 0.0
                    assert abs(float(rec[kq, 16 * g + n, j]) - want) <= 2.0 ** -21 * float(w.abs().max())
    assert torch.all(sumw.double() >= w.double().abs().sum(dim=(1, 2, 3)))                    # rounded UP: a rigorous bound
    assert float(sw) == 2.0 ** round(np.log2(sw))


def test_register_resident_weight_fragments_are_in_mfma_lane_order():
    """`stem_direct_pair_weights` / `direct_r_pair_weights` (csrc/conv_stem_direct_h.hip, conv_direct_r.hip): every (wave, tap, K step
    [, channel tile], half) is one v_mfma_f32_16x16x32_f16 A fragment -- lane l holds output channel l % 16 of the tile, input channels
    8 (l // 16) .. + 7 of the K step --, hi + lo reproduce the power-of-two-scaled weight to 22 bits."""
    torch.manual_seed(5)
    w = torch.randn(64, 64, 3, 3) / 24.0
    W, inv = wg.stem_direct_pair_weights(w)
    assert W.shape == (4, 9, 2, 2, 64, 8) and W.dtype == torch.float16 and float(1.0 / inv) == 2.0 ** round(np.log2(1.0 / inv))
    rec = (W[:, :, :, 0].double() + W[:, :, :, 1].double()) * inv                              # [q][tap][ks][lane][e]
    want = w.double().reshape(4, 16, 2, 4, 8, 9).permute(0, 5, 2, 3, 1, 4).reshape(4, 9, 2, 64, 8)   # [q][i][ks][kg][e][tap] -> lane = 16 kg + i
    assert float((rec - want).abs().max()) <= 2.0 ** -21 * float(w.abs().max())
    assert float(W.abs().max()) < 2.0 ** 15 and float(W[:, :, :, 0].abs().max()) >= 2.0 ** 13
    w2 = torch.randn(128, 64, 3, 3) / 24.0
    W2, inv2 = wg.direct_r_pair_weights(w2)
    assert W2.shape == (4, 9, 2, 2, 2, 64, 8) and W2.dtype == torch.float16
    rec2 = (W2[:, :, :, :, 0].double() + W2[:, :, :, :, 1].double()) * inv2                    # [q][tap][ks][mt][lane][e]
    want2 = w2.double().reshape(4, 2, 16, 2, 4, 8, 9).permute(0, 6, 3, 1, 4, 2, 5).reshape(4, 9, 2, 2, 64, 8)
    assert float((rec2 - want2).abs().max()) <= 2.0 ** -21 * float(w2.abs().max())


def test_split16_pair_weights_layout():
    torch.manual_seed(2)
    U4 = torch.randn(36, 64, 128) / 8.0
    U2, inv_su = wg.split16_pair_weights(U4)
    assert U2.shape == (36, 128, 2, 2, 32)
    rec = (U2[:, :, :, 0, :].double() + U2[:, :, :, 1, :].double()) * inv_su                # [36, Cout, kb, 32]
    rec = rec.reshape(36, 128, 64).transpose(1, 2)
    assert float((rec - U4.double()).abs().max()) <= 2.0 ** -21 * float(U4.abs().max())


def test_normalised_image_bound_covers_every_8_bit_value():
    b = heads.normalised_image_bound()
    v = np.arange(256, dtype=np.float32) / np.float32(255.0)
    worst = max(float(np.abs((v - np.float32(m)) / np.float32(s)).max()) for m, s in zip(heads.IMAGENET_DEFAULT_MEAN,
                                                                                         heads.IMAGENET_DEFAULT_STD))
    assert worst <= b <= worst * 1.001
