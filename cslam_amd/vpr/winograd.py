"""Winograd F(2x2, 3x3) execution of the wide 3x3 convolutions of the extractor backbone.

The backbone stays a PyTorch module (same layer list, same parameter names, checkpoints load unchanged:
vpr/backbones.py); `WinogradTrunk` only changes HOW its 3x3 / stride 1 / pad 1 convolutions with many
input channels are executed: hand-written HIP input / output transforms (csrc/winograd.hip through
`cslam_wino_input_dev` / `cslam_wino_output_dev`, bias + ReLU + the following MaxPool fused into the
output transform) around 16 plain fp32 GEMMs (`torch.bmm` = rocBLAS).  2.25x fewer multiplications than
the direct convolution MIOpen runs for the same layer (4x with the F(4x4, 3x3) tiles used where the map
sides are multiples of 4); fp32 throughout, as close to a float64 evaluation as the direct fp32 form is
(1.3e-6 / 3.7e-6 of the largest activation for F(2x2) / F(4x4) against 1.4e-6, tests/test_heads_gpu.py).
The first convolution (3 input channels) stays on torch's direct form, with bias + ReLU (+ MaxPool) applied
in one HIP pass (`cslam_bias_act_pool_dev`).
"""
import ctypes as C
import math
import os

import torch
from torch import nn

from .. import _lib
from .heads import _p, _stream

_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
_G4 = torch.tensor([[1 / 4, 0.0, 0.0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                    [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0.0, 0.0, 1.0]], dtype=torch.float64)


def wino_weights(weight, tile=2):
    """[Cout, Cin, 3, 3] -> U [n*n, Cin, Cout] float32, U[n*i+j] = (G g G^T)[i][j] (computed in float64);
    n = 4 for F(2x2,3x3) (tile=2), 6 for F(4x4,3x3) (tile=4)."""
    G = _G if tile == 2 else _G4
    g = weight.detach().to(torch.float64).cpu()
    u = torch.einsum("ik,ockl,jl->ijco", G, g, G)              # [n,n,Cin,Cout]
    return u.reshape(G.shape[0] ** 2, g.shape[1], g.shape[0]).to(torch.float32).contiguous()


def split16_weights(U4):
    """U4 [36, Cin, Cout] float32 -> (U3 [36, 3 Cin, Cout] float16 = [uh ; uh ; ul], inv_su): the weight operand of the
    split-fp16 GEMM (csrc/winograd.hip, `wino4_input_h3_kernel`): sU U = uh + ul exactly to 22 bits, sU the power of two
    that brings max |U| into [2^14, 2^15); inv_su = 1 / sU."""
    u = U4.detach().to(torch.float64)
    amax = float(u.abs().max())
    su = 2.0 ** (14 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    us = (u * su).to(torch.float32)                                  # exact: a power-of-two scale of float32 values
    uh = us.to(torch.float16)
    ul = (us - uh.to(torch.float32)).to(torch.float16)
    return torch.cat((uh, uh, ul), dim=1).contiguous(), 1.0 / su


def split16_pair_weights(U4):
    """U4 [36, Cin, Cout] float32 -> (U2 [36, Cout, Cin/32, 2, 32] float16, inv_su): the weight operand of this library's
    split-fp16 GEMM (csrc/wino_gemm.hip): rows are OUTPUT channels, every 32-channel block of a row holds its hi halves
    then its lo halves; sU U = uh + ul exactly to 22 bits, sU the power of two that brings max |U| into [2^14, 2^15)."""
    n, cin, cout = U4.shape
    assert cin % 32 == 0
    u = U4.detach().to(torch.float64)
    amax = float(u.abs().max())
    su = 2.0 ** (14 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    us = (u * su).to(torch.float32)                                  # exact: a power-of-two scale of float32 values
    uh = us.to(torch.float16)
    ul = (us - uh.to(torch.float32)).to(torch.float16)
    pair = torch.stack((uh, ul), dim=0)                              # [2, 36, Cin, Cout]
    pair = pair.view(2, n, cin // 32, 32, cout).permute(1, 4, 2, 0, 3)   # [36, Cout, Cin/32, 2, 32]
    return pair.contiguous(), 1.0 / su


def direct_pair_weights(weight):
    """conv weight [Cout, Cin, 3, 3] float32 -> (W2 [9, Cout, Cin/32, 2, 32] float16, inv_sw): the weight operand of the direct
    one-kernel convolution (csrc/conv_direct_h.hip): tap ky * 3 + kx major, rows = output channels, every 32-channel block of a row
    its hi halves then its lo halves (`split16_pair_weights` with the 9 taps in the place of the 36 Winograd frequencies)."""
    cout, cin = weight.shape[:2]
    return split16_pair_weights(weight.detach().to(torch.float32).permute(2, 3, 1, 0).reshape(9, cin, cout))


def igemm_pair_weights(weight):
    """conv weight [Cout, Cin, KH, KW] float32 -> (W2 [Cout, nk, 2, 32] float16, inv_sw): the weight operand of the implicit-GEMM
    convolution on fp16 pairs (csrc/conv_igemm.hip).  Rows = output channels; every K block of 32 holds its hi halves then its lo
    halves; sw w = wh + wl exactly to 22 bits, sw the power of two that brings max |w| into [2^14, 2^15).  K blocks: Cin a multiple
    of 32: (kh, kw, Cin / 32) order; Cin = 3 (the 7x7 stem): one block per kernel row kh, slot kw * 3 + c, the other slots zero."""
    cout, cin, kh, kw = weight.shape
    w = weight.detach().to(torch.float64).permute(0, 2, 3, 1)          # [Cout, KH, KW, Cin]
    if cin == 3:
        assert 3 * kw <= 32
        k = torch.zeros((cout, kh, 32), dtype=torch.float64, device=weight.device)
        k[:, :, :3 * kw] = w.reshape(cout, kh, 3 * kw)
        k = k.reshape(cout, kh * 32)
    else:
        assert cin % 32 == 0
        k = w.reshape(cout, kh * kw * cin)
    amax = float(k.abs().max())
    sw = 2.0 ** (14 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    ks = (k * sw).to(torch.float32)
    wh = ks.to(torch.float16)
    wl = (ks - wh.to(torch.float32)).to(torch.float16)
    nk = k.shape[1] // 32
    pair = torch.stack((wh.view(cout, nk, 32), wl.view(cout, nk, 32)), dim=2)     # [Cout, nk, 2, 32]
    return pair.contiguous(), 1.0 / sw


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def stem_pool_fits(Ho, Wo):
    """Whether the stem's output map splits into the 8 x 16-pixel tiles of the fused MaxPool2d(3, 2, 1) (csrc/conv_igemm.hip)."""
    return Ho % 8 == 0 and Wo % 16 == 0


def conv_igemm(ws, x, Wg, bias, kernel, stride, pad, relu, residual=None, amax_in=None, amax_out=None, pool=False):
    """y = act(conv(x) + bias (+ residual)) through `cslam_conv_igemm_h2_dev` (csrc/conv_igemm.hip): x [B,Cin,H,W] channels_last
    float32, Wg = `igemm_pair_weights(weight)`, kernel = (KH, KW).  amax_in: 4-byte device slot with (a bound of) max |x| (None:
    one pass over x measures it); amax_out: zeroed slot that receives max |y|.  pool (3-channel stem with ReLU, output map of
    8 x 16-pixel tiles: `stem_pool_fits`): MaxPool2d(3, 2, 1) fused, y is the pooled map (`cslam_conv_stem_pool_igemm_h2_dev`)."""
    lib = _lib.load()
    x = x.contiguous(memory_format=torch.channels_last)
    B, Cin, H, W = x.shape
    W2, inv_sw = Wg
    Cout = W2.shape[0]
    KH, KW = kernel
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    s = _stream(x)
    slot = amax_in
    if slot is None:
        slot = ws._buf("amax", 1, x.device)
        if x.numel() % 4 == 0:
            _lib.check(lib.cslam_absmax_dev(_p(x), x.numel(), _p(slot), s))
        else:                                                         # the kernel reads 16 bytes per lane: odd sizes through torch
            slot.copy_(x.abs().max().reshape(1))
    if pool:
        assert Cin == 3 and relu and residual is None and stem_pool_fits(Ho, Wo)
        y = torch.empty((B, Cout, Ho // 2, Wo // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        _lib.check(lib.cslam_conv_stem_pool_igemm_h2_dev(_p(x), _p(W2), _p(bias) if bias is not None else None, B, H, W, Cout, KH, KW,
                                                         stride, pad, _p(slot), float(inv_sw),
                                                         _p(amax_out) if amax_out is not None else None, _p(y), s))
        return y
    y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if residual is not None:
        residual = residual.contiguous(memory_format=torch.channels_last)
        assert residual.shape == y.shape
    _lib.check(lib.cslam_conv_igemm_h2_dev(_p(x), _p(W2), _p(bias) if bias is not None else None,
                                           _p(residual) if residual is not None else None, B, H, W, Cin, Cout, KH, KW, stride, pad,
                                           int(relu), _p(slot), float(inv_sw), _p(amax_out) if amax_out is not None else None, _p(y), s))
    return y


class PairAct(object):
    """An activation between the implicit-GEMM layers of a ResNet trunk.  pairs = False: t is the [B,C,H,W] channels_last float32 map;
    pairs = True: t is the PAIR-FORMAT tensor [B,H,W,C/32,2,32] float16 (csrc/conv_igemm.hip: hi and lo halves of s x, s the power of
    two that brings `bound` into [2^13, 2^14)).  amax: 4-byte device slot with the measured max |x| (or a bound of it); bound: the slot
    the pairs were scaled by (float32 maps: the same slot as amax)."""
    __slots__ = ("t", "pairs", "shape", "amax", "bound")

    def __init__(self, t, pairs, shape, amax, bound):
        self.t, self.pairs, self.shape, self.amax, self.bound = t, pairs, tuple(shape), amax, bound


def pairs_to_float(act):
    """PairAct (pair format) -> [B,C,H,W] channels_last float32 (torch; tests and debugging: the trunk never converts)."""
    B, C, H, W = act.shape
    e = torch.frexp(act.bound.view(torch.float32).clamp(1e-30, 1e30))[1].item()
    s = 2.0 ** (14 - e)
    v = (act.t[:, :, :, :, 0, :].float() + act.t[:, :, :, :, 1, :].float()) / s           # [B,H,W,C/32,32]
    return v.reshape(B, H, W, C).permute(0, 3, 1, 2)


def conv_igemm_p(ws, act, Wg, bias, kernel, stride, pad, relu, res, wl1, bmax, amax_out, bound_out, out_pairs):
    """`cslam_conv_igemm_h2p_dev`: the implicit-GEMM convolution between PairActs.  act / res (or None) in either format; the result is
    a PairAct in pair format (out_pairs) or float32, with amax_out (zeroed slot: measured max |y|) and bound_out as its slots."""
    lib = _lib.load()
    B, Cin, H, W = act.shape
    W2, inv_sw = Wg
    Cout = W2.shape[0]
    KH, KW = kernel
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    dev = act.t.device
    if out_pairs:
        y = torch.empty((B, Ho, Wo, Cout // 32, 2, 32), dtype=torch.float16, device=dev)
    else:
        y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    if res is not None:
        assert res.shape == (B, Cout, Ho, Wo) and (res.pairs or res.t.is_contiguous(memory_format=torch.channels_last))
    assert act.pairs or act.t.is_contiguous(memory_format=torch.channels_last)
    _lib.check(lib.cslam_conv_igemm_h2p_dev(
        _p(act.t), int(act.pairs), _p(act.bound), _p(W2), _p(bias) if bias is not None else None,
        _p(res.t) if res is not None else None, int(res.pairs) if res is not None else 0,
        _p(res.bound) if res is not None else None, B, H, W, Cin, Cout, KH, KW, stride, pad, int(relu), _p(act.amax),
        float(inv_sw), float(wl1), float(bmax), _p(amax_out), int(out_pairs), _p(bound_out) if out_pairs else None, _p(y), _stream(act.t)))
    return PairAct(y, bool(out_pairs), (B, Cout, Ho, Wo), amax_out, bound_out if out_pairs else amax_out)


def conv3x3_direct_p(act, Wp, bias, relu, res, wl1, bmax, amax_out, bound_out, out_pairs):
    """`cslam_conv3x3_direct_p_dev` (csrc/conv_direct_p.hip): the 3x3 / stride 1 / pad 1 convolution 64 -> 64 between PairActs with the
    weights register-resident and the patch by LDS-DMA.  act: pair format, or (no shortcut) a float32 map split while it is staged; res (or None): either format; Wp = `stem_direct_pair_weights
    (weight)`; the other arguments and the result as `conv_igemm_p`."""
    lib = _lib.load()
    B, Cin, H, W = act.shape
    assert Cin == 64 and (act.pairs or (res is None and act.t.is_contiguous(memory_format=torch.channels_last)))
    dev = act.t.device
    if out_pairs:
        y = torch.empty((B, H, W, 2, 2, 32), dtype=torch.float16, device=dev)
    else:
        y = torch.empty((B, 64, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    if res is not None:
        assert res.shape == (B, 64, H, W) and (res.pairs or res.t.is_contiguous(memory_format=torch.channels_last))
    _lib.check(lib.cslam_conv3x3_direct_p_dev(
        _p(act.t), int(act.pairs), _p(act.bound), _p(Wp[0]), _p(bias) if bias is not None else None,
        _p(res.t) if res is not None else None, int(res.pairs) if res is not None else 0,
        _p(res.bound) if res is not None else None, B, H, W, 64, 64, int(relu), _p(act.amax), float(Wp[1]), float(wl1), float(bmax),
        _p(amax_out), int(out_pairs), _p(bound_out) if out_pairs else None, _p(y), _stream(act.t)))
    return PairAct(y, bool(out_pairs), (B, 64, H, W), amax_out, bound_out if out_pairs else amax_out)


def direct_p_fits(conv_shape, kernel, stride, pad, H, W):
    """Whether a convolution takes the register-resident pair-format kernel: 64 -> 64 channels, 3x3 / stride 1 / pad 1, two images' maps
    within 32-bit buffer offsets."""
    return (tuple(conv_shape[:2]) == (64, 64) and tuple(kernel) == (3, 3) and stride == 1 and pad == 1
            and H * W * 512 + 11 * W * 256 < 2 ** 31 - 16)


def conv3x3_direct_h(x, Wd, bias, relu, pool, amax_in, amax_out=None):
    """y = [pool](relu(conv3x3(x) + bias)) through `cslam_conv3x3_direct_h_dev` (csrc/conv_direct_h.hip): x [B,Cin,H,W]
    channels_last float32 (Cin a multiple of 32), 128 output channels; Wd = `direct_pair_weights(weight)`; amax_in = 4-byte device
    slot with (a bound of) max |x|; amax_out (optional): zeroed slot that receives max |y|."""
    lib = _lib.load()
    x = x.contiguous(memory_format=torch.channels_last)
    B, Cin, H, W = x.shape
    W2, inv_sw = Wd
    Cout = W2.shape[1]
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    _lib.check(lib.cslam_conv3x3_direct_h_dev(_p(x), _p(W2), _p(bias) if bias is not None else None, B, H, W, Cin, Cout,
                                              int(relu), int(pool), _p(amax_in), float(inv_sw),
                                              _p(amax_out) if amax_out is not None else None, _p(y), _stream(x)))
    return y


def direct_r_pair_weights(weight):
    """conv weight [128, 64, 3, 3] float32 -> (W2r float16 [4, 9, 2, 2, 2, 64, 8], inv_sw): the register-resident operand of
    `cslam_conv3x3_direct_r_dev` (csrc/conv_direct_r.hip).  sW w split into exact fp16 pairs;
    W2r[q][tap][ks][mt][hi | lo][lane][e] = the pair half of w[32 q + 16 mt + lane % 16][32 ks + 8 (lane // 16) + e][tap // 3][tap % 3]:
    one v_mfma_f32_16x16x32_f16 A fragment per (q, tap, ks, mt, half), wave q of a workgroup holding [q] for the whole kernel."""
    assert tuple(weight.shape) == (128, 64, 3, 3)
    w = weight.detach().to(torch.float64)
    amax = float(w.abs().max())
    sw = 2.0 ** (14 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    ws = (w * sw).to(torch.float32)
    wh = ws.to(torch.float16)
    wl = (ws - wh.to(torch.float32)).to(torch.float16)
    pair = torch.stack((wh, wl), dim=0).reshape(2, 4, 2, 16, 2, 4, 8, 9)    # [hl][q][mt][i][ks][kg][e][tap]
    W2r = pair.permute(1, 7, 4, 2, 0, 5, 3, 6).reshape(4, 9, 2, 2, 2, 64, 8)  # [q][tap][ks][mt][hl][lane = 16 kg + i][e]
    return W2r.contiguous(), 1.0 / sw


def conv3x3_direct_r(x, Wr, bias, relu, pool, amax_in, amax_out=None):
    """y = [pool](relu(conv3x3(x) + bias)) through `cslam_conv3x3_direct_r_dev` (csrc/conv_direct_r.hip): x [B,64,H,W] channels_last
    float32, 128 output channels; Wr = `direct_r_pair_weights(weight)`; amax_in / amax_out as `conv3x3_direct_h`."""
    lib = _lib.load()
    x = x.contiguous(memory_format=torch.channels_last)
    B, Cin, H, W = x.shape
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = torch.empty((B, 128, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    _lib.check(lib.cslam_conv3x3_direct_r_dev(_p(x), _p(Wr[0]), _p(bias) if bias is not None else None, B, H, W, Cin, 128,
                                              int(relu), int(pool), _p(amax_in), float(Wr[1]),
                                              _p(amax_out) if amax_out is not None else None, _p(y), _stream(x)))
    return y


def direct_r2_pair_weights(weight):
    """conv weight [128, 128, 3, 3] float32 -> (W2r2 float16 [2, 4, 9, 2, 2, 2, 64, 8], inv_sw): the register-resident operand of
    `cslam_conv3x3_direct_r2_dev` (csrc/conv_direct_r.hip).  sW w split into exact fp16 pairs;
    W2r2[half][q][tap][ks][slab][hi | lo][lane][e] = the pair half of
    w[64 half + 16 q + lane % 16][64 slab + 32 ks + 8 (lane // 16) + e][tap // 3][tap % 3]: one v_mfma_f32_16x16x32_f16 A fragment per
    (tap, ks, slab, pair half), wave q of the workgroup that owns output-channel half `half` holding [half][q] for the whole kernel."""
    assert tuple(weight.shape) == (128, 128, 3, 3)
    w = weight.detach().to(torch.float64)
    amax = float(w.abs().max())
    sw = 2.0 ** (14 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    ws = (w * sw).to(torch.float32)
    wh = ws.to(torch.float16)
    wl = (ws - wh.to(torch.float32)).to(torch.float16)
    pair = torch.stack((wh, wl), dim=0).reshape(2, 2, 4, 16, 2, 2, 4, 8, 9)      # [hl][half][q][i][slab][ks][kg][e][tap]
    W2 = pair.permute(1, 2, 8, 5, 4, 0, 6, 3, 7).reshape(2, 4, 9, 2, 2, 2, 64, 8)  # [half][q][tap][ks][slab][hl][lane = 16 kg + i][e]
    return W2.contiguous(), 1.0 / sw


def conv3x3_direct_r2(x, Wr2, bias, relu, pool, amax_in, amax_out=None):
    """y = [pool](relu(conv3x3(x) + bias)) through `cslam_conv3x3_direct_r2_dev` (csrc/conv_direct_r.hip): x [B,128,H,W] channels_last
    float32, 128 output channels; Wr2 = `direct_r2_pair_weights(weight)`; amax_in / amax_out as `conv3x3_direct_h`."""
    lib = _lib.load()
    x = x.contiguous(memory_format=torch.channels_last)
    B, Cin, H, W = x.shape
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = torch.empty((B, 128, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    _lib.check(lib.cslam_conv3x3_direct_r2_dev(_p(x), _p(Wr2[0]), _p(bias) if bias is not None else None, B, H, W, Cin, 128,
                                               int(relu), int(pool), _p(amax_in), float(Wr2[1]),
                                               _p(amax_out) if amax_out is not None else None, _p(y), _stream(x)))
    return y


def conv3x3_direct_r_pairs(x, Wr, bias, wl1, bmax, amax_in, bound_out, amax_out=None):
    """relu(conv3x3(x) + bias), 64 -> 128 channels, written in PAIR FORMAT (`cslam_conv3x3_direct_r_pairs_dev`): returns the
    [B,H,W,4,2,32] float16 tensor scaled for the bound max|x| wl1 + bmax, which goes to the 4-byte slot bound_out."""
    lib = _lib.load()
    x = x.contiguous(memory_format=torch.channels_last)
    B, Cin, H, W = x.shape
    y = torch.empty((B, H, W, 4, 2, 32), dtype=torch.float16, device=x.device)
    _lib.check(lib.cslam_conv3x3_direct_r_pairs_dev(_p(x), _p(Wr[0]), _p(bias) if bias is not None else None, B, H, W, Cin, 128,
                                                    _p(amax_in), float(Wr[1]), float(wl1), float(bmax),
                                                    _p(amax_out) if amax_out is not None else None, _p(bound_out), _p(y), _stream(x)))
    return y


def conv3x3_direct_hp(xp, shape, bound, Wd, bias, relu, pool, amax_out=None):
    """`conv3x3_direct_h` reading a PAIR-FORMAT map xp [B,H,W,Cin/32,2,32] float16 (shape = its (B,Cin,H,W), bound = its 4-byte bound
    slot) through `cslam_conv3x3_direct_hp_dev`: the patch is staged without conversion."""
    lib = _lib.load()
    B, Cin, H, W = shape
    W2, inv_sw = Wd
    Cout = W2.shape[1]
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=xp.device, memory_format=torch.channels_last)
    _lib.check(lib.cslam_conv3x3_direct_hp_dev(_p(xp), _p(bound), _p(W2), _p(bias) if bias is not None else None, B, H, W, Cin, Cout,
                                               int(relu), int(pool), float(inv_sw), _p(amax_out) if amax_out is not None else None,
                                               _p(y), _stream(xp)))
    return y


def fused64_weights(U):
    """U [16 | 36, 64, Cout] (`wino_weights(w, 2 | 4)`; Cout 64 or 128) -> the operand order of
    `cslam_wino2_fused_c64_dev` / `cslam_wino4_fused_c64_dev`: Up[kq][xi][w][g][c][s] = U[xi][16 kq + 4 g + s][16 w + c]
    (one float4 per MFMA lane and frequency)."""
    assert U.shape[0] in (16, 36) and U.shape[1] == 64 and U.shape[2] in (64, 128)
    return U.view(U.shape[0], 4, 4, 4, U.shape[2] // 16, 16).permute(1, 0, 4, 2, 5, 3).contiguous()


def fused64_pair_weights(U4):
    """U4 [36, 64, Cout] float32 (`wino_weights(w, 4)`; Cout 64 or 128) -> (Uh int32 [4, 36, Cout/16, 4, 16, 4], inv_su):
    the weight operand of `cslam_wino4_fused_c64_h_dev` (csrc/wino_fused_h.hip): sU U split into exact fp16 pairs and
    packed one dword per value, [uh | ul << 16], in the lane order of `fused64_weights`."""
    assert U4.shape[0] == 36 and U4.shape[1] == 64 and U4.shape[2] in (64, 128)
    u = U4.detach().to(torch.float64)
    amax = float(u.abs().max())
    su = 2.0 ** (14 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    us = (u * su).to(torch.float32)
    uh = us.to(torch.float16)
    ul = (us - uh.to(torch.float32)).to(torch.float16)
    packed = (uh.view(torch.int16).to(torch.int32) & 0xFFFF) | (ul.view(torch.int16).to(torch.int32) << 16)
    cout = U4.shape[2]
    return packed.view(36, 4, 4, 4, cout // 16, 16).permute(1, 0, 4, 2, 5, 3).contiguous(), 1.0 / su


def stem_pair_weights(weight):
    """First-layer weights [64, 3, 3, 3] float32 -> (W1 int32 [4, 2, 64, 4], inv_sw, sumw float32 [64]): the operand of the
    3 -> 64 channel convolution folded into `cslam_wino4_stem_c64_h_dev` (csrc/wino_fused_h.hip).  sW w is split into exact
    fp16 pairs (sW the power of two that brings max |w| into [2^14, 2^15)); K slot (lane group g, slot j) of the 16x16x32
    MFMA holds tap (ky = g, kx = j // 3, ci = j % 3) for g < 3 and the ninth tap (ky = j, kx = 2, ci = 2) of every row for
    g = 3, j < 3 (zeros elsewhere); W1[kq][0 | 1][16 g + n][d] = halves 2d, 2d + 1 of the hi | lo parts for output channel
    16 kq + n.  sumw[co] = sum |w[co]| (float64, rounded up to float32): the kernel bounds max |first-layer output| with it."""
    assert tuple(weight.shape) == (64, 3, 3, 3)
    w = weight.detach().to(torch.float64).cpu()
    amax = float(w.abs().max())
    sw = 2.0 ** (14 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    ws = (w * sw).to(torch.float32)
    wh = ws.to(torch.float16)
    wl = (ws - wh.to(torch.float32)).to(torch.float16)
    slots = torch.zeros((2, 64, 4, 8), dtype=torch.float16)              # [hi | lo][co][g][j]
    for g in range(3):
        for j in range(8):
            slots[0, :, g, j] = wh[:, j % 3, g, j // 3]
            slots[1, :, g, j] = wl[:, j % 3, g, j // 3]
    for j in range(3):
        slots[0, :, 3, j] = wh[:, 2, j, 2]
        slots[1, :, 3, j] = wl[:, 2, j, 2]
    bits = slots.view(torch.int16).to(torch.int32) & 0xFFFF
    packed = bits[..., 0::2] | (bits[..., 1::2] << 16)                   # [2][64][4 g][4 d]
    W1 = packed.view(2, 4, 16, 4, 4).permute(1, 0, 3, 2, 4).reshape(4, 2, 64, 4).contiguous()   # [kq][hl][16 g + n][d]
    sumw = torch.nextafter(w.abs().sum(dim=(1, 2, 3)).to(torch.float32), torch.tensor(float("inf")))
    return W1.to(weight.device), 1.0 / sw, sumw.to(weight.device).contiguous()


def stem_direct_pair_weights(weight):
    """Second-layer weights [64, 64, 3, 3] float32 -> (W2r float16 [4, 9, 2, 2, 64, 8], inv_sw): the register-resident operand of
    `cslam_conv_stem_direct_h_dev` (csrc/conv_stem_direct_h.hip).  sW w (sW the power of two that brings max |w| into
    [2^14, 2^15)) is split into exact fp16 pairs; W2r[q][tap][ks][hi | lo][lane][e] = the pair half of
    w[16 q + lane % 16][32 ks + 8 (lane // 16) + e][tap // 3][tap % 3]: one v_mfma_f32_16x16x32_f16 A fragment per (q, tap, ks, half),
    wave q of a workgroup -- the owner of output channels 16 q .. 16 q + 15 -- holding [q] for the whole kernel."""
    assert tuple(weight.shape) == (64, 64, 3, 3)
    w = weight.detach().to(torch.float64)
    amax = float(w.abs().max())
    sw = 2.0 ** (14 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    ws = (w * sw).to(torch.float32)
    wh = ws.to(torch.float16)
    wl = (ws - wh.to(torch.float32)).to(torch.float16)
    pair = torch.stack((wh, wl), dim=0).reshape(2, 4, 16, 2, 4, 8, 9)       # [hl][q][i][ks][kg][e][tap]
    W2r = pair.permute(1, 6, 3, 0, 4, 2, 5).reshape(4, 9, 2, 2, 64, 8)      # [q][tap][ks][hl][lane = 16 kg + i][e]
    return W2r.contiguous(), 1.0 / sw


def conv_stem_direct_h(x0, stem, bias1, Wr, bias, pool, amax_x0, amax_out=None):
    """VGG-16's first two convolutions as ONE direct kernel (`cslam_conv_stem_direct_h_dev`): x0 planar [B,3,H,W] float32,
    stem = `stem_pair_weights(conv1_1.weight)`, Wr = `stem_direct_pair_weights(conv1_2.weight)`; amax_x0 = 4-byte device slot with
    the bits of max |x0|.  Returns ReLU(conv(ReLU(conv(x0) + bias1)) + bias) (+ MaxPool2d), channels_last."""
    lib = _lib.load()
    B, C3, H, W = x0.shape
    assert C3 == 3 and x0.is_contiguous()
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = torch.empty((B, 64, Ho, Wo), dtype=torch.float32, device=x0.device, memory_format=torch.channels_last)
    _lib.check(lib.cslam_conv_stem_direct_h_dev(
        _p(x0), _p(stem[0]), _p(bias1) if bias1 is not None else None, _p(stem[2]), float(stem[1]), _p(Wr[0]),
        _p(bias) if bias is not None else None, float(Wr[1]), B, H, W, int(pool), _p(amax_x0),
        _p(amax_out) if amax_out is not None else None, _p(y), _stream(x0)))
    return y


def wino_stem64_h(x0, stem, bias1, Uh, bias, pool, amax_x0, amax_out=None):
    """VGG-16's first two convolutions as ONE kernel (`cslam_wino4_stem_c64_h_dev`): x0 planar [B,3,H,W] float32,
    stem = `stem_pair_weights(conv1_1.weight)`, Uh = `fused64_pair_weights` of the 64 -> 64 layer; amax_x0 = 4-byte device
    slot with the bits of max |x0|.  Returns ReLU(conv(ReLU(conv(x0) + bias1)) + bias) (+ MaxPool2d), channels_last."""
    lib = _lib.load()
    B, C3, H, W = x0.shape
    assert C3 == 3 and x0.is_contiguous() and Uh[0].shape[2] == 4
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = torch.empty((B, 64, Ho, Wo), dtype=torch.float32, device=x0.device, memory_format=torch.channels_last)
    _lib.check(lib.cslam_wino4_stem_c64_h_dev(
        _p(x0), _p(stem[0]), _p(bias1) if bias1 is not None else None, _p(stem[2]), float(stem[1]), _p(Uh[0]),
        _p(bias) if bias is not None else None, B, H, W, int(pool), _p(amax_x0), float(Uh[1]),
        _p(amax_out) if amax_out is not None else None, _p(y), _stream(x0)))
    return y


def wino_fused64_h(x, Uh, bias, relu, pool, amax_in, amax_out=None, residual=None):
    """The fp16-pair form of `wino_fused64` (csrc/wino_fused_h.hip): Uh = `fused64_pair_weights(U4)`; amax_in = 4-byte
    device slot holding the bits of (a bound of) max |x|; amax_out (zeroed slot or None) receives those of max |y|."""
    lib = _lib.load()
    B, _, H, W = x.shape
    Cout = Uh[0].shape[2] * 16
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if residual is not None:
        residual = residual.contiguous(memory_format=torch.channels_last)
        assert residual.shape == y.shape
    _lib.check(lib.cslam_wino4_fused_c64_h_dev(
        _p(x), _p(Uh[0]), _p(bias) if bias is not None else None, _p(residual) if residual is not None else None,
        B, H, W, Cout, int(relu), int(pool), _p(amax_in), float(Uh[1]), _p(amax_out) if amax_out is not None else None,
        _p(y), _stream(x)))
    return y


def wino_fused64(x, Up, bias, relu, pool, residual=None):
    """64 -> 64 / 128 channel 3x3 convolution of x [B,64,H,W] (channels_last storage) as one kernel
    (csrc/wino_fused.hip); Up from `fused64_weights` (16 frequencies: the F(2x2) kernel, 36: the F(4x4) one)."""
    lib = _lib.load()
    B, _, H, W = x.shape
    Cout = Up.shape[2] * 16
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if residual is not None:
        residual = residual.contiguous(memory_format=torch.channels_last)
        assert residual.shape == y.shape
    fn = lib.cslam_wino4_fused_c64_dev if Up.shape[1] == 36 else lib.cslam_wino2_fused_c64_dev
    _lib.check(fn(_p(x), _p(Up), _p(bias) if bias is not None else None,
                  _p(residual) if residual is not None else None, B, H, W, Cout, int(relu), int(pool), _p(y), _stream(x)))
    return y


_TUNED = {"done": False}


def use_tuned_gemms():
    """Pick the fastest library solution for the strided-batched fp32 GEMM shapes of the VGG-16 trunk at the
    256-frame chunk bench.py and the batched callers use: torch's TunableOp replays the selections recorded on an
    MI355X in `tunableop_gfx950.csv` (tuning itself stays off, unknown shapes keep the library default, and a
    library / architecture mismatch makes torch ignore the file).  +8 % frames/s over the default heuristic.
    Regenerate with `PYTORCH_TUNABLEOP_ENABLED=1 python tools/extract_leg.py`.  CSLAM_TUNED_GEMM=0 disables."""
    if _TUNED["done"] or os.environ.get("CSLAM_TUNED_GEMM", "1") == "0" or not torch.cuda.is_available():
        return
    _TUNED["done"] = True
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
    try:
        import torch.cuda.tunable as tunable
        if tunable.is_enabled() or not os.path.exists(path):
            return                                   # the user drives TunableOp themselves
        tunable.enable(True)
        tunable.tuning_enable(False)
        tunable.read_file(path)
    except Exception as e:                            # never fatal: the default solutions are still correct
        import warnings
        warnings.warn("cslam_amd: tuned GEMM table not loaded (%s)" % e)


def _z_form(cin, cout):
    """Z form of the pair products (GEMM + column half of the output transform, 24 planes instead of 36) where it pays:
    measured per layer at the 256-frame chunk (profiles/r03_v4_perf_zform.log) the output kernel gains ~30 % everywhere, but
    the GEMM's 128 x 128 tiles (all the 128 Z registers per lane leave room for) lose more than that wherever the product is
    not purely HBM-bound -- only conv2_2 (128 -> 128 channels at 112 x 112: 2.08 -> 1.78 ms) comes out ahead.
    Z_FORM_MAX = largest Cin * Cout that takes it (128 * 128; 0 = off, 262144 = every layer: tests/test_wino_gemm_gpu.py runs both)."""
    return cin * cout <= Z_FORM_MAX


Z_FORM_MAX = 128 * 128
PAIR_ACTS = True              # ResNet trunks: pair-format maps between the implicit-GEMM layers (False: float32 maps, the A/B partner)
VGG_PAIRS = False             # VGG trunk: the map between conv2_1 and conv2_2 in pair format.  Built, tested and measured in round 6: conv2_2 staging pairs
                              # without the split 2.11 against 2.135 ms, conv2_1 writing them 1.15 against 1.04 -- the split was already hidden under
                              # the MFMAs, the two 8-byte stores per row are not (profiles/r06_j_vgg_pairs_ab.log): off by default
DIRECT_P = True               # ResNet trunks: the 64 -> 64 3x3 layers between pair-format maps through csrc/conv_direct_p.hip (False: the implicit GEMM)
IGEMM_CONVS = True            # ResNet trunks: strided / 1x1 / 7x7 layers through csrc/conv_igemm.hip (False: torch, the A/B partner)


def wino_conv3x3(ws, x, U, U4, bias, relu, pool=False, residual=None, U3=None, amax_in=None, amax_out=None, U2=None):
    """3x3 / stride 1 / pad 1 convolution of x [B,Cin,H,W] (channels_last storage, any H and W) through the
    Winograd pipeline; U / U4 from `wino_weights` (U4 None = F(2x2,3x3) only).  bias [Cout] or None, residual
    (channels_last, shaped like the output) is added before the ReLU.  `ws` owns the V / M workspaces.
    U2 = `split16_pair_weights(U4)` (the default for wide layers): the 36 F(4x4) products run in this library's GEMM
    (csrc/wino_gemm.hip) over the exact fp16 hi / lo pairs of both operands, three of the four partial products in fp32
    accumulators: fp32-grade, at the fp16 MFMA rate, V and M at their fp32 sizes.
    U3 = `split16_weights(U4)` (round 1's form, split16_h3=True): the same arithmetic as ONE library fp16 GEMM over
    K' = 3 Cin, operands [vh | vl | vh] x [uh ; uh ; ul].
    amax_in: 4-byte device slot already holding the bits of (a bound of) max |x| -- saves the pass over x; amax_out: zeroed
    slot that receives the same for y from the F(4x4) output transform.  Returns y; `ws.amax_written` says whether amax_out
    was filled (only the F(4x4) output kernel does it)."""
    lib = _lib.load()
    B, Cin, H, W = x.shape
    Cout = U.shape[2]
    # F(4x4) needs enough tiles to keep its 36 GEMMs efficient (single frames stay on F(2x2)) and maps whose
    # sides waste at most ~1/3 of the padded tile area (14 -> 16, 7 -> 8); tiles may hang over the map
    t4h, t4w, t2h, t2w = -(-H // 4), -(-W // 4), -(-H // 2), -(-W // 2)
    four = U4 is not None and B * t4h * t4w >= 512 and 16 * t4h * t4w <= 1.35 * H * W
    n2, Uu = (36, U4) if four else (16, U)
    T = B * t4h * t4w if four else B * t2h * t2w
    if four and U2 is not None:
        # this library's GEMM on exact fp16 pairs (csrc/wino_gemm.hip): V stored once at its fp32 size, M in fp32
        s = _stream(x)
        slot = amax_in
        if slot is None:
            slot = ws._buf("amax", 1, x.device)
            _lib.check(lib.cslam_absmax_dev(_p(x), x.numel(), _p(slot), s))
        V2 = ws._buf("V", 36 * T * Cin, x.device)                      # 36 x T x 2 Cin halfs
        M = ws._buf("M", 36 * T * Cout, x.device)
        _lib.check(lib.cslam_wino4_input_h2_dev(_p(x), B, H, W, Cin, _p(slot), _p(V2), s))
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        if residual is None and _z_form(Cin, Cout):
            # the purely HBM-bound product (conv2_2): the column half of the output transform is folded into the GEMM, which then
            # writes -- and the output kernel reads -- 24 instead of 36 planes (csrc/wino_gemm.hip `wino_zgemm_h2_kernel`)
            _lib.check(lib.cslam_wino_zgemm_h2_dev(_p(V2), _p(U2[0]), T, Cin, Cout, _p(M), s))
            _lib.check(lib.cslam_wino4_output_z_dev(
                _p(M), _p(bias) if bias is not None else None, B, H, W, Cout, int(relu), int(pool), _p(slot), float(U2[1]),
                _p(amax_out) if amax_out is not None else None, _p(y), s))
            ws.amax_written = amax_out is not None
            return y
        _lib.check(lib.cslam_wino_gemm_h2_dev(_p(V2), _p(U2[0]), T, Cin, Cout, _p(M), s))
        if residual is not None:
            residual = residual.contiguous(memory_format=torch.channels_last)
            assert residual.shape == y.shape and not pool
        _lib.check(lib.cslam_wino4_output_scaled_dev(
            _p(M), _p(bias) if bias is not None else None, _p(residual) if residual is not None else None,
            B, H, W, Cout, int(relu), int(pool), _p(slot), float(U2[1]), _p(amax_out) if amax_out is not None else None,
            _p(y), s))
        ws.amax_written = amax_out is not None
        return y
    if four and U3 is not None:
        s = _stream(x)
        slot = amax_in
        if slot is None:
            slot = ws._buf("amax", 1, x.device)                         # 4 bytes: bits of max |x|
            _lib.check(lib.cslam_absmax_dev(_p(x), x.numel(), _p(slot), s))
        V3 = ws._buf("V", (36 * T * 3 * Cin + 1) // 2, x.device).view(torch.float16)[:36 * T * 3 * Cin].view(36, T, 3 * Cin)
        _lib.check(lib.cslam_wino4_input_h3_dev(_p(x), B, H, W, Cin, _p(slot), _p(V3), s))
        # torch 2.10's TunableOp does not cover `bmm` with out_dtype; hipBLASLt's default solution is used.  Routing it to
        # rocBLAS instead is 3-13 % faster on the isolated GEMMs and not measurable on the trunk (profiles/r01_exp_split16.log)
        M = torch.bmm(V3, U3[0], out_dtype=torch.float32)
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        if residual is not None:
            residual = residual.contiguous(memory_format=torch.channels_last)
            assert residual.shape == y.shape and not pool
        _lib.check(lib.cslam_wino4_output_scaled_dev(
            _p(M), _p(bias) if bias is not None else None, _p(residual) if residual is not None else None,
            B, H, W, Cout, int(relu), int(pool), _p(slot), float(U3[1]), _p(amax_out) if amax_out is not None else None,
            _p(y), s))
        ws.amax_written = amax_out is not None
        return y
    ws.amax_written = False
    V = ws._buf("V", n2 * T * Cin, x.device).view(n2, T, Cin)
    M = ws._buf("M", n2 * T * Cout, x.device).view(n2, T, Cout)
    s = _stream(x)
    fin, fout = (lib.cslam_wino4_input_dev, lib.cslam_wino4_output_dev) if four else \
        (lib.cslam_wino_input_dev, lib.cslam_wino_output_dev)
    _lib.check(fin(_p(x), B, H, W, Cin, _p(V), s))                           # x's storage is NHWC
    torch.bmm(V, Uu, out=M)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if residual is not None:
        residual = residual.contiguous(memory_format=torch.channels_last)
        assert residual.shape == y.shape and not pool
    if four and amax_out is not None:
        _lib.check(lib.cslam_wino4_output_scaled_dev(
            _p(M), _p(bias) if bias is not None else None, _p(residual) if residual is not None else None,
            B, H, W, Cout, int(relu), int(pool), None, 1.0, _p(amax_out), _p(y), s))
        ws.amax_written = True
        return y
    _lib.check(fout(_p(M), _p(bias) if bias is not None else None, _p(residual) if residual is not None else None,
                    B, H, W, Cout, int(relu), int(pool), _p(y), s))
    return y


class _Workspace(object):
    def __init__(self):
        self._ws = {}
        self.amax_written = False

    def _buf(self, name, numel, device):
        b = self._ws.get(name)
        if b is None or b.numel() < numel or b.device != device:
            b = torch.empty(numel, dtype=torch.float32, device=device)
            self._ws[name] = b
        return b[:numel]


def fold_bn(conv, bn):
    """Eval-mode BatchNorm folded into the preceding convolution: (weight * s[co], beta - mean * s (+ bias * s)),
    s = gamma / sqrt(var + eps).  Exact in real arithmetic; computed in float64."""
    s = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps))
    w = (conv.weight.double() * s[:, None, None, None]).float().contiguous(memory_format=torch.channels_last)
    b = bn.bias.double() - bn.running_mean.double() * s
    if conv.bias is not None:
        b = b + conv.bias.double() * s
    return w, b.float().contiguous()


class _FoldedConv(object):
    """conv + eval BatchNorm as one convolution; 3x3 / stride 1 / pad 1 ones also carry Winograd weights."""

    def __init__(self, conv, bn, tile, min_in_channels, direct=False):
        self.weight, self.bias = fold_bn(conv, bn)
        self.stride, self.padding = conv.stride, conv.padding
        self.U = self.U4 = self.Wg = self.Wp = None
        self.kernel = tuple(conv.kernel_size)
        # the bound of the output the pair-format chain scales by: |y| <= max |x| wl1 + bmax (+ max |shortcut|)
        self.wl1 = float(self.weight.detach().abs().sum(dim=(1, 2, 3)).max())
        self.bmax = float(self.bias.detach().abs().max()) if self.bias is not None else 0.0
        self.in_channels = conv.in_channels
        igemm_ok = (IGEMM_CONVS and conv.dilation == (1, 1) and conv.groups == 1 and conv.stride[0] == conv.stride[1]
                    and conv.padding[0] == conv.padding[1] and conv.out_channels % 64 == 0 and conv.weight.is_cuda
                    and (conv.in_channels % 32 == 0 or (conv.in_channels == 3 and 3 * conv.kernel_size[1] <= 32)))
        if direct and igemm_ok:
            # every convolution of the trunk as this library's implicit GEMM on fp16 pairs (csrc/conv_igemm.hip): per layer faster than
            # the fp32 Winograd pipeline on ResNet-18's maps (profiles/r05_v23_igemm_layers.log), no library product anywhere
            self.Wg = igemm_pair_weights(self.weight)
            self.Wg = (self.Wg[0].to(self.weight.device), self.Wg[1])
            if (DIRECT_P and tuple(self.weight.shape) == (64, 64, 3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)):
                # layer1's 64 -> 64 convolutions between pair-format maps: the register-resident direct kernel (csrc/conv_direct_p.hip)
                self.Wp = stem_direct_pair_weights(self.weight)
                self.Wp = (self.Wp[0].to(self.weight.device), self.Wp[1])
        elif (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
                and conv.groups == 1 and conv.in_channels >= min_in_channels and conv.in_channels % 4 == 0
                and conv.out_channels % 4 == 0 and conv.weight.is_cuda):
            self.U = wino_weights(self.weight).to(self.weight.device)
            self.U4 = wino_weights(self.weight, 4).to(self.weight.device) if tile == 4 else None
        elif igemm_ok:
            # the 7x7 stem, the strided 3x3 and the 1x1 shortcut layers: this library's implicit GEMM on fp16 pairs
            # (csrc/conv_igemm.hip) instead of torch / MIOpen
            self.Wg = igemm_pair_weights(self.weight)
            self.Wg = (self.Wg[0].to(self.weight.device), self.Wg[1])
        # (layer1 of ResNet-18/34 through the one-kernel F(4x4) forms -- csrc/wino_fused.hip, round 1: 30.9k vs 31.8k frames/s; csrc/
        # wino_fused_h.hip on fp16 pairs with the shortcut fused, round 5 against the implicit GEMM: extract 45.9k vs 51.3k frames/s,
        # three alternating runs on one box -- is slower on 56 x 56 maps.  Not wired in.)

    def __call__(self, ws, x, relu, residual=None, amax_in=None, amax_out=None, pool=False):
        """amax_in: 4-byte device slot with (a bound of) max |x|, or None; amax_out: zeroed slot for max |y|.  `ws.amax_written` says
        whether amax_out was filled (the implicit-GEMM kernel always does)."""
        ws.amax_written = False
        if self.U is not None:
            return wino_conv3x3(ws, x.contiguous(memory_format=torch.channels_last), self.U, self.U4, self.bias, relu,
                                False, residual)
        if self.Wg is not None and x.shape[2] + 2 * self.padding[0] >= self.kernel[0] and x.shape[3] + 2 * self.padding[0] >= self.kernel[1]:
            # (a map smaller than the kernel: torch below, as for every geometry the implicit GEMM does not take)
            y = conv_igemm(ws, x, self.Wg, self.bias, self.kernel, self.stride[0], self.padding[0], relu, residual, amax_in, amax_out, pool)
            ws.amax_written = amax_out is not None
            return y
        y = torch.nn.functional.conv2d(x, self.weight, self.bias, self.stride, self.padding)
        if residual is not None:
            y += residual
        return torch.relu_(y) if relu else y


class WinogradResNet(_Workspace):
    """Runs the ResNet trunks of vpr/backbones.py (`resnet_trunk`: conv1, bn1, relu, maxpool, layer1..4 of
    BasicBlock / Bottleneck) like `trunk(x)` in eval mode, with every BatchNorm folded into its convolution.  Since round 5 EVERY
    convolution is a kernel of this library on fp16 pairs (`direct`, the default): the implicit GEMM (csrc/conv_igemm.hip), the
    register-resident direct kernel for layer1's 64 -> 64 layers (csrc/conv_direct_p.hip, round 6), pair-format maps between them --
    `frontend.backbone_conv` 'winograd' and 'winograd2' therefore run the SAME path for ResNets (the name is the VGG trunk's; `tile`
    only matters with direct=False: the 3x3 / stride-1 layers through the fp32 Winograd pipeline with library products, rounds 1-4's
    A/B partner; `IGEMM_CONVS = False` additionally sends the stem, strided and 1x1 layers to torch)."""

    def __init__(self, trunk, min_in_channels=64, tile=4, direct=None):
        """direct (default: IGEMM_CONVS): EVERY eligible convolution through the implicit GEMM on fp16 pairs; False: the 3x3 / stride 1
        layers through the fp32 Winograd pipeline with library products (rounds 1-4; the A/B partner)."""
        super().__init__()
        self.trunk, self.min_in_channels, self.tile = trunk, int(min_in_channels), int(tile)
        self.direct = IGEMM_CONVS if direct is None else bool(direct)
        use_tuned_gemms()
        self.refresh()

    def refresh(self):
        mods = list(self.trunk)
        mk = lambda c, b: _FoldedConv(c, b, self.tile, self.min_in_channels, self.direct)      # noqa: E731
        self.stem = mk(mods[0], mods[1])
        self.stem_pool = mods[3]
        self.blocks = []
        for layer in mods[4:]:
            for blk in layer:
                d = {"down": None if blk.downsample is None else mk(blk.downsample[0], blk.downsample[1]),
                     "c1": mk(blk.conv1, blk.bn1), "c2": mk(blk.conv2, blk.bn2),
                     "c3": mk(blk.conv3, blk.bn3) if hasattr(blk, "conv3") else None}
                self.blocks.append(d)
        return self

    @torch.no_grad()
    def __call__(self, x, x_bound=None):
        """x_bound: a known bound of max |x| (a normalised 8-bit image: heads.normalised_image_bound()) spares the pass that measures it."""
        x = x.contiguous(memory_format=torch.channels_last)
        # max |activation| travels from the epilogue that produced a map to the kernels that read it (4-byte device slots, the power-of-two
        # scale of the fp16 pairs): no pass over an activation just to measure it.  `ax` = slot of the current x, or None (unknown)
        slots = self._buf("amax_slots", 4 * len(self.blocks) + 6, x.device)
        slots.zero_()
        nslot = [0]

        def fresh():
            nslot[0] += 1
            return slots[nslot[0] - 1:nslot[0]]

        def run(conv, inp, relu, res, a_in):
            out_slot = fresh()
            y = conv(self, inp, relu, res, a_in, out_slot)
            return y, (out_slot if self.amax_written else None)
        a0 = None
        if x_bound is not None:
            a0 = fresh()
            a0.fill_(float(x_bound))
        sp = self.stem_pool
        ho = (x.shape[2] + 2 * self.stem.padding[0] - self.stem.kernel[0]) // self.stem.stride[0] + 1
        wo = (x.shape[3] + 2 * self.stem.padding[1] - self.stem.kernel[1]) // self.stem.stride[1] + 1
        if (self.stem.Wg is not None and x.shape[1] == 3 and stem_pool_fits(ho, wo) and isinstance(sp, torch.nn.MaxPool2d)
                and _pair(sp.kernel_size) == (3, 3) and _pair(sp.stride) == (2, 2) and _pair(sp.padding) == (1, 1)
                and _pair(sp.dilation) == (1, 1) and not sp.ceil_mode):
            out_slot = fresh()                            # conv1 + bn1 + relu + maxpool as one kernel: the 112 x 112 map never exists
            x = self.stem(self, x, True, None, a0, out_slot, pool=True)
            ax = out_slot
        else:
            y, ax = run(self.stem, x, True, None, a0)
            x = sp(y)                                     # max |pool(y)| <= max |y|: the slot stays a bound
        convs = [c for b in self.blocks for c in (b["down"], b["c1"], b["c2"], b["c3"]) if c is not None]
        if PAIR_ACTS and ax is not None and all(c.Wg is not None and c.in_channels % 32 == 0 for c in convs):
            # every layer is the implicit GEMM: the maps between them travel in PAIR FORMAT (written once by the producing epilogue, read
            # by LDS-DMA: no split per tap and output tile); the pooled stem output goes in as float32, the last map comes out as float32
            pslots = self._buf("pair_slots", 2 * len(convs) + 2, x.device)
            pslots.zero_()
            np_ = [0]
            last = convs[-1]

            def runp(conv, a, relu, res):
                np_[0] += 2
                if (conv.Wp is not None and (a.pairs or res is None) and conv is not last
                        and direct_p_fits(conv.weight.shape, conv.kernel, conv.stride[0], conv.padding[0], a.shape[2], a.shape[3])):
                    return conv3x3_direct_p(a, conv.Wp, conv.bias, relu, res, conv.wl1, conv.bmax, pslots[np_[0] - 2:np_[0] - 1],
                                            pslots[np_[0] - 1:np_[0]], True)
                return conv_igemm_p(self, a, conv.Wg, conv.bias, conv.kernel, conv.stride[0], conv.padding[0], relu, res, conv.wl1,
                                    conv.bmax, pslots[np_[0] - 2:np_[0] - 1], pslots[np_[0] - 1:np_[0]], conv is not last)
            cur = PairAct(x, False, x.shape, ax, ax)
            for b in self.blocks:
                idt = cur if b["down"] is None else runp(b["down"], cur, False, None)
                o = runp(b["c1"], cur, True, None)
                if b["c3"] is None:                   # BasicBlock
                    cur = runp(b["c2"], o, True, idt)
                else:                                 # Bottleneck
                    o = runp(b["c2"], o, True, None)
                    cur = runp(b["c3"], o, True, idt)
            return cur.t
        for b in self.blocks:
            idt = x if b["down"] is None else run(b["down"], x, False, None, ax)[0]
            o, ao = run(b["c1"], x, True, None, ax)
            if b["c3"] is None:                       # BasicBlock
                x, ax = run(b["c2"], o, True, idt, ao)
            else:                                     # Bottleneck
                o, ao = run(b["c2"], o, True, None, ao)
                x, ax = run(b["c3"], o, True, idt, ao)
        return x


class _Step(object):
    __slots__ = ("kind", "module", "conv", "relu", "pool", "U", "U4", "U3", "U2", "Up", "Uph", "bias", "stem", "Wd", "Wr", "Wdr", "Wdr2", "wl1", "bmax")

    def __init__(self):
        self.kind, self.module, self.conv, self.relu, self.pool = "torch", None, None, False, False
        self.U, self.U4, self.U3, self.U2, self.Up, self.Uph, self.bias, self.stem = None, None, None, None, None, None, None, None
        self.Wd = None
        self.Wr = None
        self.Wdr = None
        self.Wdr2 = None
        self.wl1 = self.bmax = None


# Which form every layer of a VGG-style trunk takes (the defaults are the measured best; the others are the A/B partners the tests and
# tools/ select through WinogradTrunk(forms={...}) -- no environment variables):
#   conv_direct      1: conv2_1 and conv2_2 as the direct one-kernel convolution on fp16 pairs; 2: conv2_2 only; 0: round 3's F(4x4) forms
#   conv_direct_r    conv2_1 through the register-resident direct kernel (csrc/conv_direct_r.hip); False: the streaming one
#   conv_direct_r2   conv2_2 through the register-resident kernel on output-channel halves (csrc/conv_direct_r.hip, round 6); False: the
#                    kernel with the weights through an LDS ring (csrc/conv_direct_h.hip)
#   stem_direct      conv1_1 + conv1_2 as ONE direct kernel (csrc/conv_stem_direct_h.hip); False: the one-kernel F(4x4) stem
#   wino_stem        conv1_1 folded into conv1_2's kernel at all; False: conv1_1 as its own fp32 kernel
#   fused_h          the one-kernel F(4x4) convolutions on fp16 pairs (csrc/wino_fused_h.hip); False: the f32-input MFMA kernels
#   split16_min_cin  F(4x4) layers from that many input channels on run their 36 products on fp16 pairs (None: 128, or 256 with
#                    split16_h3); 0: plain fp32 library GEMMs everywhere (bench.py's `value_fp32_gemms`)
TRUNK_FORMS = {"conv_direct": 1, "conv_direct_r": True, "conv_direct_r2": True, "stem_direct": True, "wino_stem": True, "fused_h": True, "split16_min_cin": None}
FP32_GEMM_FORMS = {"split16_min_cin": 0, "fused_h": False, "wino_stem": False}      # the trunk on plain fp32 library GEMMs


class WinogradTrunk(_Workspace):
    """Runs an nn.Sequential of Conv2d / ReLU / MaxPool2d like `encoder(x)`, with the eligible
    convolutions (+ their ReLU, + their MaxPool2d(2,2)) replaced by the Winograd pipeline."""

    def __init__(self, encoder, min_in_channels=256, tile=2, fused64=None, split16_h3=False, forms=None):
        """tile = 2: F(2x2,3x3) everywhere; tile = 4: F(4x4,3x3) on the maps whose sides are multiples of 4
        (F(2x2,3x3) on the others).  fused64: run the 64 -> 64 / 128 channel layers (VGG-16 conv1_2, conv2_1) through the single
        fused F(2x2,3x3) kernel instead of transform / GEMM / transform (default on)."""
        super().__init__()
        self.encoder = encoder
        self.min_in_channels = int(min_in_channels)
        self.tile = int(tile)
        self.fused64 = True if fused64 is None else bool(fused64)
        self.fused_min_blocks = 256                  # fewer tile blocks than this (single frames): the three-kernel form
        self.fused_couts = (64, 128)
        # split-fp16 GEMMs on the F(4x4) layers from this many input channels on (0 = off: plain fp32 library GEMMs).
        # Default: this library's pair GEMM (`split16_pair_weights`, csrc/wino_gemm.hip) from 128 channels on -- V is no
        # larger than its fp32 form, so every layer the three-kernel form runs gains.  split16_h3=True (constructor) selects round 1's
        # library GEMM over [vh | vl | vh] instead, whose measured optimum was 256 (profiles/r01_exp_split16.log): a test partner.
        # a known bound of max |input| (e.g. a normalised 8-bit image: heads.normalised_image_bound()) spares the stem kernel
        # its pass over the input; None = measured per call
        self.input_bound = None
        # forms['conv_direct'] = 0: conv2_1 / conv2_2 through the F(4x4) forms of round 3 (the A/B partner)
        self.forms = dict(TRUNK_FORMS)
        self.forms.update(forms or {})
        assert set(self.forms) == set(TRUNK_FORMS), "unknown trunk form"
        self.direct128 = int(self.forms["conv_direct"]) != 0
        # input widths that take the direct kernel (conv_direct = 2: conv2_2 only, conv2_1 on the one-kernel F(4x4) form)
        self.direct_cins = (128,) if int(self.forms["conv_direct"]) == 2 else (64, 128)
        self.split16_h3 = bool(split16_h3)
        self.split16_min_cin = (256 if self.split16_h3 else 128) if self.forms["split16_min_cin"] is None else int(self.forms["split16_min_cin"])
        use_tuned_gemms()
        self.refresh()

    def refresh(self):
        """(Re)build the plan and the transformed weights from the encoder's current parameters."""
        mods = list(self.encoder)
        self.steps = []
        i = 0
        while i < len(mods):
            m = mods[i]
            st = _Step()
            ok = (isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1)
                  and m.dilation == (1, 1) and m.groups == 1 and m.in_channels >= self.min_in_channels
                  and m.in_channels % 4 == 0 and m.out_channels % 4 == 0 and m.weight.is_cuda)
            if ok:
                st.kind, st.conv, st.relu, st.pool = "wino", m, False, False
                st.U = wino_weights(m.weight).to(m.weight.device)
                st.U4 = wino_weights(m.weight, 4).to(m.weight.device) if self.tile == 4 else None
                if st.U4 is not None and 0 < self.split16_min_cin <= m.in_channels:
                    if not self.split16_h3 and m.in_channels % 32 == 0 and m.out_channels % 128 == 0:
                        st.U2 = split16_pair_weights(st.U4)
                    else:
                        st.U3 = split16_weights(st.U4)
                if self.fused64 and m.in_channels == 64 and m.out_channels in self.fused_couts:
                    # F(4x4) one-kernel form on the F(4x4) trunk, the F(2x2) one on an F(2x2) trunk
                    t4 = self.tile == 4
                    st.Up = fused64_weights(st.U4 if t4 else st.U)
                    # the fp16-pair form of the one-kernel convolution (csrc/wino_fused_h.hip); forms['fused_h'] = False keeps
                    # the f32-MFMA kernel
                    if t4 and self.forms["fused_h"]:
                        st.Uph = fused64_pair_weights(st.U4)
                if (self.direct128 and self.tile == 4 and m.out_channels == 128 and m.in_channels in self.direct_cins):
                    # VGG-16 conv2_2 (128 -> 128 on 112 x 112 maps, + MaxPool2d): the direct one-kernel form on fp16 pairs
                    # (csrc/conv_direct_h.hip) -- HBM sees the activation in and out, nothing else; the F(4x4) pipeline moved 10 x
                    # the activation there (17 of the pass's 63 GB) and was bound by it: 2.6-2.7 ms against 2.8, -14 GB.  conv2_1
                    # (64 -> 128) stays on the one-kernel F(4x4) form: 1.47 ms against the direct kernel's 1.62 (a quarter of the
                    # multiplications; measured, profiles/r04_v9_direct_conv.log)
                    st.Wd = direct_pair_weights(m.weight)
                    # 64 -> 128 (conv2_1): the register-resident form (csrc/conv_direct_r.hip); forms['conv_direct_r'] = False keeps the one above
                    if m.in_channels == 64 and self.forms["conv_direct_r"]:
                        st.Wdr = direct_r_pair_weights(m.weight)
                        st.wl1 = float(m.weight.detach().abs().sum(dim=(1, 2, 3)).max())      # bound of the pair-format output: max|x| wl1 + bmax
                        st.bmax = 0.0 if m.bias is None else float(m.bias.detach().abs().max())
                    # 128 -> 128 (conv2_2): the register-resident form on output-channel halves
                    if m.in_channels == 128 and self.forms.get("conv_direct_r2", True):
                        st.Wdr2 = direct_r2_pair_weights(m.weight)
                st.bias = None if m.bias is None else m.bias.detach().to(torch.float32).contiguous()
                i += 1
                if i < len(mods) and isinstance(mods[i], nn.ReLU):
                    st.relu = True
                    i += 1
                    p = mods[i] if i < len(mods) else None
                    if isinstance(p, nn.MaxPool2d) and p.kernel_size in (2, (2, 2)) and p.stride in (2, (2, 2)) \
                            and p.padding in (0, (0, 0)) and not p.ceil_mode:
                        st.pool = True
                        i += 1
            elif (isinstance(m, nn.Conv2d) and m.in_channels == 3 and m.kernel_size == (3, 3) and m.stride == (1, 1)
                  and m.padding == (1, 1) and m.dilation == (1, 1) and m.groups == 1 and m.out_channels % 16 == 0
                  and m.out_channels <= 512 and m.weight.is_cuda and not (i + 2 < len(mods) and isinstance(
                      mods[i + 1], nn.ReLU) and isinstance(mods[i + 2], nn.MaxPool2d))):
                # first layer: hand-written direct convolution, bias + ReLU fused, planar input -> NHWC
                st.kind, st.conv = "c3", m
                st.U = m.weight.detach().to(torch.float32).permute(1, 2, 3, 0).reshape(27, m.out_channels).contiguous()
                st.bias = None if m.bias is None else m.bias.detach().to(torch.float32).contiguous()
                i += 1
                if i < len(mods) and isinstance(mods[i], nn.ReLU):
                    st.relu = True
                    i += 1
            elif (isinstance(m, nn.Conv2d) and m.groups == 1 and m.out_channels % 4 == 0 and m.bias is not None
                  and m.weight.is_cuda and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)):
                # direct convolution without its bias, then bias + ReLU (+ MaxPool) in one HIP pass
                st.kind, st.conv, st.relu = "direct", m, True
                st.bias = m.bias.detach().to(torch.float32).contiguous()
                i += 2
                p = mods[i] if i < len(mods) else None
                if isinstance(p, nn.MaxPool2d) and p.kernel_size in (2, (2, 2)) and p.stride in (2, (2, 2)) \
                        and p.padding in (0, (0, 0)) and not p.ceil_mode:
                    st.pool = True
                    i += 1
            else:
                st.kind, st.module = "torch", m
                i += 1
            self.steps.append(st)
        # first layer (3 -> 64, ReLU) followed by a one-kernel fp16-pair layer (64 -> 64, ReLU): both in ONE kernel, the
        # 64-channel map between them never reaches HBM (csrc/wino_fused_h.hip, STEM; forms['wino_stem'] = False keeps them apart)
        if self.forms["wino_stem"]:
            for a, b in zip(self.steps, self.steps[1:]):
                if (a.kind == "c3" and a.relu and a.conv.out_channels == 64 and b.kind == "wino" and b.Uph is not None
                        and b.relu and b.conv.out_channels == 64):
                    a.stem = stem_pair_weights(a.conv.weight)
                    # the direct one-kernel form of the pair (csrc/conv_stem_direct_h.hip); forms['stem_direct'] = False keeps the F(4x4) one
                    if self.forms["stem_direct"]:
                        a.Wr = stem_direct_pair_weights(b.conv.weight)
        return self

    @torch.no_grad()
    def __call__(self, x):
        """x [B,C,H,W] float32 (any memory format) -> [B,C',H',W'] float32, channels_last memory."""
        lib = _lib.load()
        # one 4-byte slot per step for max |activation| between consecutive split-fp16 layers: slot k holds max |input of
        # step k|, written by the step before it when that step can (first-layer kernel, fused fp16 kernel, F(4x4) output
        # transform); otherwise the consumer makes its own pass over x
        slots = amax_ready = None
        wants = lambda st_: st_ is not None and (st_.U3 is not None or st_.U2 is not None or st_.Uph is not None or st_.Wd is not None)   # noqa: E731
        if any(wants(st) for st in self.steps):
            slots = self._buf("amax_slots", len(self.steps) + 1, x.device)
            slots.zero_()
        skip = False
        for k, st in enumerate(self.steps):
            if skip:                                                 # this step ran inside the previous one (stem kernel)
                skip = False
                continue
            have, amax_ready = amax_ready, None
            nxt = self.steps[k + 1] if k + 1 < len(self.steps) else None
            if st.kind == "c3":
                x = x.contiguous()                                   # planar [B,3,H,W]
                B, _, H, W = x.shape
                if (st.stem is not None and B * -(-H // 16) * -(-W // 16) >= self.fused_min_blocks
                        and not (nxt.pool and (H % 2 or W % 2)) and B * H * W * 64 < 2 ** 31):
                    slot = slots[k:k + 1]
                    if self.input_bound is not None:
                        slot.fill_(float(self.input_bound))
                    elif x.numel() % 4 == 0:
                        _lib.check(lib.cslam_absmax_dev(_p(x), x.numel(), _p(slot), _stream(x)))
                    else:                                            # odd image sizes: the streaming kernel wants whole float4s
                        slot.copy_(x.abs().max().reshape(1))
                    nn2 = self.steps[k + 2] if k + 2 < len(self.steps) else None
                    want = slots[k + 2:k + 3] if wants(nn2) else None
                    if st.Wr is not None:
                        x = conv_stem_direct_h(x, st.stem, st.bias, st.Wr, nxt.bias, nxt.pool, slot, want)
                    else:
                        x = wino_stem64_h(x, st.stem, st.bias, nxt.Uph, nxt.bias, nxt.pool, slot, want)
                    amax_ready = want is not None
                    skip = True
                    continue
                Cout = st.conv.out_channels
                y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
                want = slots[k + 1:k + 2] if (wants(nxt) and Cout == 64) else None
                _lib.check(lib.cslam_conv3x3_c3_amax_dev(_p(x), _p(st.U), _p(st.bias) if st.bias is not None else None,
                                                         B, H, W, Cout, int(st.relu), _p(y),
                                                         _p(want) if want is not None else None, _stream(x)))
                amax_ready = want is not None
                x = y
                continue
            x = x.contiguous(memory_format=torch.channels_last)
            if st.kind == "torch":
                x = st.module(x)
                continue
            if st.kind == "direct":
                c = st.conv
                x = torch.nn.functional.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation)
                x = x.contiguous(memory_format=torch.channels_last)
                B, Cout, H, W = x.shape
                pool = st.pool and H % 2 == 0 and W % 2 == 0
                y = x if not pool else torch.empty((B, Cout, H // 2, W // 2), dtype=torch.float32, device=x.device,
                                                   memory_format=torch.channels_last)
                _lib.check(lib.cslam_bias_act_pool_dev(_p(x), _p(st.bias), B, H, W, Cout, 1, int(pool), _p(y), _stream(x)))
                x = y if not (st.pool and not pool) else torch.nn.functional.max_pool2d(y, 2, 2)
                continue
            x = x.contiguous(memory_format=torch.channels_last)
            if (st.Wd is not None and not (st.pool and (x.shape[2] % 2 or x.shape[3] % 2))
                    and x.shape[0] * -(-x.shape[2] // 16) * -(-x.shape[3] // 16) >= self.fused_min_blocks and x.numel() % 4 == 0):
                slot = slots[k:k + 1]
                if not have:
                    _lib.check(lib.cslam_absmax_dev(_p(x), x.numel(), _p(slot), _stream(x)))
                want = slots[k + 1:k + 2] if wants(nxt) else None
                if (VGG_PAIRS and st.Wdr is not None and st.relu and not st.pool and nxt is not None and nxt.Wd is not None
                        and nxt.Wdr is None and nxt.conv.in_channels == 128 and x.shape[2] * x.shape[3] * 512 < 2 ** 31 - 16
                        and not (nxt.pool and (x.shape[2] % 2 or x.shape[3] % 2))):
                    # conv2_1 writes pairs, conv2_2 stages them without conversion: both steps here
                    B_, _, H_, W_ = x.shape
                    bslot = self._buf("vgg_pair_bound", 1, x.device)
                    w_ = st.conv.weight.detach()
                    xp = conv3x3_direct_r_pairs(x, st.Wdr, st.bias, float(w_.abs().sum(dim=(1, 2, 3)).max()) if st.wl1 is None else st.wl1,
                                                0.0 if st.bias is None else st.bmax, slot, bslot)
                    nn2 = self.steps[k + 2] if k + 2 < len(self.steps) else None
                    want2 = slots[k + 2:k + 3] if wants(nn2) else None
                    x = conv3x3_direct_hp(xp, (B_, 128, H_, W_), bslot, nxt.Wd, nxt.bias, nxt.relu, nxt.pool, want2)
                    amax_ready = want2 is not None
                    skip = True
                    continue
                if st.Wdr is not None and x.shape[2] * x.shape[3] * 512 < 2 ** 31 - 16:
                    x = conv3x3_direct_r(x, st.Wdr, st.bias, st.relu, st.pool, slot, want)
                elif st.Wdr2 is not None and x.shape[2] * x.shape[3] * 512 < 2 ** 31 - 16:
                    x = conv3x3_direct_r2(x, st.Wdr2, st.bias, st.relu, st.pool, slot, want)
                else:
                    x = conv3x3_direct_h(x, st.Wd, st.bias, st.relu, st.pool, slot, want)
                amax_ready = want is not None
                continue
            if st.Up is not None and not (st.pool and (x.shape[2] % 2 or x.shape[3] % 2)):
                # one persistent workgroup per compute unit: worth it from one tile block per CU on (a single 224 x 224
                # frame has 196: VGG-16 at B = 1 502 us through it, 459 us through the three-kernel form)
                bh, bw = (16, 16) if st.Up.shape[1] == 36 else (8, 16)
                nblk = x.shape[0] * -(-x.shape[2] // bh) * -(-x.shape[3] // bw) * st.Up.shape[2] // 4
                if nblk >= self.fused_min_blocks:
                    if st.Uph is not None and x.numel() < 2 ** 31:
                        slot = slots[k:k + 1]
                        if not have:
                            _lib.check(lib.cslam_absmax_dev(_p(x), x.numel(), _p(slot), _stream(x)))
                        want = slots[k + 1:k + 2] if wants(nxt) else None
                        x = wino_fused64_h(x, st.Uph, st.bias, st.relu, st.pool, slot, want)
                        amax_ready = want is not None
                    else:
                        x = wino_fused64(x, st.Up, st.bias, st.relu, st.pool)
                    continue
            want = slots[k + 1:k + 2] if wants(nxt) else None
            y = wino_conv3x3(self, x, st.U, st.U4, st.bias, st.relu, st.pool, U3=st.U3, U2=st.U2,
                             amax_in=slots[k:k + 1] if have else None, amax_out=want)
            amax_ready = want is not None and self.amax_written
            x = y
        return x
