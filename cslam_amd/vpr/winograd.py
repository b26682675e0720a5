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


_TUNED = {"done": False}


def use_tuned_gemms():
    """Pick the fastest library solution for the strided-batched fp32 GEMM shapes of the VGG-16 trunk at the
    256-frame chunk bench.py and the batched callers use: torch's TunableOp replays the selections recorded on an
    MI355X in `tunableop_gfx950.csv` (tuning itself stays off, unknown shapes keep the library default, and a
    library / architecture mismatch makes torch ignore the file).  +8 % frames/s over the default heuristic.
    Regenerate with `PYTORCH_TUNABLEOP_ENABLED=1 python tools/extract_leg.py`.  CSLAM_TUNED_GEMM=0 disables."""
    if _TUNED["done"] or os.environ.get("CSLAM_TUNED_GEMM", "1") == "0":
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


class _Step(object):
    __slots__ = ("kind", "module", "conv", "relu", "pool", "U", "U4", "bias")

    def __init__(self):
        self.kind, self.module, self.conv, self.relu, self.pool = "torch", None, None, False, False
        self.U, self.U4, self.bias = None, None, None


class WinogradTrunk(object):
    """Runs an nn.Sequential of Conv2d / ReLU / MaxPool2d like `encoder(x)`, with the eligible
    convolutions (+ their ReLU, + their MaxPool2d(2,2)) replaced by the Winograd pipeline."""

    def __init__(self, encoder, min_in_channels=256, tile=2):
        """tile = 2: F(2x2,3x3) everywhere; tile = 4: F(4x4,3x3) on the maps whose sides are multiples of 4
        (F(2x2,3x3) on the others)."""
        self.encoder = encoder
        self.min_in_channels = int(min_in_channels)
        self.tile = int(tile)
        use_tuned_gemms()
        self._ws = {}
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
        return self

    def _buf(self, name, numel, device):
        b = self._ws.get(name)
        if b is None or b.numel() < numel or b.device != device:
            b = torch.empty(numel, dtype=torch.float32, device=device)
            self._ws[name] = b
        return b[:numel]

    @torch.no_grad()
    def __call__(self, x):
        """x [B,C,H,W] float32 (any memory format) -> [B,C',H',W'] float32, channels_last memory."""
        lib = _lib.load()
        for st in self.steps:
            if st.kind == "c3":
                x = x.contiguous()                                   # planar [B,3,H,W]
                B, _, H, W = x.shape
                Cout = st.conv.out_channels
                y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
                _lib.check(lib.cslam_conv3x3_c3_dev(_p(x), _p(st.U), _p(st.bias) if st.bias is not None else None,
                                                    B, H, W, Cout, int(st.relu), _p(y), _stream(x)))
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
            B, Cin, H, W = x.shape
            if H % 2 or W % 2 or H < 2 or W < 2:                # odd maps: the direct form
                x = st.conv(x)
                if st.relu:
                    x = torch.relu_(x)
                if st.pool:
                    x = torch.nn.functional.max_pool2d(x, 2, 2)
                continue
            x = x.contiguous(memory_format=torch.channels_last)
            Cout = st.conv.out_channels
            # F(4x4) needs enough tiles to keep its 36 GEMMs efficient; single frames stay on F(2x2)
            four = st.U4 is not None and H % 4 == 0 and W % 4 == 0 and B * (H // 4) * (W // 4) >= 512
            n2, U = (36, st.U4) if four else (16, st.U)
            T = B * (H // 4) * (W // 4) if four else B * (H // 2) * (W // 2)
            V = self._buf("V", n2 * T * Cin, x.device).view(n2, T, Cin)
            M = self._buf("M", n2 * T * Cout, x.device).view(n2, T, Cout)
            s = _stream(x)
            fin, fout = (lib.cslam_wino4_input_dev, lib.cslam_wino4_output_dev) if four else \
                (lib.cslam_wino_input_dev, lib.cslam_wino_output_dev)
            _lib.check(fin(_p(x), B, H, W, Cin, _p(V), s))                           # x's storage is NHWC
            torch.bmm(V, U, out=M)
            Ho, Wo = (H // 2, W // 2) if st.pool else (H, W)
            y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device,
                            memory_format=torch.channels_last)
            _lib.check(fout(_p(M), _p(st.bias) if st.bias is not None else None, B, H, W, Cout,
                            int(st.relu), int(st.pool), _p(y), s))
            x = y
        return x
