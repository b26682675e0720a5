"""CosPlace global descriptor on an MI355X (reference: cslam/vpr/cosplace.py and
cslam/vpr/cosplace_utils/{network,layers}.py).

Drop-in class `CosPlace(params, node)` with `compute_embedding(keyframe) -> np.ndarray`.
Backbone (ResNet-18/50/101/152 or VGG-16 trunk) runs on PyTorch-ROCm; the transform and the
aggregation head L2Norm -> GeM -> Flatten -> Linear -> L2Norm are HIP kernels.
"""
from os.path import isfile, join

import numpy as np
import torch
from torch import nn

from .. import _lib
from . import heads
from .backbones import get_backbone
from .winograd import WinogradResNet, WinogradTrunk
from .netvlad import _share_dir

IMAGENET_DEFAULT_MEAN = heads.IMAGENET_DEFAULT_MEAN
IMAGENET_DEFAULT_STD = heads.IMAGENET_DEFAULT_STD


class GeoLocalizationNet(object):
    """Backbone + aggregation head (reference cosplace_utils/network.py:19-35).  Parameter
    names of the reference checkpoint: 'backbone.*', 'aggregation.1.p' (GeM),
    'aggregation.3.weight|bias' (Linear)."""

    def __init__(self, backbone, fc_output_dim, device, backbone_conv="winograd"):
        self.backbone_name = backbone
        # 'winograd' | 'winograd2' | 'direct': how the 3x3 / stride 1 convolutions run (vpr/winograd.py); the
        # Winograd modes also fold every eval-mode BatchNorm into its convolution
        self.backbone_conv = backbone_conv
        self.runner = None
        self._epoch = object()
        self.backbone, self.features_dim = get_backbone(backbone)
        self.backbone = self.backbone.to(device).eval().to(memory_format=torch.channels_last)
        for p in self.backbone.parameters():
            p.requires_grad_(False)
        self.gem_p, self.gem_eps = 3.0, 1e-6
        lin = nn.Linear(self.features_dim, fc_output_dim)
        self.fc_weight = lin.weight.detach().to(device).contiguous()
        self.fc_bias = lin.bias.detach().to(device).contiguous()
        self.device = device

    def load_state_dict(self, state):
        bb = {k[len("backbone."):]: v for k, v in state.items() if k.startswith("backbone.")}
        self.backbone.load_state_dict(bb)
        self.runner = None                     # folded / transformed weights are rebuilt on the next forward
        self._epoch = object()
        self.gem_p = float(state["aggregation.1.p"].reshape(-1)[0])
        self.fc_weight = state["aggregation.3.weight"].float().to(self.device).contiguous()
        self.fc_bias = state["aggregation.3.bias"].float().to(self.device).contiguous()

    def runner_epoch(self):
        """Changes whenever the weights were reloaded (a captured graph of the old weights is then stale)."""
        return self._epoch

    def make_runner(self):
        tile = 4 if self.backbone_conv == "winograd" else 2
        return (WinogradTrunk(self.backbone, 64, tile) if self.backbone_name == "vgg16"
                else WinogradResNet(self.backbone, 64, tile))

    @torch.no_grad()
    def forward(self, x, backbone_dtype=None, runner=None, x_bound=None):
        if backbone_dtype is not None and backbone_dtype != torch.float32:
            with torch.autocast("cuda", dtype=backbone_dtype):
                f = self.backbone(x)
            f = f.float()
        elif self.backbone_conv in ("winograd", "winograd2"):
            if runner is None:
                if self.runner is None:
                    self.runner = self.make_runner()
                runner = self.runner
            f = runner(x, x_bound) if isinstance(runner, WinogradResNet) else runner(x)
        else:
            f = self.backbone(x)
        return heads.gem_fc_head(f, self.gem_p, self.gem_eps, self.fc_weight, self.fc_bias)


class CosPlace(object):
    """CosPlace matcher"""

    def __init__(self, params, node):
        self.params = params
        self.node = node
        self.enable = self.params['frontend.nn_checkpoint'].lower() != 'disable'
        self.descriptor_dim = self.params.get('frontend.cosplace.descriptor_dim', 64)
        if not self.enable:
            return
        _lib.require_gpu()
        if not torch.cuda.is_available():
            raise _lib.CslamHipError("CosPlace needs PyTorch-ROCm with a visible MI355X")
        self.device = torch.device("cuda")
        self.crop = int(self.params["frontend.image_crop_size"])
        self.backbone_conv = str(self.params.get('frontend.backbone_conv', 'winograd')).lower()
        if self.backbone_conv not in ('winograd', 'winograd2', 'direct'):
            raise ValueError("frontend.backbone_conv must be 'winograd', 'winograd2' or 'direct'")
        self.model = GeoLocalizationNet(self.params['frontend.cosplace.backbone'], self.descriptor_dim,
                                        self.device, self.backbone_conv)
        # frontend.hip_graph: true replays one-keyframe calls from a captured HIP graph of the whole pipeline.
        # Off by default: on ROCm 7.2 replaying the ~45-node graph takes 6.2 ms against 1.2 ms for launching the
        # same kernels one by one (tools/perf_online.py), so plain launches are the faster online path today.
        self.use_graph = bool(self.params.get('frontend.hip_graph', False))
        self._online = None
        self._online_runner = None
        self._online_model = None
        self._lanes, self._lanes_epoch = [], None
        ckpt = self.params['frontend.nn_checkpoint']
        if ckpt == 'random':               # benchmark / test mode: seeded random weights, no files
            self.random_init(int(self.params.get('frontend.random_seed', 0)))
            return
        resume_ckpt = join(_share_dir(), ckpt)
        if isfile(resume_ckpt):
            self._log("info", "loading checkpoint '{}'".format(resume_ckpt))
            self.model.load_state_dict(torch.load(resume_ckpt, map_location="cpu"))
        else:
            self._log("error", "Error: Checkpoint path is incorrect {}".format(resume_ckpt))
            raise SystemExit()             # the reference calls exit() here (cosplace.py:68-70)

    def _log(self, level, msg):
        if self.node is not None:
            getattr(self.node.get_logger(), level)(msg)

    def random_init(self, seed=0):
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():
            for m in self.model.backbone.modules():
                if isinstance(m, nn.Conv2d):
                    fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                    if m.bias is not None:
                        m.bias.zero_()
            w = torch.randn((self.descriptor_dim, self.model.features_dim), generator=g) / self.model.features_dim ** 0.5
            self.model.fc_weight = w.to(self.device).contiguous()
            self.model.fc_bias = torch.zeros(self.descriptor_dim, device=self.device)
        self.model.runner = None
        self.model._epoch = object()

    @torch.no_grad()
    def compute_embeddings_device(self, frames_u8, backbone_dtype=None, _runner=None):
        """frames [B,H,W,3] uint8 (device) -> descriptors [B, d] float32 (device)."""
        # the transform writes channels_last storage (what the trunk reads); a normalised 8-bit image is bounded by its
        # normalisation constants: saves the first convolution a pass over it
        x = heads.preprocess(frames_u8.contiguous(), self.crop, channels_last=True)
        return self.model.forward(x, backbone_dtype, _runner, heads.normalised_image_bound())

    def compute_embeddings_batch_device(self, frames_u8, chunk=1000, lanes=2):
        """frames [B,H,W,3] uint8 (device) -> descriptors [B, d] float32 (device), `chunk` frames per pass of the pipeline, the passes
        alternating over `lanes` HIP streams, each with its own trunk runner and workspaces (heads.extract_over_lanes: two passes in
        flight fill each other's tails and launch gaps).  Each chunk runs exactly the kernels of `compute_embeddings_device` on its
        frames, so the descriptors do not depend on `lanes`.  (Since round 5 every kernel of the ResNet runner is this library's: no
        library convolution with per-stream set-up is left.)"""
        if lanes <= 1 or frames_u8.shape[0] <= chunk or self.backbone_conv not in ('winograd', 'winograd2'):
            outs = [self.compute_embeddings_device(frames_u8[s:s + chunk]) for s in range(0, int(frames_u8.shape[0]), chunk)]
            return outs[0] if len(outs) == 1 else torch.cat(outs)

        def lane_runner(i):
            if i == 0:
                if self.model.runner is None:
                    self.model.runner = self.model.make_runner()
                return self.model.runner                               # lane 0 shares the single-pass runner (and its workspaces)
            return self.model.make_runner()
        if self._lanes_epoch is not self.model.runner_epoch():         # the weights were reloaded: folded weights are rebuilt
            self._lanes, self._lanes_epoch = [], self.model.runner_epoch()
        return heads.extract_over_lanes(self._lanes, frames_u8, chunk, lanes, lane_runner,
                                        lambda fr, runner: self.compute_embeddings_device(fr, _runner=runner))

    def compute_embedding(self, keyframe):
        """Global image descriptor of one RGB keyframe (reference cosplace.py:81-105)."""
        if not self.enable:
            return np.random.rand(self.descriptor_dim)
        keyframe = np.ascontiguousarray(keyframe)
        if self.use_graph and keyframe.dtype == np.uint8 and keyframe.ndim == 3:
            if self._online is None or self._online_model is not self.model.runner_epoch():
                # the graph owns its trunk runner: its workspaces must never move under a captured pointer
                self._online_runner = self.model.make_runner() if self.backbone_conv != 'direct' else None
                self._online = heads.OnlineGraph(
                    lambda fr: self.compute_embeddings_device(fr, _runner=self._online_runner), self.device)
                self._online_model = self.model.runner_epoch()
            e = self._online(keyframe)
            if e is not None:
                return e
        frame = torch.from_numpy(keyframe).to(self.device).unsqueeze(0)
        return self.compute_embeddings_device(frame)[0].cpu().numpy()
