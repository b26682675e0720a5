"""NetVLAD global descriptor on an MI355X (reference: cslam/vpr/netvlad.py).

Drop-in class `NetVLAD(params, node)` with `compute_embedding(keyframe) -> np.ndarray`.
Pipeline per frame (reference :212-241): CenterCrop -> bicubic Resize(224) -> ToTensor ->
Normalize [HIP: cslam_preprocess_dev] -> VGG-16 conv5_3 encoder [PyTorch-ROCm] ->
NetVLAD soft-assignment + residual aggregation + intra-norm + L2 [HIP: cslam_vlad_aggregate_dev]
-> PCA projection + L2 [HIP fp32-MFMA GEMM: cslam_pca_project_dev].
`compute_embeddings_device` is the batched, device-resident form used by bench.py and the
batched matcher: frames and descriptors never leave HBM.
"""
import os
import pickle
from os.path import isfile, join

import numpy as np
import torch
from torch import nn

from .. import _lib
from . import heads
from .backbones import vgg16_features_trunk
from .winograd import WinogradTrunk

IMAGENET_DEFAULT_MEAN = heads.IMAGENET_DEFAULT_MEAN
IMAGENET_DEFAULT_STD = heads.IMAGENET_DEFAULT_STD


def _share_dir():
    try:
        from ament_index_python.packages import get_package_share_directory
        return get_package_share_directory("cslam")
    except Exception:
        return ""


def _strip_module(state):
    return {k.replace(".module.", "."): v for k, v in state.items()}


class NetVLADLayer(object):
    """Parameters of the reference's NetVLADLayer (netvlad.py:28-61): 1x1 soft-assignment conv
    (no bias in vladv1) and cluster centroids; forward runs in HIP."""

    def __init__(self, num_clusters=64, dim=512, device="cuda", vladv2=False):
        self.num_clusters, self.dim, self.vladv2, self.alpha = num_clusters, dim, bool(vladv2), 0
        self.conv_weight = torch.zeros((num_clusters, dim), dtype=torch.float32, device=device)
        self.conv_bias = None
        self.centroids = torch.rand((num_clusters, dim), dtype=torch.float32, device=device)

    def load(self, conv_weight, centroids, conv_bias=None):
        dev = self.centroids.device
        self.conv_weight = torch.as_tensor(conv_weight, dtype=torch.float32).reshape(self.num_clusters, self.dim).contiguous().to(dev)
        self.centroids = torch.as_tensor(centroids, dtype=torch.float32).contiguous().to(dev)
        self.conv_bias = None if conv_bias is None else torch.as_tensor(conv_bias, dtype=torch.float32).contiguous().to(dev)

    def init_params(self, clsts, traindescs):
        """Training-time initialisation from k-means clusters `clsts` [K, C] and sampled training descriptors
        `traindescs` [n, C] (reference netvlad.py:63-92; host arithmetic, as there -- it runs once before training
        and is not on the extract path).  vladv1: alpha = -ln(0.01) / mean gap between the two largest
        unit-cluster dots of every descriptor, assignment weight = alpha * clsts/|clsts|, no bias.  vladv2: alpha
        from the two nearest training descriptors of every cluster -- the reference squares what
        `kneighbors(clsts, 2)[1]` returns, i.e. the neighbour INDICES, and that is reproduced --, weight =
        2 alpha * centroids, bias = -alpha * |centroids|."""
        clsts = np.asarray(clsts)
        traindescs = np.asarray(traindescs)
        assert clsts.shape == (self.num_clusters, self.dim) and traindescs.ndim == 2 and traindescs.shape[1] == self.dim
        if not self.vladv2:
            assign = clsts / np.linalg.norm(clsts, axis=1, keepdims=True)
            dots = np.dot(assign, traindescs.T)
            dots.sort(0)
            dots = dots[::-1, :]
            self.alpha = (-np.log(0.01) / np.mean(dots[0, :] - dots[1, :])).item()
            self.load(self.alpha * assign, clsts, None)
        else:
            c64, t64 = clsts.astype(np.float64), traindescs.astype(np.float64)
            d2 = (c64 * c64).sum(1)[:, None] - 2.0 * c64 @ t64.T + (t64 * t64).sum(1)[None, :]
            two = np.argsort(d2, axis=1, kind="stable")[:, :2]                 # 2 nearest descriptors, nearest first
            ds_sq = np.square(two)
            self.alpha = (-np.log(0.01) / np.mean(ds_sq[:, 1] - ds_sq[:, 0])).item()
            cent = torch.as_tensor(clsts, dtype=torch.float32)
            self.load((2.0 * self.alpha * cent).numpy(), clsts, (-self.alpha * cent.norm(dim=1)).numpy())
        return self

    def forward(self, x):
        return heads.vlad_aggregate(x, self.conv_weight, self.conv_bias, self.centroids)

    __call__ = forward


class NetVLAD(object):
    """NetVLAD matcher"""

    def __init__(self, params, node):
        self.params = params
        self.node = node
        self.enable = self.params['frontend.nn_checkpoint'].lower() != 'disable'
        if not self.enable:
            return
        _lib.require_gpu()
        if not torch.cuda.is_available():
            raise _lib.CslamHipError("NetVLAD needs PyTorch-ROCm with a visible MI355X")
        self.device = torch.device("cuda")
        self.crop = int(self.params["frontend.image_crop_size"])
        # channels_last: MIOpen's NHWC fp32 igemm kernels are ~6 % faster than NCHW on gfx950 (measured)
        self.encoder = vgg16_features_trunk().to(self.device).eval().to(memory_format=torch.channels_last)
        # How the 3x3 convolutions with >= 64 input channels are executed (vpr/winograd.py):
        #   'winograd'  (default) F(4x4,3x3) on maps whose sides are multiples of 4, F(2x2,3x3) on the others
        #   'winograd2' F(2x2,3x3) only;  'direct' every layer through torch / MIOpen.
        # Measured on VGG-16 at B = 256: 7.9k / 5.4k / 3.6k frames/s; max error against a float64 trunk
        # 3.7e-6 / 1.3e-6 / 1.4e-6 of the largest activation.
        self.backbone_conv = str(self.params.get('frontend.backbone_conv', 'winograd')).lower()
        if self.backbone_conv not in ('winograd', 'winograd2', 'direct'):
            raise ValueError("frontend.backbone_conv must be 'winograd', 'winograd2' or 'direct'")
        self.trunk = None
        # frontend.hip_graph: true replays one-keyframe calls from a captured HIP graph of the whole pipeline.
        # Off by default: on ROCm 7.2 replaying the ~45-node graph takes 6.2 ms against 1.2 ms for launching the
        # same kernels one by one (tools/perf_online.py), so plain launches are the faster online path today.
        self.use_graph = bool(self.params.get('frontend.hip_graph', False))
        # frontend.trunk_forms: which kernel form the layers of the VGG trunk take (vpr/winograd.py TRUNK_FORMS; None = the defaults;
        # winograd.FP32_GEMM_FORMS = plain fp32 library GEMMs, what bench.py prints as `value_fp32_gemms`)
        self.trunk_forms = self.params.get('frontend.trunk_forms')
        self._online = None
        self._online_trunk = None
        self._lanes = []                       # (stream, trunk) per extraction lane of compute_embeddings_batch_device
        self.pool = NetVLADLayer(num_clusters=64, dim=512, device=self.device)
        self.pca_components = None     # [Dout, Din] device
        self.pca_pairs = None          # the same as exact fp16 hi / lo pairs (heads.pca_pair_weights): batches
        self.pca_mean_proj = None      # [Dout] = mean @ components.T
        self.pca_inv_scale = None
        for p in self.encoder.parameters():
            p.requires_grad_(False)

        ckpt = self.params['frontend.nn_checkpoint']
        if ckpt == 'random':               # benchmark / test mode: seeded random weights, no files
            self.random_init(int(self.params.get('frontend.random_seed', 0)),
                             int(self.params.get('frontend.netvlad.pca_dim', 4096)))
            return
        pkg_folder = _share_dir()
        resume_ckpt = join(pkg_folder, ckpt)
        pca_name = self.params.get('frontend.netvlad.pca_checkpoint')
        if pca_name is None and node is not None:
            pca_name = node.get_parameter('frontend.netvlad.pca_checkpoint').value
        if isfile(resume_ckpt):
            checkpoint = torch.load(resume_ckpt, map_location="cpu")
            self.load_state_dict(checkpoint['state_dict'])
        else:
            print("Error: Checkpoint path is incorrect")     # reference prints and continues (:198-199)
        pca = pickle.load(open(join(pkg_folder, pca_name), 'rb'))
        self.set_pca(pca.components_, pca.mean_, getattr(pca, "explained_variance_", None),
                     bool(getattr(pca, "whiten", False)))

    # ------------------------------------------------------------------ weights ----
    def load_state_dict(self, state):
        """Reference checkpoint layout: 'encoder.<idx>.weight|bias', 'pool.conv.weight',
        'pool.centroids' (optionally with DataParallel's '.module.')."""
        state = _strip_module(state)
        enc = {k[len("encoder."):]: v for k, v in state.items() if k.startswith("encoder.")}
        self.encoder.load_state_dict(enc)
        self.trunk = None                      # transformed weights are rebuilt on the next forward
        self._online = self._online_trunk = None
        self._lanes = []
        self.pool.load(state["pool.conv.weight"], state["pool.centroids"], state.get("pool.conv.bias"))

    def set_pca(self, components, mean, explained_variance=None, whiten=False):
        comp = np.ascontiguousarray(components, dtype=np.float32)
        mean = np.asarray(mean, dtype=np.float32)
        self.pca_components = heads.padded_rows(comp.shape[0], comp.shape[1], self.device)
        self.pca_components.copy_(torch.from_numpy(comp))
        self.pca_pairs = heads.pca_pair_weights(self.pca_components)
        self.pca_mean_proj = torch.from_numpy((mean.reshape(1, -1) @ comp.T).reshape(-1).astype(np.float32)).to(self.device)
        self.pca_inv_scale = None
        if whiten:
            scale = np.sqrt(np.asarray(explained_variance, dtype=np.float32))
            scale[scale < np.finfo(np.float32).eps] = np.finfo(np.float32).eps
            self.pca_inv_scale = torch.from_numpy((1.0 / scale).astype(np.float32)).to(self.device)

    def random_init(self, seed=0, pca_dim=4096):
        """Seeded random weights of the reference architecture (no checkpoint ships with cslam:
        models/.gitignore); used for throughput runs and structural tests."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():
            for m in self.encoder.modules():
                if isinstance(m, nn.Conv2d):
                    fan_in = m.in_channels * 9
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                    m.bias.zero_()
        self.trunk = None
        self._online = self._online_trunk = None
        self._lanes = []
        # centroids on the scale of the L2-normalised local descriptors they cluster and an assignment tied to them
        # (the relation of the reference's init_params, netvlad.py:63-92: conv weight = 2 * alpha * centroid), so
        # that the random-weight descriptors actually depend on the image (uniform centroids of norm ~13 swamp the
        # unit-norm features: every residual is ~ -centroid and all descriptors coincide)
        cent = torch.randn((64, 512), generator=g)
        cent = cent / cent.norm(dim=1, keepdim=True) * 0.5
        w = 2.0 * 10.0 * cent
        self.pool.load(w, cent)
        din = 64 * 512
        comp = torch.randn((pca_dim, din), generator=g, dtype=torch.float32) / din ** 0.5
        self.pca_components = heads.padded_rows(pca_dim, din, self.device)
        self.pca_components.copy_(comp)
        self.pca_pairs = heads.pca_pair_weights(self.pca_components)
        self.pca_mean_proj = torch.zeros(pca_dim, dtype=torch.float32, device=self.device)
        self.pca_inv_scale = None

    # ------------------------------------------------------------------ forward ----
    @torch.no_grad()
    def compute_embeddings_device(self, frames_u8, backbone_dtype=None, _trunk=None):
        """frames [B,H,W,3] uint8 (device) -> descriptors [B, d] float32 (device)."""
        x = heads.preprocess(frames_u8.contiguous(), self.crop)
        if self.backbone_conv == 'direct' or (backbone_dtype is not None and backbone_dtype != torch.float32):
            x = x.contiguous(memory_format=torch.channels_last)
        if backbone_dtype is not None and backbone_dtype != torch.float32:
            with torch.autocast("cuda", dtype=backbone_dtype):
                f = self.encoder(x)
            f = f.float()
        elif self.backbone_conv in ('winograd', 'winograd2'):
            if _trunk is not None:
                f = _trunk(x)
            else:
                if self.trunk is None:
                    self.trunk = WinogradTrunk(self.encoder, min_in_channels=64,
                                               tile=4 if self.backbone_conv == 'winograd' else 2, forms=self.trunk_forms)
                    # a normalised 8-bit image is bounded by its normalisation constants: saves the trunk a pass over it
                    self.trunk.input_bound = heads.normalised_image_bound()
                f = self.trunk(x)
        else:
            f = self.encoder(x)
        v = self.pool(f)
        # the VLAD vector is L2-normalised (netvlad.py:130): |v| <= 1 (one ulp of slack) spares the projection its max-pass
        return heads.pca_project(v, self.pca_components, self.pca_mean_proj, self.pca_inv_scale, self.pca_pairs, 1.0 + 2.0 ** -20)

    def compute_embeddings_batch_device(self, frames_u8, chunk=512, lanes=2):
        """frames [B,H,W,3] uint8 (device) -> descriptors [B, d] float32 (device), `chunk` frames per pass of the pipeline, the
        passes alternating over `lanes` HIP streams, each with its own trunk workspaces (heads.extract_over_lanes).  Each chunk
        runs exactly the kernels of `compute_embeddings_device` on its frames, so the descriptors do not depend on `lanes`."""
        if lanes <= 1 or frames_u8.shape[0] <= chunk or self.backbone_conv not in ('winograd', 'winograd2'):
            outs = [self.compute_embeddings_device(frames_u8[s:s + chunk]) for s in range(0, int(frames_u8.shape[0]), chunk)]
            return outs[0] if len(outs) == 1 else torch.cat(outs)

        def lane_trunk(i):
            if i == 0 and self.trunk is not None:
                return self.trunk                                   # lane 0 shares the single-pass trunk (and its workspaces)
            trunk = WinogradTrunk(self.encoder, min_in_channels=64, tile=4 if self.backbone_conv == 'winograd' else 2, forms=self.trunk_forms)
            trunk.input_bound = heads.normalised_image_bound()
            if i == 0:
                self.trunk = trunk
            return trunk
        return heads.extract_over_lanes(self._lanes, frames_u8, chunk, lanes, lane_trunk,
                                        lambda fr, trunk: self.compute_embeddings_device(fr, _trunk=trunk))

    def compute_embedding(self, keyframe):
        """Global image descriptor of one RGB keyframe (reference :212-245)."""
        if not self.enable:
            return np.random.rand(128)
        keyframe = np.ascontiguousarray(keyframe)
        if self.use_graph and keyframe.dtype == np.uint8 and keyframe.ndim == 3:
            if self._online is None:
                # the graph owns its trunk runner: its V / M workspaces must never move under a captured pointer
                if self.backbone_conv != 'direct':
                    self._online_trunk = WinogradTrunk(self.encoder, min_in_channels=64, tile=2)
                self._online = heads.OnlineGraph(
                    lambda fr: self.compute_embeddings_device(fr, _trunk=self._online_trunk), self.device)
            e = self._online(keyframe)
            if e is not None:
                return e
        frame = torch.from_numpy(keyframe).to(self.device).unsqueeze(0)
        return self.compute_embeddings_device(frame)[0].cpu().numpy()
