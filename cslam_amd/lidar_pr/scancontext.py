#!/usr/bin/env python
"""Scan Context descriptor for point clouds on an MI355X.

Drop-in for cslam/lidar_pr/scancontext.py:3-16 (`ScanContext(params, node).compute_embedding`), the
lidar counterpart of the NetVLAD / CosPlace extractors (gdlcd.py:49-54): polar binning of the cloud,
maximum height per bin, 20 rings x 60 sectors out to 80 m.  The binning is hand-written HIP
(csrc/scancontext.hip `sc_from_cloud_kernel` behind `cslam_scancontext_from_cloud_dev`); there is no
CPU path.  Clouds are taken as float64 (float32 clouds are widened exactly first -- the reference's
arithmetic on float32 scalars depends on the numpy version's promotion rules, float64 does not).

Batch extension: `compute_embeddings(list of clouds)` bins all frames in one launch.
"""
import ctypes as C

import numpy as np

from .. import _lib


class ScanContext:
    """
    Scan Context descriptor for point clouds
    From: https://github.com/irapkaist/scancontext
    """

    def __init__(self, params, node, device=0):
        self.node = node
        self.params = params
        self.shape = [20, 60]    # Same as in ScanContext paper
        self.max_length = 80     # Same as in ScanContext paper
        self.device = device
        _lib.require_gpu()
        self._lib = _lib.load()

    def compute_embedding(self, keyframe):
        """Scan context of one point cloud [n, >=3] -> float64 [rings * sectors]."""
        return self.compute_embeddings([keyframe])[0]

    def compute_embeddings(self, keyframes):
        import torch
        clouds = [np.ascontiguousarray(np.asarray(k)[:, :3], dtype=np.float64) for k in keyframes]
        offsets = np.zeros(len(clouds) + 1, dtype=np.int64)
        offsets[1:] = np.cumsum([len(c) for c in clouds])
        dev = torch.device("cuda", self.device)
        pts = torch.from_numpy(np.concatenate(clouds, axis=0) if clouds else np.zeros((0, 3))).to(dev)
        off = torch.from_numpy(offsets).to(dev)
        out = torch.empty((len(clouds), self.shape[0] * self.shape[1]), dtype=torch.float64, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(self._lib.cslam_scancontext_from_cloud_dev(
                pts.data_ptr(), off.data_ptr(), len(clouds), self.shape[0], self.shape[1],
                float(self.max_length), out.data_ptr(), status.data_ptr(), st))
        res = out.cpu().numpy()
        if int(status.item()) != 0:
            raise IndexError("index %d is out of bounds for axis 2 with size %d (a point at exactly 360 degrees, "
                             "as in the reference's ptcloud2sc)" % (self.shape[1], self.shape[1]))
        return res
