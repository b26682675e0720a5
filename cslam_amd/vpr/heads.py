"""torch-tensor front ends of the HIP descriptor heads (C ABI: include/cslam_hip.h).

PyTorch is plumbing here (device memory + the current stream); the arithmetic is in
cslam_amd/csrc/heads.hip and gemm_nt.hip.  Every function requires CUDA(ROCm) tensors.
"""
import ctypes as C

import torch

from .. import _lib

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _chk(t, dtype=torch.float32):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise _lib.CslamHipError("HIP heads need contiguous device tensors of dtype %s" % dtype)


def preprocess(frames_u8, crop, out_hw=224, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD, channels_last=False):
    """[B,H,W,3] uint8 RGB -> [B,3,out_hw,out_hw] float32 (netvlad.py:202-208); channels_last: the same tensor in channels_last
    storage, written so by the kernel (no transposing copy in front of a trunk that reads NHWC)."""
    _chk(frames_u8, torch.uint8)
    B, H, W, _ = frames_u8.shape
    out = torch.empty((B, 3, out_hw, out_hw), dtype=torch.float32, device=frames_u8.device,
                      memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    lib = _lib.load()
    _lib.check((lib.cslam_preprocess_nhwc_dev if channels_last else lib.cslam_preprocess_dev)(_p(frames_u8), B, H, W, int(crop), int(out_hw), C.byref(m),
                                                C.byref(s), _p(out), _stream(out)))
    return out


def normalised_image_bound(mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD):
    """max |(v - mean) / std| over v in [0, 1] and the three channels: a rigorous bound of |preprocess(...)| (float32
    arithmetic of the kernel included: one ulp of slack)."""
    b = max(max(abs(0.0 - m) / s, abs(1.0 - m) / s) for m, s in zip(mean, std))
    return float(b) * (1.0 + 2.0 ** -20)


def padded_rows(n, d, device, dtype=torch.float32):
    """[n, d] view whose row pitch avoids multiples of 256 floats (power-of-two pitches put the same
    column of every row into the same L2 set; see csrc/bank.hip)."""
    pitch = d + 32 if d % 256 == 0 else d
    return torch.empty((n, pitch), dtype=dtype, device=device)[:, :d]


def vlad_aggregate(feat, assign_w, assign_b, centroids, out=None):
    """[B,C,h,w] -> [B, 64*C]  NetVLADLayer.forward (netvlad.py:94-130).  A channels_last feature map (what the Winograd trunk
    writes) is used in place by the batch kernel (B > 8, h w <= 256); otherwise the map must be contiguous NCHW."""
    _chk(assign_w); _chk(centroids)
    B, Cc = feat.shape[:2]
    P = feat.shape[2] * feat.shape[3]
    K = centroids.shape[0]
    if out is None:
        out = padded_rows(B, K * Cc, feat.device)
    assert out.shape == (B, K * Cc) and out.stride(1) == 1
    if (B > 8 and P <= 256 and feat.is_cuda and feat.dtype == torch.float32 and not feat.is_contiguous()
            and feat.is_contiguous(memory_format=torch.channels_last)):
        _lib.check(_lib.load().cslam_vlad_aggregate_nhwc_dev(
            _p(feat), _p(assign_w), _p(assign_b) if assign_b is not None else None, _p(centroids), B, Cc, P, K, _p(out),
            out.stride(0), _stream(out)))
        return out
    feat = feat.contiguous()
    _chk(feat)
    _lib.check(_lib.load().cslam_vlad_aggregate_dev(_p(feat), _p(assign_w), _p(assign_b) if assign_b is not None else None,
                                                    _p(centroids), B, Cc, P, K, _p(out), out.stride(0), _stream(out)))
    return out


def gem_fc_head(feat, p, eps, W, b):
    """[B,C,h,w] -> [B,Dout]  L2Norm->GeM->Flatten->Linear->L2Norm (network.py:23-29).  A channels_last map (what the trunks of
    vpr/winograd.py write) is read in place; otherwise the map must be contiguous NCHW."""
    _chk(W)
    B, Cc = feat.shape[:2]
    P = feat.shape[2] * feat.shape[3]
    out = torch.empty((B, W.shape[0]), dtype=torch.float32, device=feat.device)
    if (feat.is_cuda and feat.dtype == torch.float32 and Cc % 4 == 0 and not feat.is_contiguous()
            and feat.is_contiguous(memory_format=torch.channels_last)):
        _lib.check(_lib.load().cslam_gem_fc_head_nhwc_dev(_p(feat), float(p), float(eps), _p(W),
                                                          _p(b) if b is not None else None, B, Cc, P, W.shape[0],
                                                          _p(out), _stream(out)))
        return out
    feat = feat.contiguous()
    _chk(feat)
    _lib.check(_lib.load().cslam_gem_fc_head_dev(_p(feat), float(p), float(eps), _p(W),
                                                 _p(b) if b is not None else None, B, Cc, P, W.shape[0],
                                                 _p(out), _stream(out)))
    return out


PCA_PAIR_SPLITS = 8             # K splits of the pair form of the projection (the "frequencies" of the pair GEMM): 8 vs 16 within 5 %
PCA_PAIR_MIN_BATCH = 32         # below this the f32 forms (matrix-vector / f32-MFMA tile) are used


def pca_pair_weights(components, splits=PCA_PAIR_SPLITS):
    """components [Dout, Din] float32 -> (W2 [S, Dout, Din/S/32, 2, 32] float16, inv_sw) for `cslam_pca_project_pairs_dev`:
    sW * components split into exact fp16 hi / lo pairs (sW the power of two that brings max |W| into [2^14, 2^15)), rows
    = output components, every 32-wide block of a row's K split holds its hi halves then its lo halves.  None when the
    shape does not fit (Din a multiple of 32 S, Dout of 128)."""
    import math
    dout, din = components.shape
    if din % (32 * splits) or dout % 128:
        return None
    w = components.detach().to(torch.float32)
    amax = float(w.abs().max())
    sw = 2.0 ** (14 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    ws = w * sw
    wh = ws.to(torch.float16)
    wl = (ws - wh.to(torch.float32)).to(torch.float16)
    ks = din // splits
    pair = torch.stack((wh.view(dout, splits, ks // 32, 32), wl.view(dout, splits, ks // 32, 32)), dim=3)   # [Dout,S,kb,2,32]
    return pair.permute(1, 0, 2, 3, 4).contiguous(), 1.0 / sw


def pca_project(x, components, mean_proj, inv_scale, pairs=None, x_bound=0.0):
    """[B,Din] -> normalised [B,Dout] (netvlad.py:234-236); x / components may be pitched views.  pairs =
    `pca_pair_weights(components)`: batches run on the fp16 matrix pipe from exact hi / lo pairs (fp32-grade); x_bound > 0 =
    a known bound of max |x| (1 for L2-normalised VLAD vectors)."""
    if pairs is not None and x.shape[0] >= PCA_PAIR_MIN_BATCH and x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1:
        out = torch.empty((x.shape[0], components.shape[0]), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().cslam_pca_project_pairs_dev(_p(x), x.stride(0), float(x_bound), _p(pairs[0]), float(pairs[1]), pairs[0].shape[0],
                                                           _p(mean_proj) if mean_proj is not None else None,
                                                           _p(inv_scale) if inv_scale is not None else None,
                                                           x.shape[0], x.shape[1], components.shape[0], _p(out), _stream(out)))
        return out
    for t in (x, components):
        if not (t.is_cuda and t.dtype == torch.float32 and t.stride(1) == 1):
            raise _lib.CslamHipError("pca_project needs float32 device tensors with unit column stride")
    out = torch.empty((x.shape[0], components.shape[0]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().cslam_pca_project_dev(_p(x), x.stride(0), _p(components), components.stride(0),
                                                 _p(mean_proj) if mean_proj is not None else None,
                                                 _p(inv_scale) if inv_scale is not None else None,
                                                 x.shape[0], x.shape[1], components.shape[0], _p(out), _stream(out)))
    return out


def l2_normalize_(x, eps=1e-12, zero_norm_to_one=False):
    """In-place row normalisation of a [n,d] tensor."""
    _chk(x)
    _lib.check(_lib.load().cslam_l2_normalize_dev(_p(x), x.shape[0], x.shape[1], x.stride(0), float(eps),
                                                  int(zero_norm_to_one), _stream(x)))
    return x


def extract_over_lanes(lane_list, frames_u8, chunk, lanes, lane_runner, run_chunk):
    """The batch form of an extractor: `chunk` frames per pass of its pipeline, the passes alternating over `lanes` HIP streams.
    `lane_list` is the extractor's own list of (stream, runner) pairs, grown here with `lane_runner(i)` (a trunk runner with its
    own V / M workspaces); `run_chunk(frames, runner)` -> [b, d] runs one pass on the CURRENT stream.  Two passes in flight fill
    each other's tails and launch gaps (NetVLAD: 16.5 -> 15.3 ms per 256 frames with two lanes, no gain from a third, none from
    offsetting the lanes by part of a pass: profiles/r03_v25_two_lanes.log).  The lanes start behind the caller's stream and the
    caller's stream continues behind them: no host synchronisation.  (CosPlace: offered since round 5 -- while its ResNet runner
    launched library convolutions, 7 x 7 stem and strided layers, their per-stream set-up made a second lane 16 x slower,
    profiles/r03_v30_c2_lanes_rejected.log.)"""
    B = int(frames_u8.shape[0])
    starts = list(range(0, B, chunk))
    lanes = min(lanes, len(starts))
    while len(lane_list) < lanes:
        lane_list.append((torch.cuda.Stream(device=frames_u8.device), lane_runner(len(lane_list))))
    cur = torch.cuda.current_stream(frames_u8.device)
    out = None
    for st, _ in lane_list[:lanes]:
        st.wait_stream(cur)
    for i, s in enumerate(starts):
        st, runner = lane_list[i % lanes]
        with torch.cuda.stream(st):
            d = run_chunk(frames_u8[s:s + chunk], runner)
            if out is None:
                # The shared result lives on the CALLER's stream: allocated inside a lane's context the caching allocator would
                # carve it from a block that lane has just freed (chunk 0's intermediates, whose kernels are still queued),
                # and the other lanes' copies into it are not ordered against that lane's queue.  A block of the caller's
                # stream was last used by work the lanes already wait for (the fork above); every lane that writes it is
                # recorded, so it is not handed out again before the lanes are done with it.
                with torch.cuda.stream(cur):
                    out = torch.empty((B, d.shape[1]), dtype=d.dtype, device=d.device)
                for ls, _ in lane_list[:lanes]:
                    out.record_stream(ls)
            out[s:s + d.shape[0]].copy_(d)
    for st, _ in lane_list[:lanes]:
        cur.wait_stream(st)
    return out


class OnlineGraph(object):
    """The one-keyframe pipeline (H2D of the frame excluded) captured once per frame shape in a HIP graph and
    replayed: the online path is ~45 small launches per keyframe, launch-bound when issued one by one.
    `fn(frames_u8 [1,H,W,3] device) -> [1,d] device` must only launch on the current stream and keep every
    buffer it touches alive and in place (dedicated workspaces; the C side never frees what a graph may hold).
    Any capture failure disables the graph for good and the caller falls back to plain launches of the same
    kernels."""

    def __init__(self, fn, device):
        self.fn, self.device = fn, device
        self.entries = {}
        self.failed = False

    def _capture(self, shape):
        inp = torch.empty((1,) + tuple(shape), dtype=torch.uint8, device=self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):                       # lazy initialisation, workspace growth, library handles
                self.fn(inp)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        # captured on the stream the warm-up ran on: the C side keeps its scratch per (device, stream) and cannot allocate
        # while a stream is capturing
        with torch.cuda.graph(graph, stream=side):
            out = self.fn(inp)
        pin_in = torch.empty(tuple(shape), dtype=torch.uint8).pin_memory()
        pin_out = torch.empty(tuple(out.shape), dtype=out.dtype).pin_memory()
        return graph, inp, out, pin_in, pin_out

    def __call__(self, frame):
        """frame: numpy uint8 [H,W,3] -> numpy [d], or None when graphs are unavailable."""
        if self.failed:
            return None
        key = tuple(frame.shape)
        try:
            ent = self.entries.get(key)
            if ent is None:
                ent = self.entries[key] = self._capture(key)
            graph, inp, out, pin_in, pin_out = ent
            pin_in.copy_(torch.from_numpy(frame))
            inp[0].copy_(pin_in, non_blocking=True)
            graph.replay()
            pin_out.copy_(out, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            return pin_out[0].numpy().copy()
        except Exception as e:                        # noqa: BLE001 - capture support varies; same kernels either way
            import warnings
            warnings.warn("cslam_amd: HIP graph capture of the online path failed (%s); using plain launches" % e)
            self.failed = True
            self.entries.clear()
            return None
