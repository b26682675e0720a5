"""oracle/heads_oracle.py -- TEST INFRASTRUCTURE (numpy restatement, float32 like the reference).

CPU restatements of the descriptor heads and the image transform of the reference:
    vlad_forward          NetVLADLayer.forward            cslam/vpr/netvlad.py:94-130
    cosplace_head         L2Norm->GeM->Flatten->Linear->L2Norm
                                        cslam/vpr/cosplace_utils/network.py:23-29, layers.py:8-36
    pca_transform_normalize   pca.transform + sklearn normalize      cslam/vpr/netvlad.py:234-236
    preprocess            CenterCrop->Resize(bicubic)->ToTensor->Normalize  netvlad.py:202-208
Pinned by tests/golden/heads_g.npz (outputs of the reference's own NetVLADLayer / GeM /
L2Norm modules, sklearn's PCA and Pillow, produced by oracle/gen_golden_heads.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math

import numpy as np

F = np.float32


def l2_normalize(x, axis, eps=1e-12):
    """torch.nn.functional.normalize(p=2): x / max(||x||, eps)"""
    n = np.sqrt(np.sum(x * x, axis=axis, keepdims=True, dtype=F)).astype(F)
    return (x / np.maximum(n, F(eps))).astype(F)


def sk_normalize(x):
    """sklearn.preprocessing.normalize (l2): zero-norm rows are left untouched"""
    n = np.sqrt(np.sum(x * x, axis=1, keepdims=True, dtype=x.dtype))
    n[n == 0] = 1
    return x / n


def vlad_forward(x, conv_w, conv_b, centroids):
    """x [N,C,H,W] f32; conv_w [K,C]; conv_b [K] or None; centroids [K,C] -> [N, K*C]"""
    N, C = x.shape[:2]
    K = conv_w.shape[0]
    x = l2_normalize(x.astype(F), 1)                                   # :105-106
    xf = x.reshape(N, C, -1)
    sa = np.einsum("kc,ncp->nkp", conv_w.astype(F), xf).astype(F)      # 1x1 conv, :109
    if conv_b is not None:
        sa = sa + conv_b.astype(F)[None, :, None]
    sa = sa - sa.max(axis=1, keepdims=True)
    e = np.exp(sa).astype(F)
    a = (e / e.sum(axis=1, keepdims=True, dtype=F)).astype(F)          # softmax over clusters, :110
    # sum_p a[n,k,p] (x[n,c,p] - cent[k,c])                              :115-124
    vlad = np.einsum("nkp,ncp->nkc", a, xf).astype(F) - a.sum(axis=2, dtype=F)[:, :, None] * centroids.astype(F)[None]
    vlad = l2_normalize(vlad.astype(F), 2)                              # intra-normalisation :126
    vlad = vlad.reshape(N, -1)
    return l2_normalize(vlad, 1)                                        # :128


def cosplace_head(x, p, eps, W, b):
    """x [N,C,H,W] -> [N,Dout]"""
    N, C = x.shape[:2]
    x = l2_normalize(x.astype(F), 1)
    g = np.power(np.maximum(x, F(eps)), F(p)).reshape(N, C, -1).mean(axis=2, dtype=F)
    g = np.power(g, F(1.0) / F(p)).astype(F)
    y = g @ W.astype(F).T
    if b is not None:
        y = y + b.astype(F)
    return l2_normalize(y.astype(F), 1)


def pca_transform_normalize(x, components, mean, explained_variance=None, whiten=False):
    """sklearn PCA.transform (1.7: X @ C.T - mean @ C.T, / sqrt(var) if whiten) + normalize"""
    y = x @ components.T - (mean.reshape(1, -1) @ components.T)
    if whiten:
        scale = np.sqrt(explained_variance)
        scale[scale < np.finfo(scale.dtype).eps] = np.finfo(scale.dtype).eps
        y = y / scale
    return sk_normalize(y)


# ---- Pillow antialiased bicubic (Resample.c, 8 bits per channel), restated -----------------
def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _coeffs(in_size, out_size):
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, kk = [], np.zeros((out_size, ksize), dtype=np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / fscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds.append((xmin, xmax))
    return bounds, kk


def pil_bicubic_resize_u8(img, out_size):
    """img [h,w,3] uint8 (square) -> [out,out,3] uint8; horizontal pass then vertical pass,
    each rounded (+2^21 >> 22) and clipped to uint8."""
    h, w, _ = img.shape
    bx, kx = _coeffs(w, out_size)
    tmp = np.zeros((h, out_size, 3), dtype=np.uint8)
    src = img.astype(np.int64)
    for xo, (xmin, xn) in enumerate(bx):
        acc = (src[:, xmin:xmin + xn, :] * kx[xo, :xn][None, :, None]).sum(axis=1) + (1 << 21)
        tmp[:, xo, :] = np.clip(acc >> 22, 0, 255)
    by, ky = _coeffs(h, out_size)
    out = np.zeros((out_size, out_size, 3), dtype=np.uint8)
    src = tmp.astype(np.int64)
    for yo, (ymin, yn) in enumerate(by):
        acc = (src[ymin:ymin + yn, :, :] * ky[yo, :yn][:, None, None]).sum(axis=0) + (1 << 21)
        out[yo] = np.clip(acc >> 22, 0, 255)
    return out


def preprocess(img, crop, out_size, mean, std):
    """img [H,W,3] uint8 -> [3,out,out] float32"""
    H, W, _ = img.shape
    if H < crop or W < crop:
        # torchvision CenterCrop (functional.center_crop): zero-pad a frame smaller than the crop, (crop - H) // 2 rows
        # before and (crop - H + 1) // 2 after, then crop the padded frame
        pt, pl = ((crop - H) // 2 if H < crop else 0), ((crop - W) // 2 if W < crop else 0)
        pb, pr = ((crop - H + 1) // 2 if H < crop else 0), ((crop - W + 1) // 2 if W < crop else 0)
        img = np.pad(img, ((pt, pb), (pl, pr), (0, 0)))
        H, W, _ = img.shape
    top, left = int(round((H - crop) / 2.0)), int(round((W - crop) / 2.0))
    c = img[top:top + crop, left:left + crop]
    r = pil_bicubic_resize_u8(c, out_size)
    t = r.astype(F).transpose(2, 0, 1) / F(255)
    return ((t - np.asarray(mean, dtype=F)[:, None, None]) / np.asarray(std, dtype=F)[:, None, None]).astype(F)
