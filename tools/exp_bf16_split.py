"""Numerical feasibility of error-compensated bf16 products for the trunk GEMMs (not built):
a float is split into 3 bf16 parts; bf16 x bf16 products are exact in fp32, the accumulation stays fp32.
Error of C = A B against float64, relative to max|C|, for plain fp32, the 3-product and the 6-product forms.  CPU only."""
import torch

torch.manual_seed(0)


def split3(x):
    h = x.to(torch.bfloat16).float()
    m = (x - h).to(torch.bfloat16).float()
    l = (x - h - m).to(torch.bfloat16).float()
    return h, m, l


def run(M, K, N):
    A, B = torch.randn(M, K), torch.randn(K, N)
    ref = A.double() @ B.double()
    s = ref.abs().max()
    (ah, am, al), (bh, bm, bl) = split3(A), split3(B)
    mm = lambda a, b: a @ b          # fp32 accumulate; the operands are bf16-representable so every product is exact
    c3 = mm(ah, bh) + (mm(ah, bm) + mm(am, bh))
    c6 = c3 + ((mm(ah, bl) + mm(al, bh)) + mm(am, bm))
    e = lambda c: float((c.double() - ref).abs().max() / s)
    print(f"M={M} K={K} N={N}: fp32 {e(A @ B):.2e}   bf16x3 {e(c3):.2e}   bf16x6 {e(c6):.2e}")


for K in (128, 256, 512):
    run(2048, K, 512)


def split2h(x):
    """fp16 two-way split (11 + 11 significant bits); x scaled by a power of two so that max|x| < 2^15 (exact)."""
    sc = 2.0 ** (14 - torch.floor(torch.log2(x.abs().max())))
    xs = x * sc
    h = xs.to(torch.float16).float()
    l = (xs - h).to(torch.float16).float()
    return h, l, sc


def run_h(M, K, N, amp):
    # activations with a wide range (post-ReLU times a Winograd-like amplification), weights O(1/sqrt(K))
    A = torch.relu(torch.randn(M, K)) * torch.exp(2.0 * torch.randn(M, 1)) * amp
    B = torch.randn(K, N) / K ** 0.5
    ref = A.double() @ B.double()
    s = ref.abs().max()
    (ah, al, sa), (bh, bl, sb) = split2h(A), split2h(B)
    c3 = ((ah @ bh) + ((ah @ bl) + (al @ bh))) / (sa * sb)
    e = lambda c: float((c.double() - ref).abs().max() / s)
    print(f"fp16 hi/lo x3, M={M} K={K} N={N} amp={amp:g}: fp32 {e(A @ B):.2e}   fp16x3 {e(c3):.2e}")


for K in (128, 512):
    for amp in (1.0, 100.0):
        run_h(2048, K, 512, amp)
