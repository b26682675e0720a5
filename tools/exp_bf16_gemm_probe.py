"""Throughput probe for DESIGN.md section 8's next-round item: what the library bf16 MFMA path reaches on the trunk's
36-frequency batched GEMM shapes when K is 6x (six error-compensated products) or 3x, beside today's fp32 GEMM.
bf16 output here (probe of the MFMA rate only; the real kernel accumulates and writes fp32)."""
import torch

dev = "cuda"
shapes = [("conv2_2", 200704, 128, 128), ("conv3_2", 50176, 256, 256), ("conv4_2", 12544, 512, 512), ("conv5_x", 4096, 512, 512)]


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, T, K, N in shapes:
    a32, b32 = torch.randn(36, T, K, device=dev), torch.randn(36, K, N, device=dev)
    ms32 = t(lambda: torch.bmm(a32, b32))
    flop = 2.0 * 36 * T * K * N
    line = f"{name}: tiles {T} K {K} N {N}: fp32 {ms32:.3f} ms ({flop / ms32 / 1e9:.0f} TFLOP/s)"
    del a32, b32
    for mult in (3, 6):
        a = torch.randn(36, T, mult * K, device=dev).bfloat16()
        b = torch.randn(36, mult * K, N, device=dev).bfloat16()
        ms = t(lambda: torch.bmm(a, b))
        line += f" | bf16 K x{mult} {ms:.3f} ms ({flop * mult / ms / 1e9:.0f} TFLOP/s bf16, {flop / ms / 1e9:.0f} fp32-equivalent)"
        del a, b
    print(line, flush=True)
