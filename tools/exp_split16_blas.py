"""fp16 -> fp32 strided-batched GEMM of the split-fp16 trunk layers through both BLAS back ends torch can use
(TunableOp does not cover bmm with out_dtype in torch 2.10: tools/tune_split16.py tunes nothing)."""
import torch

shapes = [(50176, 256, 256), (12544, 256, 512), (12544, 512, 512), (4096, 512, 512)]
for lib in ("hipblaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    for T, cin, cout in shapes:
        a = torch.randn(36, T, 3 * cin, device="cuda").half()
        b = torch.randn(36, 3 * cin, cout, device="cuda").half()
        try:
            torch.bmm(a, b, out_dtype=torch.float32)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                torch.bmm(a, b, out_dtype=torch.float32)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"{lib}: [36,{T},{3 * cin}] x [36,{3 * cin},{cout}]: {ms:.3f} ms ({2.0 * 36 * T * 3 * cin * cout / ms / 1e9:.0f} TFLOP/s fp16)", flush=True)
        except Exception as e:
            print(f"{lib}: [36,{T},{3 * cin}] x [{cout}]: failed: {str(e)[:120]}", flush=True)
        del a, b
