"""float64 GEMM rate of the trailing update of the blocked junction Cholesky by operand form (GPU box):
C[M,N] -= op(A) op(B), K = 2048, through torch (rocBLAS / hipBLASLt underneath)."""
import sys, time
import torch
sys.path.insert(0, ".")
dev = "cuda"
K = 2048
for (M, N) in ((4096, 28000), (4096, 12000), (2048, 28000)):
    Akm = torch.randn(M, K, device=dev, dtype=torch.float64)      # row-major [M][K]  (k contiguous)
    Bkm = torch.randn(N, K, device=dev, dtype=torch.float64)      # row-major [N][K]
    Amk = Akm.T.contiguous()                                       # row-major [K][M]  (m contiguous)
    Bnk = Bkm.T.contiguous()                                       # row-major [K][N]
    C = torch.zeros(M, N, device=dev, dtype=torch.float64)
    forms = {
        "A[M][K] B[N][K]^T (both k-contiguous)": lambda: C.addmm_(Akm, Bkm.T, alpha=-1.0),
        "A[K][M]^T B[K][N] (both k-strided)": lambda: C.addmm_(Amk.T, Bnk, alpha=-1.0),
        "A[M][K] B[K][N]": lambda: C.addmm_(Akm, Bnk, alpha=-1.0),
        "A[K][M]^T B[N][K]^T": lambda: C.addmm_(Amk.T, Bkm.T, alpha=-1.0),
    }
    for name, f in forms.items():
        for _ in range(2): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"M={M} N={N} K={K} {name}: {dt*1e3:.2f} ms = {2.0*M*N*K/dt/1e12:.1f} TFLOP/s", flush=True)
