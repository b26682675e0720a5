import sys, time, torch
sys.path.insert(0, ".")
from cslam_amd.mac.chain_solver_gpu import blocked_cholesky_
for m in (8192, 16384, 32768):
    g = torch.Generator(device="cuda").manual_seed(0)
    B = torch.randn((m, 256), generator=g, device="cuda", dtype=torch.float64)
    A = B @ B.T + torch.eye(m, device="cuda", dtype=torch.float64) * m
    torch.cuda.synchronize(); t0 = time.perf_counter(); L1 = torch.linalg.cholesky(A); torch.cuda.synchronize(); t1 = time.perf_counter()
    for bs in (1024, 2048, 4096):
        A2 = A.clone(); torch.cuda.synchronize(); t2 = time.perf_counter(); L2 = blocked_cholesky_(A2, bs); torch.cuda.synchronize(); t3 = time.perf_counter()
        err = float((torch.tril(L2) - L1).abs().max())
        print(f"m={m}: torch.linalg.cholesky {t1-t0:.2f}s ({m**3/3/(t1-t0)/1e12:.1f} TF)  blocked bs={bs} {t3-t2:.2f}s ({m**3/3/(t3-t2)/1e12:.1f} TF) err {err:.1e}", flush=True)

from cslam_amd.mac.chain_solver_gpu import BlockedCholeskySolve
for m in (8192, 32768):
    g = torch.Generator(device="cuda").manual_seed(0)
    B = torch.randn((m, 256), generator=g, device="cuda", dtype=torch.float64)
    A = B @ B.T + torch.eye(m, device="cuda", dtype=torch.float64) * m
    L = torch.linalg.cholesky(A)
    rhs = torch.randn((m, 4), generator=g, device="cuda", dtype=torch.float64)
    x1 = torch.cholesky_solve(rhs, L); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): x1 = torch.cholesky_solve(rhs, L)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    bsol = BlockedCholeskySolve(L); x2 = bsol.solve(rhs); torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(5): x2 = bsol.solve(rhs)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"m={m} solve 4 rhs: cholesky_solve {(t1-t0)/5*1e3:.1f} ms, blocked {(t3-t2)/5*1e3:.1f} ms, diff {float((x1-x2).abs().max()/x1.abs().max()):.1e}", flush=True)
