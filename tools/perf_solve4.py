"""Time of cslam_chol_solve4_dev (the junction solve of one TraceMIN iteration) on a row-major and a column-major factor."""
import sys, time, torch
sys.path.insert(0, ".")
from cslam_amd.mac.chain_solver_gpu import BlockedCholeskySolve, blocked_cholesky_
for m in (8192, 32768):
    g = torch.Generator(device="cuda").manual_seed(0)
    B = torch.randn((m, 256), generator=g, device="cuda", dtype=torch.float64)
    A = B @ B.T + torch.eye(m, device="cuda", dtype=torch.float64) * m
    rhs = torch.randn((m, 4), generator=g, device="cuda", dtype=torch.float64)
    L_row = blocked_cholesky_(A.clone())
    L_col = torch.linalg.cholesky(A)
    ref = torch.cholesky_solve(rhs, L_col)
    for name, L, bs in (("row-major", L_row, 2048), ("column-major", L_col, 2048), ("row-major bs=1024", L_row, 1024), ("row-major bs=512", L_row, 512)):
        s = BlockedCholeskySolve(L, bs)
        x = s.solve(rhs); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): x = s.solve(rhs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        gb = 2 * (m * m / 2) * 8 / 1e9
        print(f"m={m} {name}: {dt*1e3:.3f} ms per solve ({gb/dt/1e3:.2f} TB/s of the factor's triangle twice), diff {float((x-ref).abs().max()/ref.abs().max()):.1e}", flush=True)
