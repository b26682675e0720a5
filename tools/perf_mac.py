"""Fiedler-pair timing at pose-graph scale (GPU box): python tools/perf_mac.py [poses_per_robot] [loop_edges]"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_mac_gpu import _pose_graph
from cslam_amd.mac.chain_solver_gpu import fiedler_tracemin_chain_gpu
from cslam_amd.mac.fiedler import fiedler_tracemin_lu

P = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
L = _pose_graph(8, P, m, 1)
n = L.shape[0]
fiedler_tracemin_chain_gpu(_pose_graph(2, 500, 10, 0))     # warm up libraries
st = {'t0': time.perf_counter()}
t0 = time.perf_counter(); l2, v2 = fiedler_tracemin_chain_gpu(L, stats=st); t1 = time.perf_counter()
print('   breakdown:', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if k != 't0'})
res = np.linalg.norm(L @ v2 - l2 * v2, 1) / abs(L).sum(axis=1).max()
print(f"n={n} loop_edges={m}: chain_gpu lambda2={l2:.6e} residual={res:.2e} time {t1-t0:.2f}s", flush=True)
if n <= 200000 or (len(sys.argv) > 3 and sys.argv[3] == "ref"):
    t0 = time.perf_counter(); l1, v1 = fiedler_tracemin_lu(L); t1 = time.perf_counter()
    print(f"   reference algorithm (SuperLU TraceMIN, host): lambda2={l1:.6e} time {t1-t0:.2f}s  rel diff {abs(l1-l2)/l1:.1e}")
