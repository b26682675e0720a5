import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from cslam_amd.nns_matching import NearestNeighborsMatching
rng = np.random.default_rng(0)
bank = rng.standard_normal((100000, 4096)).astype(np.float32)
bank /= np.linalg.norm(bank, axis=1, keepdims=True)
m = NearestNeighborsMatching(); m.add_items(bank, range(100000))
q = bank[123] + 0.01 * rng.standard_normal(4096).astype(np.float32)
def T(fn, n=30):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
for name, fn in [("search f32 k5", lambda: m.search(q, 5)), ("search f64 k5", lambda: m.search(q.astype(np.float64), 5)),
                 ("search f32 k1", lambda: m.search(q, 1)), ("search f64 k1", lambda: m.search(q.astype(np.float64), 1)),
                 ("search f32 k10", lambda: m.search(q, 10)),
                 ("batch f32 k5", lambda: m.search_batch(q[None], 5)), ("search f32 k5 again", lambda: m.search(q, 5))]:
    print(name, f"{T(fn):.0f} us", "kernel ms", m.last_kernel_ms())
