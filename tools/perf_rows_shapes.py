"""Row-sharded bank (bench.py --shard-mode rows): what ONE rank does per step at N = 1, 2, 4, 8 GPUs, measured on
one GPU -- N*1024 queries against a 100k/N-row shard (same pair work at every N) and the merge of N lists.
    python tools/perf_rows_shapes.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from cslam_amd import nns_matching as nnm
from cslam_amd.sharded import merge_topk_device

d, k, B, total = 4096, 5, 1024, 100_000
gen = torch.Generator(device="cuda").manual_seed(1234)


def best(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)


for N in (1, 2, 4, 8):
    rows, nq = total // N, N * B
    bank = torch.randn((rows, d), generator=gen, device="cuda"); bank /= bank.norm(dim=1, keepdim=True)
    nn = nnm.NearestNeighborsMatching(); nn.add_items_device(bank)
    q = torch.randn((nq, d), generator=gen, device="cuda"); q /= q.norm(dim=1, keepdim=True)
    out = nn.search_device(q, k, mode=nnm.MODE_MFMA)
    t = best(lambda: nn.search_device(q, k, mode=nnm.MODE_MFMA, out=out))
    km = nn.last_kernel_ms()
    lists = (torch.stack([out[0][:B]] * N), torch.stack([out[1][:B]] * N), torch.stack([out[2][:B]] * N))
    offs = [g * rows for g in range(N)]
    tm = best(lambda: merge_topk_device(*lists, offs))
    fl = 2.0 * rows * nq * d
    print(f"N={N}: shard {rows} rows x {nq} queries: search {t*1e3:.2f} ms (mfma kernel {km:.2f} ms = "
          f"{fl/km/1e9:.1f} TFLOP/s), uncertified {nn.last_stats()[0]}; merge of {N} lists x {B} queries {tm*1e6:.0f} us")
    del nn, bank, q
