#!/usr/bin/env python
"""Steady-state per-kernel summary of a rocprofv3 --kernel-trace CSV of tools/extract_leg.py: keeps the last
`iters` passes (cut at `preprocess_fused_kernel`), so MIOpen's find-mode kernels of the warm-up are excluded.
With --pmc <counter> the CSV is a `--pmc` pass and the counter is averaged per launch instead of the time.

    python tools/kernel_trace_summary.py <kernel_trace.csv | counter_collection.csv> [--iters 4] [--frames 256] [--pmc NAME]
"""
import argparse
import collections
import csv


def short(n):
    if "Cijk" in n:
        i = n.find("_MT")
        return "rocBLAS sgemm " + (n[i + 1:i + 14] if i >= 0 else "")
    if n.startswith("_ZN2ck"):
        return "MIOpen/CK grouped_conv_fwd"
    if n.startswith("void "):
        n = n[5:]
    return n.split("(")[0][:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--pmc", default=None)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    if a.pmc:
        # counter_collection.csv: one row per (dispatch, counter); dispatch order = Dispatch_Id
        byd = collections.OrderedDict()
        for r in rows:
            if r["Counter_Name"] != a.pmc:
                continue
            d = byd.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])
            d[1] += float(r["Counter_Value"])
        seq = [byd[k] for k in sorted(byd)]
        idx = [i for i, (n, _) in enumerate(seq) if ("preprocess_fused" in n or "preprocess_tile" in n)]
        sel = seq[idx[-a.iters]:]
        agg = collections.defaultdict(lambda: [0.0, 0])
        for n, v in sel:
            agg[short(n)][0] += v
            agg[short(n)][1] += 1
        print(f"{a.pmc}: per-pass totals over the last {a.iters} passes of {a.frames} frames")
        for k, (v, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
            print(f"  {v / a.iters:16.0f} per pass  x{c / a.iters:5.1f}  {k}")
        return
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if ("preprocess_fused" in r["Kernel_Name"] or "preprocess_tile" in r["Kernel_Name"])]
    sel = rows[idx[-a.iters]:]
    t0 = int(sel[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in sel)
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        k = short(r["Kernel_Name"])
        agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[k][1] += 1
    tot = sum(v[0] for v in agg.values())
    print(f"last {a.iters} passes x {a.frames} frames: wall {(t1 - t0) / 1e6:.1f} ms, kernel time {tot / 1e6:.1f} ms "
          f"-> {a.iters * a.frames / ((t1 - t0) / 1e9):.0f} frames/s")
    print("   ms/pass      %  launches/pass  kernel")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:20]:
        print(f"{v[0] / 1e6 / a.iters:10.3f} {100 * v[0] / tot:6.1f} {v[1] / a.iters:10.1f}     {k}")


if __name__ == "__main__":
    main()
