#!/usr/bin/env python
"""Board power and shader clock of GPU 0 while a command runs: is the chip at its power cap during the timed steps?

DESIGN.md sections 3.1 / 3.6 argue that the fp16-pair kernels are bound by the chip's power budget (the matrix pipe busy 0.6 of its
cycles at 1.5-2.0 GHz instead of 2.4): this script puts the board's own sensors beside that argument.  It samples the amdgpu hwmon
files of card 0 (power1_average or power1_input in microwatts, power1_cap, freq1_input = sclk in Hz) every `period` seconds from a
thread while the command runs as a child process, and prints one JSON object: samples, power mean / p50 / p95 / max, the cap, sclk
mean / min / max, each over the samples whose power is above half of the maximum (the busy part of the run) and over all samples.
Falls back to `rocm-smi --showpower --showclocks --json` (about 1 Hz) when the hwmon files are not there.

    python tools/power_trace.py [--period 0.02] -- python bench.py --steps 60 --warmup 5"""
import glob
import json
import os
import statistics
import subprocess
import sys
import threading
import time


def all_hwmon():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        files = {f: os.path.join(d, f) for f in ("power1_average", "power1_input", "power1_cap", "freq1_input")
                 if os.path.exists(os.path.join(d, f))}
        if "power1_average" in files or "power1_input" in files:
            out.append((d, files))
    return out


def find_hwmon():
    """The card under load: a box shows the sensors of all eight GPUs of its node but runs on one -- the one whose power rises when a
    short matrix load runs on the visible device (first card with a power sensor if torch is not there)."""
    cards = all_hwmon()
    if len(cards) <= 1:
        return cards[0] if cards else (None, {})
    try:
        import torch
        pf = [c[1].get("power1_average") or c[1].get("power1_input") for c in cards]
        a = torch.randn((8192, 8192), device="cuda", dtype=torch.float16)
        peak = [0] * len(cards)
        t_end = time.time() + 2.0
        while time.time() < t_end:
            for _ in range(20):
                a @ a
            for i, f in enumerate(pf):
                peak[i] = max(peak[i], read_int(f) or 0)
        torch.cuda.synchronize()
        return cards[max(range(len(cards)), key=lambda i: peak[i])]
    except Exception:  # noqa: BLE001
        return cards[0]


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        card = next(iter(json.loads(out).values()))
    except Exception:  # noqa: BLE001 -- a sensor that cannot be read is a missing sample, not a failed run
        return None, None
    p = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k and v not in ("N/A", "")), None)
    c = next((v for k, v in card.items() if k.lower().startswith("sclk clock speed")), None)
    mhz = float(c.strip("()Mhz ")) if isinstance(c, str) and c.strip("()Mhz ").replace(".", "", 1).isdigit() else None
    return p, mhz


def stats(v):
    if not v:
        return None
    s = sorted(v)
    return {"n": len(s), "mean": round(statistics.fmean(s), 1), "p50": round(s[len(s) // 2], 1), "p95": round(s[(len(s) * 95) // 100], 1),
            "min": round(s[0], 1), "max": round(s[-1], 1)}


def bins(samples):
    """[second since the first sample, power samples, sclk samples] of every one-second bin"""
    out = {}
    for t, w, f in samples:
        if w is None:
            continue
        b = out.setdefault(int(t - samples[0][0]), ([], []))
        b[0].append(w)
        if f is not None:
            b[1].append(f)
    return [(k, v[0], v[1]) for k, v in sorted(out.items())]


def main():
    argv = sys.argv[1:]
    period = 0.02
    if argv and argv[0] == "--period":
        period = float(argv[1])
        argv = argv[2:]
    assert argv and argv[0] == "--", __doc__
    cmd = argv[1:]
    hw, files = find_hwmon()
    pfile = files.get("power1_average") or files.get("power1_input")
    samples = []          # (t, watts, sclk MHz)
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            t = time.time()
            if pfile:
                p = read_int(pfile)
                f = read_int(files["freq1_input"]) if "freq1_input" in files else None
                samples.append((t, p / 1e6 if p is not None else None, f / 1e6 if f is not None else None))
                stop.wait(period)
            else:
                p, f = smi_sample()
                samples.append((t, p, f))
                stop.wait(0.5)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.time()
    child = subprocess.run(cmd, capture_output=True, text=True)
    wall = time.time() - t0
    stop.set()
    th.join(timeout=15)
    pw = [s[1] for s in samples if s[1] is not None]
    busy_thr = 0.5 * max(pw) if pw else 0.0
    busy = [s for s in samples if s[1] is not None and s[1] >= busy_thr]
    cap = read_int(files["power1_cap"]) if "power1_cap" in files else None
    res = {
        "command": " ".join(cmd), "rc": child.returncode, "wall_s": round(wall, 2),
        "source": (hw + " (" + ", ".join(sorted(files)) + ")") if pfile else "rocm-smi --showpower --showclocks --json",
        "period_s": period if pfile else 0.5,
        "power_cap_W": cap / 1e6 if cap is not None else None,
        "power_W_all": stats(pw),
        "power_W_busy": stats([s[1] for s in busy]),
        "sclk_MHz_all": stats([s[2] for s in samples if s[2] is not None]),
        "sclk_MHz_busy": stats([s[2] for s in busy if s[2] is not None]),
        "busy_definition": "samples with power >= half of the maximum sample",
        "per_second": [[int(sec), round(statistics.fmean(w), 0), round(statistics.fmean(f), 0) if f else None]
                       for sec, w, f in bins(samples)],
        "child_stdout_tail": child.stdout.strip().splitlines()[-1][:400] if child.stdout.strip() else "",
    }
    print(json.dumps(res))
    if child.returncode != 0:
        sys.stderr.write(child.stderr[-2000:])
    return child.returncode


if __name__ == "__main__":
    sys.exit(main())
