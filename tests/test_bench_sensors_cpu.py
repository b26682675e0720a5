"""bench.py's board sensors (the `board` entry of the bench line): a context figure that must never break a run -- no hwmon files, unreadable
files and a card that does not match the torch device all end in None or in a marked fallback, and a readable card is averaged correctly."""
import importlib.util
import os
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_board_sensors_absent_is_none():
    b = _bench()
    s = b.BoardSensors(torch, 0)
    if s.files:          # a box with an amdgpu card: nothing to check here
        return
    s.start()
    assert s.finish() is None


def test_board_sensors_reads_a_card(tmp_path, monkeypatch):
    b = _bench()
    hw = tmp_path / "card0" / "device" / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (hw / "power1_input").write_text("1250000000\n")
    (hw / "power1_cap").write_text("1400000000\n")
    (hw / "freq1_input").write_text("2100000000\n")
    import glob as _glob
    real = _glob.glob
    monkeypatch.setattr(_glob, "glob", lambda pat: [str(hw)] if pat.startswith("/sys/class/drm") else real(pat))
    s = b.BoardSensors(torch, 0)
    assert s.files and not s.matched
    s.start()
    time.sleep(0.08)
    r = s.finish()
    assert r["power_W_mean"] == 1250.0 and r["power_W_max"] == 1250.0 and r["power_cap_W"] == 1400.0
    assert r["sclk_MHz_mean"] == 2100.0 and r["samples"] >= 2 and "first card" in r["source"]
    (hw / "power1_input").write_text("garbage\n")        # an unreadable sensor is a missing sample, not an error
    s.start()
    time.sleep(0.03)
    assert s.finish() is None
