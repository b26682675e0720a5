"""GPU test: a plain-C program drives the hot path through include/cslam_hip.h only (-m gpu)."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_client(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "cslam_amd")
    r = subprocess.run(["gcc", "-std=c11", "-O1", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c"), "-L" + libdir, "-lcslam_hip", "-lm",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ)
    # The program needs libamdhip64.so.7.  A fresh box pages the SYSTEM copy (/opt/rocm/lib) in from cold storage: 110 s measured
    # for the first process that touches it (profiles/r04_v74_tests_gpu_durations.log), 40 % of the suite's margin for a program
    # that runs in under a second.  torch bundles the same runtime (same SONAME, file name without the version) and this pytest
    # process has already loaded it: hand that copy to the loader through a directory holding the versioned name.  The client
    # itself is unchanged -- plain C against include/cslam_hip.h -- and falls back to the system runtime without torch.
    path = [libdir]
    try:
        import torch
        tl = os.path.join(os.path.dirname(torch.__file__), "lib")
        if os.path.exists(os.path.join(tl, "libamdhip64.so")):
            rt = tmp_path / "rt"
            rt.mkdir()
            for soname, name in (("libamdhip64.so.7", "libamdhip64.so"), ("libhsa-runtime64.so.1", "libhsa-runtime64.so")):
                if os.path.exists(os.path.join(tl, name)):
                    os.symlink(os.path.join(tl, name), str(rt / soname))
            path += [str(rt), tl]
    except ImportError:
        pass
    env["LD_LIBRARY_PATH"] = ":".join(path + ["/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C ABI smoke ok" in r.stdout
