#!/usr/bin/env python
"""Drop-in proof (BUILD CONTAINER ONLY -- /root/reference does not exist on the GPU box and this script is never
run there): the reference's own unit tests, unmodified, against this package.

cslam stays unedited: the interpreter is started with `cslam_amd/shim` on PYTHONPATH, whose sitecustomize registers
`cslam_amd.dropin`'s finder, so the tests' `from cslam.algebraic_connectivity_maximization import ...`,
`from cslam.broker import Broker`, `from cslam.loop_closure_sparse_matching import ...` resolve to cslam_amd.
Runs tests/test_algebraic_connectivity.py and tests/test_broker.py (host-only: 18 tests); with --gpu also
tests/test_sparse_matching.py (needs an MI355X for the descriptor banks).

The only stub is a 4-line `numba` module (`jit` = identity): the reference imports numba in cslam/mac/utils.py:9-10 and
never uses it, and with the drop-in installed that module is not even imported.  Nothing is written under
/root/reference (no bytecode, no pytest cache).
"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CSLAM_REFERENCE", "/root/reference")


def run(gpu=False, quiet=True):
    if not os.path.isdir(os.path.join(REF, "tests")):
        raise SystemExit(f"{REF}/tests not found: this proof only runs where the reference checkout exists")
    files = ["test_algebraic_connectivity.py", "test_broker.py"] + (["test_sparse_matching.py"] if gpu else [])
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "numba"))
        with open(os.path.join(tmp, "numba", "__init__.py"), "w") as f:
            f.write("def jit(*a, **k):\n    if len(a) == 1 and callable(a[0]) and not k:\n        return a[0]\n"
                    "    return lambda fn: fn\n")
        # a verdict file written by a conftest-free plugin: which module objects the tests actually exercised
        probe = os.path.join(tmp, "probe_plugin.py")
        with open(probe, "w") as f:
            f.write("import sys\n"
                    "def pytest_sessionfinish(session, exitstatus):\n"
                    "    names = ['cslam.algebraic_connectivity_maximization', 'cslam.broker',\n"
                    "             'cslam.loop_closure_sparse_matching', 'cslam.nns_matching', 'cslam.mac.mac']\n"
                    "    print()\n"
                    "    for n in names:\n"
                    "        m = sys.modules.get(n)\n"
                    "        print('DROPIN', n, '->', getattr(m, '__name__', None), getattr(m, '__file__', None))\n")
        env = dict(os.environ)
        env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "cslam_amd", "shim"), ROOT, REF, tmp])
        env["PYTHONDONTWRITEBYTECODE"] = "1"
        env.pop("CSLAM_AMD_DROPIN", None)
        cmd = [sys.executable, "-m", "pytest", "-p", "no:cacheprovider", "-p", "probe_plugin", "-s",
               "-q" if quiet else "-v"] + files
        r = subprocess.run(cmd, cwd=os.path.join(REF, "tests"), env=env, capture_output=True, text=True)
    out = r.stdout + r.stderr
    routed = [ln for ln in out.splitlines() if ln.startswith("DROPIN")]
    ok = r.returncode == 0 and routed and all("cslam_amd" in ln for ln in routed if "-> None" not in ln)
    return ok, out, routed


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true", help="also run test_sparse_matching.py (needs an MI355X)")
    a = ap.parse_args()
    ok, out, routed = run(a.gpu, quiet=False)
    print(out[-6000:])
    print("drop-in proof:", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)
