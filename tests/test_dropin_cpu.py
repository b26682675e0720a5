"""cslam_amd.dropin: the ten reference module names resolve to this package, with cslam itself unedited.

Two layers: (1) everywhere (also on the GPU box, where the reference checkout does not exist): a stand-in EMPTY `cslam`
package tree -- directories with empty __init__.py, no reference code -- proves that both install() forms map every
name of `_MAP`; (2) in the build container only: the reference's own test files run unmodified through the
start-up shim (tools/ref_dropin_proof.py).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

_CHECK = r"""
import importlib, sys
import cslam_amd.dropin as d
if MODE == "eager":
    d.install()
elif MODE == "lazy":
    d.install_lazy()
# MODE == "shim": sitecustomize has already registered the finder
assert len(d._MAP) == 10
for ref_name, our_name in d._MAP.items():
    m = importlib.import_module(ref_name)
    assert m.__name__ == our_name and m is sys.modules[our_name], (ref_name, m)
    # the module keeps its OWN import spec (importlib.reload / pkgutil / inspect see where it came from)
    assert m.__spec__ is not None and m.__spec__.name == our_name, (ref_name, m.__spec__)
before = sys.modules["cslam_amd.broker"].Broker
assert importlib.reload(sys.modules["cslam_amd.broker"]).Broker is not before      # reload re-executes the module
from cslam.nns_matching import NearestNeighborsMatching
from cslam.loop_closure_sparse_matching import LoopClosureSparseMatching
from cslam.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot
from cslam.vpr.netvlad import NetVLAD
from cslam.vpr.cosplace import CosPlace
from cslam.broker import Broker
import cslam_amd.nns_matching as ours
assert NearestNeighborsMatching is ours.NearestNeighborsMatching
# a module that is NOT in the map still comes from the (stand-in) cslam package
import cslam.untouched
assert cslam.untouched.MARK == "reference"
d.uninstall()
assert "cslam.nns_matching" not in sys.modules
print("OK", MODE)
"""


def _standin(tmp_path):
    for pkg in ("cslam", "cslam/vpr", "cslam/mac", "cslam/lidar_pr"):
        os.makedirs(tmp_path / pkg, exist_ok=True)
        (tmp_path / pkg / "__init__.py").write_text("")
    (tmp_path / "cslam" / "untouched.py").write_text("MARK = 'reference'\n")
    return str(tmp_path)


@pytest.mark.parametrize("mode", ["eager", "lazy", "shim"])
def test_install_maps_all_ten_modules(tmp_path, mode):
    standin = _standin(tmp_path)
    path = [ROOT, standin]
    if mode == "shim":
        path.insert(0, os.path.join(ROOT, "cslam_amd", "shim"))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(path), PYTHONDONTWRITEBYTECODE="1")
    env.pop("CSLAM_AMD_DROPIN", None)
    r = subprocess.run([sys.executable, "-c", f"MODE = {mode!r}\n" + _CHECK], env=env, capture_output=True, text=True,
                       cwd=standin)
    assert r.returncode == 0 and ("OK " + mode) in r.stdout, r.stdout + r.stderr


def test_shim_can_be_switched_off(tmp_path):
    standin = _standin(tmp_path)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "cslam_amd", "shim"), ROOT, standin]),
               CSLAM_AMD_DROPIN="0", PYTHONDONTWRITEBYTECODE="1")
    code = ("import importlib.util as u; s = u.find_spec('cslam.nns_matching'); "
            "print('none' if s is None else s.origin)")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=standin)
    assert r.returncode == 0 and r.stdout.strip() == "none", r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tests")), reason="reference checkout only exists in the build container")
def test_reference_unit_tests_pass_unmodified_through_the_shim():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import ref_dropin_proof
    finally:
        sys.path.pop(0)
    ok, out, routed = ref_dropin_proof.run(gpu=False)
    assert ok, out[-4000:]
    assert "18 passed" in out
    assert any("cslam.algebraic_connectivity_maximization -> cslam_amd." in ln for ln in routed)
    assert any("cslam.broker -> cslam_amd." in ln for ln in routed)
