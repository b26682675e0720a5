"""Start-up hook that routes cslam's hot-path modules to cslam_amd WITHOUT editing cslam.

    PYTHONPATH=<repo>/cslam_amd/shim:<repo>:$PYTHONPATH ros2 run cslam loop_closure_detection_node.py ...

Python imports a module named `sitecustomize` at interpreter start when one is importable; this one registers
`cslam_amd.dropin`'s lazy finder, so `from cslam.nns_matching import NearestNeighborsMatching` (and the nine other
names of `cslam_amd.dropin._MAP`) resolve to the MI355X implementations while every other cslam module keeps coming
from the installed reference.  CSLAM_AMD_DROPIN=0 switches it off.  A distribution's own sitecustomize (found later
on the path) is chained so that it keeps working.
"""
import os
import sys

if os.environ.get("CSLAM_AMD_DROPIN", "1") != "0":
    try:
        import cslam_amd.dropin as _dropin
        _dropin.install_lazy()
    except Exception as _e:                                  # never break interpreter start-up
        sys.stderr.write("cslam_amd shim: drop-in not installed (%r)\n" % (_e,))

# chain to the sitecustomize this one shadows, if any
_here = os.path.dirname(os.path.abspath(__file__))
_rest = [p for p in sys.path if os.path.abspath(p or ".") != _here]
try:
    import importlib.machinery as _m
    _spec = _m.PathFinder.find_spec("sitecustomize", _rest)
    if _spec is not None and _spec.loader is not None and os.path.abspath(_spec.origin or "") != os.path.abspath(__file__):
        import importlib.util as _u
        _mod = _u.module_from_spec(_spec)
        _spec.loader.exec_module(_mod)
except Exception:
    pass
