"""Route the reference's module names to this package without editing cslam.

    import cslam_amd.dropin; cslam_amd.dropin.install()      # before the cslam node imports them

After install(), `from cslam.nns_matching import NearestNeighborsMatching`,
`from cslam.loop_closure_sparse_matching import LoopClosureSparseMatching`,
`from cslam.algebraic_connectivity_maximization import ...`, `from cslam.vpr.netvlad import NetVLAD`
and `from cslam.vpr.cosplace import CosPlace` -- the imports made by
cslam/global_descriptor_loop_closure_detection.py:39-60 and
cslam/loop_closure_sparse_matching.py:2-4 -- resolve to the MI355X implementations, and so do
`cslam.lidar_pr.scancontext_matching` (lcsm.py:3), `cslam.lidar_pr.scancontext` (gdlcd.py:50) and `cslam.broker` (gdlcd.py:9,329).  Every other
cslam module (ROS glue, neighbour manager, lidar handler, ...) keeps coming from the installed
reference.
"""
import importlib
import importlib.abc
import importlib.util
import sys

_MAP = {
    "cslam.nns_matching": "cslam_amd.nns_matching",
    "cslam.loop_closure_sparse_matching": "cslam_amd.loop_closure_sparse_matching",
    "cslam.algebraic_connectivity_maximization": "cslam_amd.algebraic_connectivity_maximization",
    "cslam.mac.mac": "cslam_amd.mac.mac",
    "cslam.mac.utils": "cslam_amd.mac.utils",
    "cslam.vpr.netvlad": "cslam_amd.vpr.netvlad",
    "cslam.vpr.cosplace": "cslam_amd.vpr.cosplace",
    "cslam.lidar_pr.scancontext_matching": "cslam_amd.lidar_pr.scancontext_matching",
    "cslam.lidar_pr.scancontext": "cslam_amd.lidar_pr.scancontext",
    "cslam.broker": "cslam_amd.broker",
}


def install():
    """Eager form: import the ten replacements now and register them under the reference's names."""
    for ref_name, our_name in _MAP.items():
        sys.modules[ref_name] = importlib.import_module(our_name)
    return sorted(_MAP)


class _DropinFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Lazy form: answers the import of one of the ten reference names with the cslam_amd module of `_MAP`, imported
    at that moment (a process that never imports them -- every other ROS node sharing the PYTHONPATH -- pays nothing)."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname in _MAP:
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        # The import machinery stamps the module it gets back with the 'cslam.<name>' spec (module.__spec__ = spec).  The
        # module is the real cslam_amd one, so its own spec is put back in exec_module: importlib.reload, pkgutil, inspect
        # and pickling-by-spec keep seeing where it came from.
        module = importlib.import_module(_MAP[spec.name])
        self._own_spec = getattr(self, "_own_spec", {})
        self._own_spec[id(module)] = (module, module.__spec__)
        return module

    def exec_module(self, module):
        saved = getattr(self, "_own_spec", {}).pop(id(module), None)
        if saved is not None and saved[0] is module:
            module.__spec__ = saved[1]


def install_lazy():
    """Register the finder ahead of the path-based one; idempotent.  This is what `cslam_amd/shim/sitecustomize.py`
    (and the optional `.pth` line of INTEGRATION.md) calls, so that cslam itself stays unedited."""
    for f in sys.meta_path:
        if isinstance(f, _DropinFinder):
            return f
    f = _DropinFinder()
    sys.meta_path.insert(0, f)
    return f


def uninstall():
    """Undo install() / install_lazy() (tests)."""
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, _DropinFinder)]
    for ref_name, our_name in _MAP.items():
        m = sys.modules.get(ref_name)
        if m is not None and getattr(m, "__name__", None) == our_name:
            del sys.modules[ref_name]
