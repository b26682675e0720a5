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
    for ref_name, our_name in _MAP.items():
        sys.modules[ref_name] = importlib.import_module(our_name)
    return sorted(_MAP)
