"""cslam_amd -- MI355X-native loop-closure place recognition for Swarm-SLAM (cslam).

Hand-written HIP (gfx950) behind a C ABI (include/cslam_hip.h, libcslam_hip.so) with
Python host classes that keep the reference's duck-typed API:

    cslam_amd.nns_matching.NearestNeighborsMatching            (cslam/nns_matching.py)
    cslam_amd.loop_closure_sparse_matching.LoopClosureSparseMatching
    cslam_amd.algebraic_connectivity_maximization.{AlgebraicConnectivityMaximization, EdgeInterRobot}
    cslam_amd.vpr.netvlad.NetVLAD / cslam_amd.vpr.cosplace.CosPlace
"""
__version__ = "0.1.0"
