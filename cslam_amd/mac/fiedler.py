"""`frontend.mac_fiedler_solver: tracemin_lu` -- the reference's own Fiedler computation, called the way the reference calls it.

cslam/mac/mac.py:35-59 obtains the pair from a PRIVATE networkx function,
    la.algebraicconnectivity._get_fiedler_func('tracemin_lu')(L, x=None, normalized=False, tol=1e-8,
                                                              seed=np.random.RandomState(7))
(networkx 2.7 / 2.8 pinned by the reference, 3.4.2 in this image).  This module makes exactly that call: it is the host-side
solver a robot without a GPU keeps (the 'auto' policy of algebraic_connectivity_maximization.py selects it only there) and
the A/B partner of the chain-reduced HIP solvers, whose iterates are the same (tests/test_mac_gpu.py).  The restatement of the
algorithm that the tests check all solvers against lives in oracle/fiedler_oracle.py.
"""
import numpy as np


def fiedler_tracemin_lu(L, tol=1e-8, seed=None):
    """(lambda_2, v_2) of the connected, unnormalised Laplacian L through networkx's TraceMIN + SuperLU."""
    try:
        from networkx.linalg import algebraicconnectivity as ac
    except ImportError as e:                     # the reference depends on networkx; a host without it has no such solver
        raise RuntimeError("frontend.mac_fiedler_solver 'tracemin_lu' needs networkx (the reference's dependency): %s" % e)
    if seed is None:
        seed = np.random.RandomState(7)
    return ac._get_fiedler_func("tracemin_lu")(L, x=None, normalized=False, tol=tol, seed=seed)
