"""`frontend.mac_fiedler_solver: tracemin_lu` -- the reference's own Fiedler computation, called the way the reference calls it.

cslam/mac/mac.py:35-59 obtains the pair from a PRIVATE networkx function,
    la.algebraicconnectivity._get_fiedler_func('tracemin_lu')(L, x=None, normalized=False, tol=1e-8,
                                                              seed=np.random.RandomState(7))
(networkx 2.7 / 2.8 pinned by the reference, 3.4.2 in this image).  This module makes exactly that call: it is the host-side
solver a robot without a GPU keeps (the 'auto' policy of algebraic_connectivity_maximization.py selects it only there) and
the A/B partner of the chain-reduced HIP solvers, whose iterates are the same (tests/test_mac_gpu.py).  The restatement of the
algorithm that the tests check all solvers against lives in oracle/fiedler_oracle.py.
"""
import warnings

import numpy as np

NETWORKX_TESTED = ("2.7", "3.4.2")       # the reference pins 2.7 / 2.8; this image has 3.4.2 (tests/test_mac_cpu.py runs on it)
_warned = []


def _networkx_tracemin():
    """networkx's private TraceMIN + SuperLU Fiedler function, or None (with ONE warning saying why) when networkx is missing or
    a version has moved the private symbol the reference relies on."""
    try:
        from networkx.linalg import algebraicconnectivity as ac
        return ac._get_fiedler_func("tracemin_lu")
    except Exception as e:                       # ImportError, AttributeError (symbol renamed), NetworkXError (method dropped)
        if not _warned:
            _warned.append(True)
            warnings.warn("frontend.mac_fiedler_solver 'tracemin_lu': networkx's _get_fiedler_func('tracemin_lu') is not available "
                          "(%s: %s); using this package's chain-reduced TraceMIN (same iterates, numpy/scipy only). Tested networkx "
                          "versions: %s .. %s" % (type(e).__name__, e, NETWORKX_TESTED[0], NETWORKX_TESTED[1]), RuntimeWarning)
        return None


def fiedler_tracemin_lu(L, tol=1e-8, seed=None):
    """(lambda_2, v_2) of the connected, unnormalised Laplacian L through networkx's TraceMIN + SuperLU.  Without networkx (or
    with a version that no longer has the private function) the same TraceMIN iteration runs on chain_solver.py's inner solves:
    a missing dependency must not turn into run_mac_solver's "Fiedler solve failed, re-draw the start point" retries, which end
    in the pseudo-greedy selection without a word."""
    if seed is None:
        seed = np.random.RandomState(7)
    f = _networkx_tracemin()
    if f is None:
        from .chain_solver import fiedler_tracemin_chain
        return fiedler_tracemin_chain(L, tol=tol, seed=seed)
    return f(L, x=None, normalized=False, tol=tol, seed=seed)
