"""Frank-Wolfe maximisation of the algebraic connectivity (reference: cslam/mac/mac.py).

Same iterates as the reference's MAC class: L(w) = L_fixed + sum_k w_k weight_k L_k,
gradient g_k = weight_k (v_i - v_j)^2 from the Fiedler vector, linear maximisation oracle =
top-k of the gradient, step 2/(it+2), dual bound / gap test, final rounding with weight
tie-break.  The per-edge Python loops of the reference (mac.py:123-129, 178-180) are
vectorised with the same operation order, so results are bit-identical.
"""
import os
from collections import namedtuple

import numpy as np

from .fiedler import fiedler_tracemin_lu
from .chain_solver import fiedler_tracemin_chain
from .utils import weight_graph_lap_from_edge_list, weight_graph_lap_from_edges

MACResult = namedtuple('MACResult', ['w', 'F_unrounded', 'objective_values', 'duality_gaps'])


class MAC:

    def __init__(self, fixed_measurements, candidate_measurements, num_poses, fiedler_solver='tracemin_lu'):
        # 'tracemin_lu'  sparse LU inner solves, the reference's path (networkx + SuperLU)
        # 'chain'        chain-reduced inner solves, host numpy (same iterates to ~1e-15)
        # 'chain_gpu'    chain-reduced inner solves and every O(n) step in HIP (large graphs)
        # 'chain_hip'    the same computation behind the C ABI's one-call `cslam_fiedler` (native host code, no torch ops)
        self.fiedler_solver = fiedler_solver
        self.solver_may_fall_back = False         # set by the 'auto' policy (acm.py): chain_hip -> chain_gpu without rocSOLVER
        self._fixed = fixed_measurements          # kept for the native Frank-Wolfe loop (fw_subset, 'chain_hip')
        self.L_odom = weight_graph_lap_from_edge_list(fixed_measurements, num_poses)
        self.num_poses = num_poses
        self.weights = np.array([m.weight for m in candidate_measurements])
        self.edge_list = np.array([(m.i, m.j) for m in candidate_measurements])
        self.verbose = False

    def find_fiedler_pair(self, L, method='tracemin_lu', tol=1e-8):
        """(lambda_2(L), v_2(L)); reference mac.py:35-59."""
        assert method == 'tracemin_lu'
        if self.fiedler_solver == 'chain_gpu':
            import os
            from .chain_solver_gpu import fiedler_tracemin_chain_gpu
            if os.environ.get('CSLAM_MAC_TIMING'):
                import time
                st = {'t0': time.perf_counter()}
                out = fiedler_tracemin_chain_gpu(L, tol=tol, seed=np.random.RandomState(7), stats=st)
                print('      [fiedler: nJ=%d setup %.0f ms, %d TraceMIN iterations %.0f ms = %.2f ms each]' % (
                    st['nJ'], st['setup_s'] * 1e3, st['iters'], st['loop_s'] * 1e3, st['loop_s'] * 1e3 / max(st['iters'], 1)), flush=True)
                return out
            return fiedler_tracemin_chain_gpu(L, tol=tol, seed=np.random.RandomState(7))
        if self.fiedler_solver == 'chain_hip':
            import os
            from .._lib import CslamLimitError, CslamUnsupportedError
            from .chain_solver_gpu import fiedler_tracemin_chain_gpu, fiedler_tracemin_hip
            st = {} if os.environ.get('CSLAM_MAC_TIMING') else None
            try:
                out = fiedler_tracemin_hip(L, tol=tol, seed=7, stats=st)
            except CslamLimitError:
                # CSLAM_E_LIMIT: more junctions than the dense factor takes -> the torch-driven solver's sparse-LU junction solve
                return fiedler_tracemin_chain_gpu(L, tol=tol, seed=np.random.RandomState(7))
            except CslamUnsupportedError:
                # CSLAM_E_UNSUPPORTED: rocBLAS / rocSOLVER not found on this host.  Only the 'auto' policy may change solver
                if not self.solver_may_fall_back:
                    raise
                self.fiedler_solver = 'chain_gpu'
                return fiedler_tracemin_chain_gpu(L, tol=tol, seed=np.random.RandomState(7))
            if st is not None:
                print('      [fiedler (cslam_fiedler): %d TraceMIN iterations, %.0f ms in all]' % (st['iters'], st['total_s'] * 1e3), flush=True)
            return out
        if self.fiedler_solver == 'chain':
            return fiedler_tracemin_chain(L, tol=tol, seed=np.random.RandomState(7))
        return fiedler_tracemin_lu(L, tol=tol, seed=np.random.RandomState(7))

    def combined_laplacian(self, w, tol=1e-10):
        """L(w): fixed edges plus candidates weighted by w (reference mac.py:61-77)."""
        idx = np.where(w > tol)
        prod = w[idx] * self.weights[idx]
        C1 = weight_graph_lap_from_edges(self.edge_list[idx], prod, self.num_poses)
        return self.L_odom + C1

    def evaluate_fiedler_pair(self, w, method='tracemin_lu', tol=1e-8):
        return self.find_fiedler_pair(self.combined_laplacian(w), method, tol)

    def evaluate_objective(self, w):
        return self.find_fiedler_pair(self.combined_laplacian(w))[0]

    def grad_from_fiedler(self, fiedler_vec):
        """Supergradient of lambda_2 w.r.t. w (reference mac.py:112-130):
        kdelta = weight_k (v_i - v_j);  grad_k = kdelta (v_i - v_j)."""
        if len(self.weights) == 0:
            return np.zeros(0)
        d = fiedler_vec[self.edge_list[:, 0]] - fiedler_vec[self.edge_list[:, 1]]
        return (self.weights * d) * d

    def round_solution(self, w, k):
        """Top-k indicator, ties broken arbitrarily (reference mac.py:132-147)."""
        idx = np.argpartition(w, -k)[-k:]
        rounded = np.zeros(len(w))
        if k > 0:
            rounded[idx] = 1.0
        return rounded

    def simple_random_round(self, w, k):
        """Independent Bernoulli rounding (reference mac.py:149-166)."""
        x = np.zeros(len(w))
        for i in range(len(w)):
            if w[i] > np.random.rand():
                x[i] = 1.0
        return x

    def round_solution_tiebreaker(self, w, k, decimal_tol=10):
        """Top-k of w rounded to `decimal_tol` decimals, ties towards larger edge weight
        (reference mac.py:168-189)."""
        zipped = np.empty(len(w), dtype=[('w', 'float'), ('weight', 'float')])
        zipped['w'] = w.round(decimals=decimal_tol)
        zipped['weight'] = self.weights
        idx = np.argpartition(zipped, -k, order=['w', 'weight'])[-k:]
        rounded = np.zeros(len(w))
        if k > 0:
            rounded[idx] = 1.0
        return rounded

    def _fw_subset_hip(self, w_init, k, max_iters, duality_gap_tol):
        """The same loop behind the C ABI (`cslam_mac_fw_subset`, csrc/fiedler.hip): Laplacian updates, Fiedler pairs, gradient,
        top-k and rounding in native code -- what a host without Python calls."""
        import ctypes as C
        from .. import _lib
        from .utils import EdgeArrays
        lib = _lib.load()
        fx = EdgeArrays.from_edges(self._fixed)
        fi, fj, fw = (np.ascontiguousarray(a) for a in (fx.i.astype(np.int64), fx.j.astype(np.int64), fx.weight.astype(np.float64)))
        ci = np.ascontiguousarray(self.edge_list[:, 0], dtype=np.int64)
        cj = np.ascontiguousarray(self.edge_list[:, 1], dtype=np.int64)
        cw = np.ascontiguousarray(self.weights, dtype=np.float64)
        w0 = np.ascontiguousarray(w_init, dtype=np.float64)
        sel, wu, up = np.empty(len(cw)), np.empty(len(cw)), C.c_double(0.0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        try:
            import torch
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        except ImportError:                                       # pragma: no cover
            st = None
        _lib.check(lib.cslam_mac_fw_subset(int(self.num_poses), len(fw), p(fi), p(fj), p(fw), len(cw), p(ci), p(cj), p(cw), p(w0),
                                           int(k), int(max_iters), float(duality_gap_tol), 1e-8, p(sel), p(wu), C.byref(up), None, st))
        return sel, wu, float(up.value)

    def fw_subset(self, w_init, k, max_iters=5, duality_gap_tol=1e-8, trace=None):
        """Frank-Wolfe on the relaxed subset selection (reference mac.py:191-233).
        Returns (rounded solution, unrounded iterate, dual upper bound)."""
        if self.fiedler_solver == 'chain_hip' and trace is None and len(self.weights) > 0 and self.num_poses > 4 \
                and os.environ.get('CSLAM_MAC_FW', 'hip') != 'python':
            from .._lib import CslamLimitError, CslamUnsupportedError
            try:
                return self._fw_subset_hip(w_init, k, max_iters, duality_gap_tol)
            except CslamLimitError:
                pass                      # more junctions than the dense factor takes: the Python loop, whose Fiedler pairs
                                          # fall back to the sparse-LU junction solve (find_fiedler_pair)
            except CslamUnsupportedError:
                if not self.solver_may_fall_back:
                    raise                 # an explicitly requested 'chain_hip' without its libraries stays an error
                self.fiedler_solver = 'chain_gpu'
            # (every other CslamHipError propagates; CslamGraphError is the caller's retry policy, acm.py:436-466)
        # Host fallback (more junctions than the dense factor takes, or no HIP solver): Frank-Wolfe over the relaxed selection weights.
        # `relaxed` walks from w_init towards the vertex the linear oracle names (the k heaviest gradient entries), step 2 / (t + 2);
        # `dual_bound` is the best upper bound f + <grad, vertex - relaxed> seen; the walk ends when it closes on the objective.
        relaxed, dual_bound, closed = w_init, float("inf"), False
        for t in range(max_iters):
            objective, fiedler_vec = self.evaluate_fiedler_pair(relaxed)
            gradient = self.grad_from_fiedler(fiedler_vec)
            if trace is not None:
                trace.append((float(objective), float(np.linalg.norm(gradient))))
            vertex = self.round_solution(gradient, k)
            towards = vertex - relaxed
            dual_bound = min(dual_bound, objective + gradient @ towards)
            closed = dual_bound - objective < duality_gap_tol
            if closed:
                break
            relaxed = relaxed + (2.0 / (t + 2.0)) * towards
        if self.verbose:
            print("Duality gap tolerance reached, found optimal solution" if closed else "Reached maximum iterations")
        return self.round_solution_tiebreaker(relaxed, k), relaxed, dual_bound
