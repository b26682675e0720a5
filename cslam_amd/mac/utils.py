"""Graph-Laplacian helpers of the MAC sparsifier (reference: cslam/mac/utils.py).

Vectorised numpy assembly; the triplet ORDER is kept identical to the reference's
per-edge loop ((i,i), (j,j), (i,j), (j,i) for each edge in turn, utils.py:61-84) so that
scipy's duplicate summation adds the same numbers in the same order and L is bit-identical.
"""
from collections import namedtuple

import numpy as np
from scipy.sparse import coo_matrix, csr_matrix

# Define Edge container (reference: cslam/mac/utils.py:13)
Edge = namedtuple('Edge', ['i', 'j', 'weight'])


def _laplacian_from_arrays(i, j, w, n):
    i = np.asarray(i, dtype=np.int64).ravel()
    j = np.asarray(j, dtype=np.int64).ravel()
    w = np.asarray(w, dtype=np.float64).ravel()
    rows = np.stack([i, j, i, j], axis=1).ravel()
    cols = np.stack([i, j, j, i], axis=1).ravel()
    data = np.stack([w, w, -w, -w], axis=1).ravel()
    return csr_matrix(coo_matrix((data, (rows, cols)), shape=[n, n]))


class EdgeArrays(object):
    """Column form of a list of Edge (i, j, weight arrays), in the same order.  Lets the million
    odometry edges of a large pose graph skip the per-edge Python objects; the Laplacian built from
    it is bit-identical to the one built from the equivalent list."""

    def __init__(self, i, j, weight):
        self.i = np.asarray(i, dtype=np.int64)
        self.j = np.asarray(j, dtype=np.int64)
        self.weight = np.asarray(weight, dtype=np.float64)

    def __len__(self):
        return len(self.i)

    @staticmethod
    def from_edges(edges):
        if isinstance(edges, EdgeArrays):
            return edges
        if len(edges) == 0:
            return EdgeArrays(np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0))
        arr = np.array([(e.i, e.j, e.weight) for e in edges], dtype=np.float64)
        return EdgeArrays(arr[:, 0].astype(np.int64), arr[:, 1].astype(np.int64), arr[:, 2])

    def concat(self, other):
        o = EdgeArrays.from_edges(other)
        return EdgeArrays(np.concatenate([self.i, o.i]), np.concatenate([self.j, o.j]),
                          np.concatenate([self.weight, o.weight]))


def weight_graph_lap_from_edge_list(edges, num_vars):
    """Sparse weighted graph Laplacian from a list of Edge (reference utils.py:47-84)."""
    if isinstance(edges, EdgeArrays):
        if len(edges) == 0:
            return csr_matrix((num_vars, num_vars), dtype=np.float64)
        return _laplacian_from_arrays(edges.i, edges.j, edges.weight, num_vars)
    if len(edges) == 0:
        return csr_matrix((num_vars, num_vars), dtype=np.float64)
    arr = np.array([(e.i, e.j, e.weight) for e in edges], dtype=np.float64)
    return _laplacian_from_arrays(arr[:, 0].astype(np.int64), arr[:, 1].astype(np.int64), arr[:, 2], num_vars)


def weight_graph_lap_from_edges(edges, weights, num_poses):
    """Sparse weighted graph Laplacian from an [m,2] index array and weights (utils.py:87-126)."""
    edges = np.asarray(edges)
    if edges.size == 0:
        return csr_matrix((num_poses, num_poses), dtype=np.float64)
    return _laplacian_from_arrays(edges[:, 0], edges[:, 1], weights, num_poses)


def split_measurements(measurements):
    """odometry (|i-j| <= 1) vs loop closures (reference utils.py:129-146)."""
    odom = [m for m in measurements if abs(m.j - m.i) <= 1]
    lc = [m for m in measurements if abs(m.j - m.i) > 1]
    return odom, lc


def select_measurements(measurements, w):
    """Edges whose selection weight is exactly one (reference utils.py:149-158)."""
    assert len(measurements) == len(w)
    return [m for m, wi in zip(measurements, w) if wi == 1.0]
