"""Skeleton-graph adjacency construction (numpy, float64).

Behavioural counterpart of the reference's ``graph/tools.py`` for the one
recipe the CTR-GCN hot path consumes, ``get_spatial_graph``
(reference: graph/tools.py:10-14 edge2mat, :28-36 normalize_digraph,
:39-44 get_spatial_graph).  Written vectorised instead of with python loops;
results are bit-identical (checked in tests/test_graph.py against arrays
generated from the reference, tests/golden/graphs.npz).

Conventions (same as the reference):
  * a link ``(i, j)`` sets ``A[j, i] = 1``;
  * normalisation divides every *column* by its sum (columns that sum to
    zero stay zero);
  * the spatial graph is ``stack(I, norm(inward), norm(outward))``.
"""
import numpy as np


def edge2mat(link, num_node):
    A = np.zeros((num_node, num_node))
    if len(link):
        idx = np.asarray(link, dtype=np.int64)
        A[idx[:, 1], idx[:, 0]] = 1
    return A


def normalize_digraph(A):
    col = A.sum(axis=0)
    scale = np.zeros_like(col)
    nz = col > 0
    scale[nz] = col[nz] ** (-1)
    # the reference multiplies by a dense diagonal matrix; scaling columns
    # directly gives the same float64 values (one multiply per entry).
    return A * scale[None, :]


def get_spatial_graph(num_node, self_link, inward, outward):
    return np.stack((edge2mat(self_link, num_node),
                     normalize_digraph(edge2mat(inward, num_node)),
                     normalize_digraph(edge2mat(outward, num_node))))


def links_from_parents(parents_1based):
    """``parents_1based[k]`` is the 1-based parent of joint ``k+1`` (0 = root).

    Returns (self_link, inward, outward, neighbor) with 0-based indices, in
    the reference's orientation: inward links point child -> parent.
    """
    n = len(parents_1based)
    self_link = [(i, i) for i in range(n)]
    inward = [(k, p - 1) for k, p in enumerate(parents_1based) if p > 0]
    outward = [(j, i) for (i, j) in inward]
    return self_link, inward, outward, inward + outward


class SpatialGraph:
    """Common base: ``Graph(labeling_mode='spatial').A`` -> float64 (3, V, V)."""
    parents = ()

    def __init__(self, labeling_mode='spatial', **_unused):
        self.num_node = len(self.parents)
        (self.self_link, self.inward, self.outward,
         self.neighbor) = links_from_parents(self.parents)
        self.A = self.get_adjacency_matrix(labeling_mode)

    def get_adjacency_matrix(self, labeling_mode=None):
        if labeling_mode is None:
            return self.A
        if labeling_mode != 'spatial':
            raise ValueError()
        return get_spatial_graph(self.num_node, self.self_link,
                                 self.inward, self.outward)
