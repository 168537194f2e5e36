"""Synthetic large skeleton graph for the V=64 roofline configuration
(BASELINE.json configs[4]; the reference has no such graph).

A balanced 4-ary tree over ``num_node`` joints pushed through the same
``get_spatial_graph`` recipe as the real skeletons.
"""
from . import tools


class Graph(tools.SpatialGraph):
    def __init__(self, labeling_mode='spatial', num_node=64, arity=4):
        self.parents = tuple(0 if k == 0 else (k - 1) // arity + 1
                             for k in range(num_node))
        super().__init__(labeling_mode)
