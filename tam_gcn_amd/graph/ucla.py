"""N-UCLA 20-joint skeleton graph (reference: graph/ucla.py:7-32).

The topology is stored as a parent table (joint k+1 -> parent, 0 = root);
joint 3 (spine) is the root.  ``Graph(labeling_mode='spatial', scale=1).A``
matches the reference's array bit for bit (tests/test_graph.py).
"""
from . import tools

#            1  2  3  4  5  6  7  8  9 10  11  12 13  14  15  16 17  18  19  20
_PARENTS = (2, 3, 0, 3, 3, 5, 6, 7, 3, 9, 10, 11, 1, 13, 14, 15, 1, 17, 18, 19)

num_node = len(_PARENTS)
self_link, inward, outward, neighbor = tools.links_from_parents(_PARENTS)


class Graph(tools.SpatialGraph):
    parents = _PARENTS

    def __init__(self, labeling_mode='spatial', scale=1):
        super().__init__(labeling_mode)
